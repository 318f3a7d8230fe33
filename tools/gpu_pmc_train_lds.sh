#!/bin/bash
# LDS counters of the grad step only (one rocprofv3 --pmc pass, kernel trace only) + dW micro-bench.  usage: tools/gpu_pmc_train_lds.sh <tag>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${1:-ldst}
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 $R/tools/prof_train.py 2 > $OUT/pmc_lds.log 2>&1 || tail -3 $OUT/pmc_lds.log
python3 $R/tools/summarize_prof.py $OUT 2>&1 | grep -v "at::native\|rocclr\|^void  " | grep "grid=129024\|grid=196608\|grid=92160\|grid=124416\|mha_bwd" | cut -c1-260
cd $R && python tools/bench_dw.py 2>&1 | grep -v amdgpu && python -m pytest tests/test_hip_grads.py -q -m gpu -k "weight_gradient or live_rows or golden" 2>&1 | tail -2
