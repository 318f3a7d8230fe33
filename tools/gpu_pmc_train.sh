#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of tools/prof_train.py (NRMS grad steps).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-train}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_train.py 4 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; }
for pass in "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
            "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/tools/prof_train.py 2 > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $OUT/pmc_$name.log; }
done
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "at::native\|rocclr\|^void  " $OUT/summary.txt | cut -c1-${COLS:-330} | head -${LINES_:-60}
