#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of tools/prof_news.py.
# usage: tools/gpu_profile.sh <tag> [n_news passes S D h]
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
shift
ARGS="${@:-1310 6 50 768 16}"
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_news.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; }
for pass in "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/tools/prof_news.py $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $OUT/pmc_$name.log; }
done
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "at::native\|rocclr\|^void  " $OUT/summary.txt
# the same command the bench line comes from, under the kernel tracer
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_trace.log 2>&1 || { echo "bench trace failed"; tail -5 $OUT/bench_trace.log; }
tail -1 $OUT/bench_trace.log | cut -c1-900
find $OUT/bench_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}' 
