#!/bin/bash
# Runs on the GPU box (via gpurun): FETCH_SIZE / WRITE_SIZE PMC passes (each in its own run, kernel trace only) of
# tools/prof_news.py for the environment it is started with.  usage: [XNRS_...=v] tools/gpu_traffic.sh <tag>
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-traffic}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/tools/prof_news.py 1310 4 50 768 16 > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $OUT/pmc_$name.log; }
done
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -i "gemm\|kernel_name\|Kernel" $OUT/summary.txt | cut -c1-260
