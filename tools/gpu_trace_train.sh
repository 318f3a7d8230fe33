#!/bin/bash
# kernel trace of one grad-step driver: tools/gpu_trace_train.sh <model> [steps]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
M=${1:-naml}; N=${2:-6}
OUT=$R/gpurun_out/trace_$M
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/prof_train.py $N $M > $OUT/log.txt 2>&1 || tail -n 3 $OUT/log.txt
python3 $R/tools/trace_summary.py $OUT/t --all > $OUT/summary.txt 2>&1
rm -rf $OUT/t
cut -c1-150 $OUT/summary.txt | head -n 42
