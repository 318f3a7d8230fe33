#!/bin/bash
# usage: tools/gpu_trace.sh <tag> <script> [args]  -- rocprofv3 kernel trace + per-(kernel,grid) summary
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/$@ > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
tail -2 $OUT/run.log | cut -c1-300
python3 $R/tools/summarize_prof.py $OUT | grep -v "at::native\|rocclr" | head -40
