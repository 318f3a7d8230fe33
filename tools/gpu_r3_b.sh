#!/bin/bash
# round 3: fused additive encoder -- parity tests, then StandardRec / NAML forward timing (A/B by knob) + kernel trace
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3b
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_additive_fused.py -x -q > gpurun_out/r3b/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r3b/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_other_models.py > gpurun_out/r3b/ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3b/ab.log | tail -20
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3b/trace_standard -- python3 $R/tools/prof_other_models.py standard 5 > $R/gpurun_out/r3b/trace_standard.log 2>&1)
python3 tools/trace_summary.py gpurun_out/r3b/trace_standard > gpurun_out/r3b/trace_standard_summary.txt 2>&1
head -8 gpurun_out/r3b/trace_standard_summary.txt | cut -c1-170
