#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3f
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_news_fused.py tests/test_hip_parity.py -x -q > gpurun_out/r3f/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r3f/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_news_fused.py 2>&1 | grep -v amdgpu.ids | tail -30
