#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3g
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r3g/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_news_fused.py 2>&1 | grep -v amdgpu.ids | tail -6
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
print('latency', bench.latency_extra(torch.device('cuda', 0)))
PY
