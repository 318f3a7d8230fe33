#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3c
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_additive_fused.py tests/test_hip_parity.py tests/test_hip_naml_ids.py -x -q > gpurun_out/r3c/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -8 gpurun_out/r3c/tests.log
python tools/bench_af.py 2>&1 | grep -v amdgpu.ids
XNRS_AF_FBUF=2 python tools/bench_af.py 2>&1 | grep fused
python tools/bench_af.py 2560 50 768 256 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/bench_other_models.py > gpurun_out/r3c/ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r3c/ab.log | tail -20
