#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3d
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_additive_fused.py tests/test_hip_parity.py tests/test_hip_naml_ids.py -x -q > gpurun_out/r3d/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -8 gpurun_out/r3d/tests.log
python tools/bench_af.py 2>&1 | grep -v amdgpu.ids
python tools/bench_af.py 2560 50 768 256 2>&1 | grep -v "amdgpu.ids\|plain"
python tools/bench_af.py 3840 50 768 256 2>&1 | grep -v "amdgpu.ids\|plain"
python tools/bench_af.py 8192 30 768 256 2>&1 | grep -v "amdgpu.ids\|plain"
python tools/bench_af.py 8192 30 320 256 2>&1 | grep -v "amdgpu.ids\|plain"
