#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3e
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_grads.py tests/test_hip_training.py tests/test_hip_random_shapes.py -x -q > gpurun_out/r3e/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -8 gpurun_out/r3e/tests.log
[ $rc -ne 0 ] && exit $rc
python tools/bench_train_ab.py XNRS_GEMM_DW_TILE 128 256 2>&1 | grep -v amdgpu.ids
XNRS_GEMM_DW_TILE=128 python tools/bench_train_ab.py XNRS_GEMM_DW 1 2>&1 | grep -v amdgpu.ids | tail -2
