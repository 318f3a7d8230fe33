#!/bin/bash
# round 4, call g: side lane of the backward -- A/B timing, gradient equality, the grad / train-step / capture tests
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 300 python3 tools/bench_side_lane.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_side_lane.txt
timeout -k 10 600 python -m pytest tests/test_hip_grads.py tests/test_hip_train_step.py tests/test_hip_random_shapes.py tests/test_hip_two_ranks.py -q -x 2>&1 | tail -n 5
