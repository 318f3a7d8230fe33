#!/bin/bash
# round 4, call i: fold cache + per-tower gradient sums: full GPU suite, the three grad steps, launch census
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 6
timeout -k 10 300 python3 tools/bench_side_lane.py 2>&1 | grep -v amdgpu.ids | tail -n 3
timeout -k 10 200 python3 tools/prof_train_ops.py nrms 2>&1 | grep -v amdgpu.ids | head -n 24
