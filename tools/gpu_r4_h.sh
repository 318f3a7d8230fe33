#!/bin/bash
# round 4, call h: integrated gradients -- tests, the IG extra (loop with the unused weight gradients skipped; batched form)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 400 python -m pytest tests/test_hip_explain.py tests/test_hip_grads.py tests/test_hip_train_step.py -q -x 2>&1 | tail -n 8
timeout -k 10 300 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ig.txt
import json, torch, bench
print(json.dumps(bench.ig_step_extra(torch.device("cuda", 0)), indent=1))
PY
