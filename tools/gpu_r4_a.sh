set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_train_step.py -x -q 2>&1 | tail -n 30 > gpurun_out/r4_t1.log
python -m pytest tests/test_hip_grads.py tests/test_hip_training.py tests/test_hip_random_shapes.py tests/test_hip_two_ranks.py -x -q 2>&1 | tail -n 15 > gpurun_out/r4_t2.log
python tools/bench_train.py > gpurun_out/r4_train.log 2>&1
tail -n 8 gpurun_out/r4_t1.log; tail -n 5 gpurun_out/r4_t2.log; tail -n 12 gpurun_out/r4_train.log | cut -c1-1500
