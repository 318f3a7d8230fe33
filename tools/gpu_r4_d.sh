cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3 4; do python -m pytest tests/test_hip_train_step.py -q 2>&1 | tail -n 1; done
python -m pytest tests -m gpu -q 2>&1 | tail -n 12 > gpurun_out/r4_gpu_suite.log; tail -n 3 gpurun_out/r4_gpu_suite.log
python tools/bench_train.py > gpurun_out/r4_train.log 2>&1; cut -c1-100 gpurun_out/r4_train.log | tail -n 5
