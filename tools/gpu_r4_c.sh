set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/debug_graph_step.py 8 > gpurun_out/r4_dbg.log 2>&1
python -m pytest tests/test_hip_train_step.py tests/test_hip_grads.py tests/test_hip_training.py tests/test_hip_random_shapes.py -q 2>&1 | tail -n 15 > gpurun_out/r4_t1.log
python tools/bench_train.py nrms standard naml > gpurun_out/r4_train.log 2>&1
tail -n 6 gpurun_out/r4_dbg.log; tail -n 6 gpurun_out/r4_t1.log; cut -c1-120 gpurun_out/r4_train.log | tail -n 6
