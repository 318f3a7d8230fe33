cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_train_step.py tests/test_hip_grads.py tests/test_hip_training.py tests/test_hip_two_ranks.py -q 2>&1 | tail -n 4
echo "--- streams on"; python tools/bench_train.py nrms standard naml 2>&1 | cut -c1-90
echo "--- streams off"; XNRS_TRAIN_STREAMS=0 python tools/bench_train.py nrms standard naml 2>&1 | cut -c1-90
python tools/bench_dropout_cost.py 2>&1 | tail -n 2
