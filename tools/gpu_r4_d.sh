cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- packet capture off + NO cache workarounds"
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 XNRS_HIP_LIB=$GRAFT_REPO_ROOT/xnrs_amd/libxnrs_hip_nowa.so python tools/debug_graph_step.py 8 2>&1 | tail -n 4 | cut -c1-200
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 XNRS_HIP_LIB=$GRAFT_REPO_ROOT/xnrs_amd/libxnrs_hip_nowa.so NOCLEAR=1 python tools/debug_graph_step.py 8 2>&1 | tail -n 4 | cut -c1-200
echo "--- full gpu suite (conftest sets packet capture off)"
python -m pytest tests -m gpu -q 2>&1 | tail -n 12 > gpurun_out/r4_gpu_suite.log; tail -n 3 gpurun_out/r4_gpu_suite.log
for i in 1 2 3; do python -m pytest tests/test_hip_train_step.py -q 2>&1 | tail -n 1; done
