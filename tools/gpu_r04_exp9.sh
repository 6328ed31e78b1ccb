# the 8-step B_7b fixture on the GPU: host-driven replay, whole-step graphs
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp9
mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_step_pipeline_gpu.py -m gpu -q -k "B_7b" > $O/tests_b7b.log 2>&1; tail -6 $O/tests_b7b.log | cut -c1-400
