cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_step_pipeline_gpu.py tests/test_xgmi_allreduce_gpu.py tests/test_tp_world2_gpu.py -q -x -s -k "E_70b_w2 or world2 or one_gpu_matches" 2>&1 | grep -E "passed|failed|token-identical|Error|assert|rank" | tail -12
