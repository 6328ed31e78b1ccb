cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_step_pipeline_gpu.py -q -x -s -k "D_13b_w4" 2>&1 | grep -E "passed|failed|token-identical|Error|assert" | tail -8
