cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_step_pipeline_gpu.py -q 2>&1 | tail -3
timeout 400 python tools/step_time.py D 2>&1 | grep -E "host-driven|whole-step|device-driven"
timeout 400 python tools/step_time.py B 2>&1 | grep -E "host-driven|whole-step|device-driven"
