cd $GRAFT_REPO_ROOT
timeout 200 python tools/kbench.py verify 2>&1 | tail -5
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_e2e_gpu.py tests/test_properties_gpu.py tests/test_probe_gpu.py -q 2>&1 | tail -3
