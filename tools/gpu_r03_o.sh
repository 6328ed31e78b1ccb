cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3/o
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_e2e_gpu.py tests/test_step_pipeline_gpu.py tests/test_baselines_gpu.py tests/test_probe_gpu.py tests/test_properties_gpu.py tests/test_api_gpu.py tests/test_integration_stub_gpu.py -q -x 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_loop -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 8 --no-kernel-rooflines > $O/prof_loop.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $(find $O/prof_loop -name "*results.db" | head -1) 60 | grep -E "verify|sample|total kernel"; grep ms_per_step $O/prof_loop.log | cut -c1-200
find $O -name "*.db" -delete
