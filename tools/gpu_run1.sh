set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 300 python tools/step_time.py > gpurun_out/r2/step_time.log 2>&1; tail -5 gpurun_out/r2/step_time.log
timeout 400 python -m pytest tests/test_ts_linear_gpu.py tests/test_baselines_gpu.py tests/test_step_pipeline_gpu.py -q > gpurun_out/r2/tests3.log 2>&1; tail -5 gpurun_out/r2/tests3.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_piped -o piped -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-tuned-growmap --no-autoregressive > $GRAFT_REPO_ROOT/gpurun_out/r2/prof_piped.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/r2/prof_piped | head; python tools/rocprof_summary.py $(find gpurun_out/r2/prof_piped -name "*results.db" | head -1) 45 > gpurun_out/r2/prof_piped_summary.md; head -60 gpurun_out/r2/prof_piped_summary.md; find gpurun_out/r2/prof_piped -name "*.db" -size +20M -delete
