cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_tp_native_gpu.py tests/test_ts_linear_gpu.py tests/test_hip_kernels.py -q > gpurun_out/r2/tests5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/tests5.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2/tests5.log | tail -20
SEQUOIA_TS_EXCLUSIVE=1 SEQUOIA_TP_FORCE_HOOKS=1 timeout 900 python bench.py --config E --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-autoregressive --sync-loop > gpurun_out/r2/bench_E_tp1.log 2>&1; tail -3 gpurun_out/r2/bench_E_tp1.log | cut -c1-1800
