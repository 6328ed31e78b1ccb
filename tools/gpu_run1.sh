cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2/tests4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/tests4.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2/tests4.log | tail -20
timeout 600 python bench.py --steps 200 --warmup 8 --no-cpu-baseline > gpurun_out/r2/bench_v2.log 2>&1; tail -1 gpurun_out/r2/bench_v2.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','mean_accepted_len')}, d['host_driven_loop'], d['mi355x_growmap'], d['autoregressive_baseline'])
print(d['roofline']); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
