cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/n
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/tests.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/n/bench_driver_cmd.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","mean_accepted_len","prefill_steps_in_timed_region")})
print(d["roofline"]); print(d["kernels"]["verify_stochastic"]); print({k:d["cpu_baseline"][k] for k in ("value","cores","seconds_per_step_by_threads","reference_over_port")})
PY
