# Round-3 measurement pass on the GPU box: default bench line (the driver's command and the 200-step default), configs C / D
# short lines, rocprofv3 kernel statistics of the default bench command and of the loop alone, step-time comparison.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3/final
mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 300 python tools/step_time.py > $O/step_time.log 2>&1; tail -3 $O/step_time.log
for c in C D; do timeout 600 python bench.py --config $c --steps 60 --warmup 5 --no-cpu-baseline --no-autoregressive > $O/bench_config$c.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-tuned-growmap --no-autoregressive > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $(find $O/prof -name "*results.db" | head -1) 45 > $O/kernel_stats.md; find $O/prof -name "*.db" -delete; head -14 $O/kernel_stats.md
python - <<'PY'
import json
for n in ("bench_default","bench_driver_cmd","bench_configC","bench_configD"):
    try:
        d=json.loads(open(f"gpurun_out/r3/final/{n}.json").read().strip().splitlines()[-1])
        print(n, {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("value","ms_per_step","mean_accepted_len")}, "host", (d.get("host_driven_loop") or {}).get("ms_per_step"), "roof", round(d["roofline"]["frac"],3), d["roofline"]["traffic"])
    except Exception as e:
        print(n, "failed", e)
PY
