# Round 4: the growmap search for configuration D on this round's kernels, and both growmaps timed side by side.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp5
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_ts_linear_gpu.py -m gpu -q -k "autotune" > $O/tests_autotune.log 2>&1; tail -2 $O/tests_autotune.log
timeout 900 python -m sequoia_amd.growmap_tuning --config D --out $O/MI355X-synthetic-1.3b-13b-stochastic.json > $O/tune_d.log 2> $O/tune_d.err; tail -2 $O/tune_d.log | cut -c1-1500; tail -3 $O/tune_d.err | cut -c1-300
COMMON="--config D --steps 200 --warmup 6 --no-cpu-baseline --no-autoregressive --no-other-configs --no-reference-metric --no-tuned-growmap"
timeout 600 python bench.py $COMMON > $O/bench_D_reference_growmap.json 2> $O/bench_D_ref.err
timeout 600 python bench.py $COMMON --growmap $O/MI355X-synthetic-1.3b-13b-stochastic.json > $O/bench_D_mi355x_growmap.json 2> $O/bench_D_tuned.err
python - <<'PY'
import json
for f in ("bench_D_reference_growmap", "bench_D_mi355x_growmap"):
    try:
        d = json.loads(open(f"gpurun_out/r04/exp5/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "tok/s", round(d["ms_per_step"], 3), "ms/step", round(d["mean_accepted_len"], 3), d["config"]["workload"][-90:])
    except Exception as e:
        print(f, "failed", e)
PY
