# Round 4, fourth GPU session: pair tuning of the remaining small-row shapes (68m draft, 7B as config E's draft / config C's target).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp4
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 400 python tools/ts_tune_pairs.py --arch JackFram/llama-68m --layers 2 --rows 16 32 48 --names qkv o down --out $O/pairs_68m.json > $O/pairs_68m.log 2>&1; tail -9 $O/pairs_68m.log
timeout 500 python tools/ts_tune_pairs.py --rows 64 65 --names qkv o down --out $O/pairs_7b_64.json > $O/pairs_7b_64.log 2>&1; tail -6 $O/pairs_7b_64.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-autoregressive --no-tuned-growmap > $O/bench_quick.json 2> $O/bench_quick.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/exp4/bench_quick.json").read().strip().splitlines()[-1])
print("B", round(d["ms_per_step"], 3), "ms/step", round(d["value"], 1), "tok/s", "ref metric", d.get("value_reference_metric"))
for c, o in (d.get("other_configs") or {}).items():
    print(c, o.get("ms_per_step"), o.get("value"), o.get("error"))
print({k: round(v["avg_us"], 2) for k, v in d["kernels"].items()})
PY
