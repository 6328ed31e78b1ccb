# Round 4, closing session: the whole GPU suite and the driver's command on the final tree; the growmap search for B re-run.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp7
mkdir -p $O profiles
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; tail -6 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/exp7/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("B", round(d["ms_per_step"], 3), "ms/step", round(d["value"], 1), "tok/s", d["mean_accepted_len"], "roof", round(d["roofline"]["frac"], 3), d["roofline"]["traffic"], "ref", round(d["value_reference_metric"], 1))
for c, o in d["other_configs"].items():
    print(c, round(o["ms_per_step"], 3), round(o["value"], 1), o.get("mi355x_growmap"))
print("tuned B", d["mi355x_growmap"])
PY
timeout 600 python -m sequoia_amd.growmap_tuning --config B --out $O/MI355X-synthetic-68m-7b-stochastic.json > $O/tune_b.log 2> $O/tune_b.err; tail -1 $O/tune_b.log | cut -c1-900
COMMON="--steps 400 --warmup 6 --no-cpu-baseline --no-autoregressive --no-other-configs --no-reference-metric --no-tuned-growmap --no-kernel-rooflines"
timeout 300 python bench.py $COMMON --growmap MI355X-synthetic-68m-7b-stochastic 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped 32-node', round(d['value'],1), round(d['ms_per_step'],3), d['mean_accepted_len'])"
timeout 300 python bench.py $COMMON --growmap $O/MI355X-synthetic-68m-7b-stochastic.json 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('re-searched', round(d['value'],1), round(d['ms_per_step'],3), d['mean_accepted_len'])"
