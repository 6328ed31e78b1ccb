# the driver's N > 1 flow with two ranks on the ONE GPU: replicas line + tensor-parallel child (gloo transport for the setup)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp8
mkdir -p $O
export PYTHONUNBUFFERED=1
SEQUOIA_BENCH_ONE_DEVICE=1 timeout 1100 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > $O/bench_replicas2.json 2> $O/bench_replicas2.err
tail -4 $O/bench_replicas2.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/exp8/bench_replicas2.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "rccl_ranks", "ms_per_step", "scaling", "mean_accepted_len")})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "step", round(d["step_roofline"]["frac"], 3))
t = d.get("tp_70b") or {}
print("tp_70b", {k: t.get(k) for k in ("value", "ms_per_step", "n_gpus", "allreduce_kind", "xgmi_status", "xgmi_self_check", "step_loop", "error", "stderr")})
PY
