cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/k
mkdir -p $O
timeout 600 python -m pytest tests/test_xgmi_allreduce_gpu.py -q -x -k "whole_step" 2>&1 | tail -6
SEQUOIA_BENCH_ONE_DEVICE=1 SEQUOIA_TS_EXCLUSIVE=1 timeout 900 python bench.py --gpus 2 --config E --backend gloo --steps 8 --warmup 2 --no-cpu-baseline --no-autoregressive --no-tuned-growmap --no-tp-extra > $O/bench_E_tp2_tpdraft.json 2> $O/bench_E_tp2_tpdraft.err; echo "rc=$?"; tail -4 $O/bench_E_tp2_tpdraft.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/k/bench_E_tp2_tpdraft.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","mean_accepted_len","allreduce")}, d["config"]["step_loop"])
PY
