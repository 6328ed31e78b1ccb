cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3/sc1
mkdir -p $O
L=sequoia_amd/lib
cp $L/libsequoia_hip.so $L/base.so
run() {
  timeout 600 python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-autoregressive --no-tuned-growmap --no-kernel-rooflines > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
print("$1", d["ms_per_step"], d["value"], d["mean_accepted_len"])
PY
}
run A1
cp $L/exp_sc1.so $L/libsequoia_hip.so; run B1
cp $L/base.so $L/libsequoia_hip.so; run A2
cp $L/exp_sc1.so $L/libsequoia_hip.so; run B2
timeout 300 python -m pytest tests/test_ts_linear_gpu.py -x -q 2>&1 | tail -2
cp $L/base.so $L/libsequoia_hip.so
