# Round 4, first GPU session: probes + plan tuning + the GPU suite with the fail-closed trace tests.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r04_exp1.sh'
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp1
mkdir -p $O
export PYTHONUNBUFFERED=1
echo "== launch probe"; timeout 120 tools/launch_probe > $O/launch_probe.log 2>&1; cat $O/launch_probe.log
echo "== hot / cold weights (Infinity Cache)"
for spec in "gate_up 230x1" "qkv 128x2" "o+res 64x4" "down+res 64x4"; do
  set -- $spec
  echo "cold $1"; TS_ONLY=$1 TS_CANDS=$2 timeout 120 tools/ts_bench 128 2>&1 | tail -1
  echo "hot  $1"; TS_NBUF=1 TS_ONLY=$1 TS_CANDS=$2 timeout 120 tools/ts_bench 128 2>&1 | tail -1
done > $O/hot_cold.log 2>&1; cat $O/hot_cold.log
echo "== attention shapes"; timeout 300 python tools/kbench.py attn > $O/kbench_attn.log 2>&1; grep attn_ $O/kbench_attn.log
echo "== 13B plans"
timeout 600 python tools/ts_tune.py --archs meta-llama/Llama-2-13b-hf --layers 4 --only qkv gate_up --out $O/plans_13b.json --detail $O/tune_13b.json > $O/tune_13b.log 2>&1; tail -16 $O/tune_13b.log
echo "== 7B plans with 8-tile candidates"
timeout 600 python tools/ts_tune.py --archs meta-llama/Llama-2-7b-hf --layers 8 --mtp 4 5 8 --only qkv gate_up --out $O/plans_7b.json --detail $O/tune_7b.json > $O/tune_7b.log 2>&1; tail -6 $O/tune_7b.log
echo "== pairs"
timeout 400 python tools/ts_tune_pairs.py --rows 128 --out $O/pairs_7b.json > $O/pairs_7b.log 2>&1; tail -2 $O/pairs_7b.log
timeout 400 python tools/ts_tune_pairs.py --arch meta-llama/Llama-2-13b-hf --layers 4 --rows 64 --out $O/pairs_13b.json > $O/pairs_13b.log 2>&1; tail -2 $O/pairs_13b.log
echo "== GPU suite"
timeout 1200 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; tail -15 $O/tests_gpu.log
