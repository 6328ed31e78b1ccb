cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/j
mkdir -p $O
timeout 600 python -m pytest tests/test_ts_linear_gpu.py -q -x 2>&1 | tail -4
timeout 900 python tools/ts_tune_tp.py --tp 8 4 2 --rows 129 128 --only gate_up --out $O/ts_plans_gfx950.json --detail $O/r03_ts_linear_tuning_tp_gate_up.json > $O/tune.log 2>&1; echo rc=$?
grep "^tp" $O/tune.log | cut -c1-400
timeout 600 python tools/ts_tune_tp.py --tp 1 --arch meta-llama/Llama-2-13b-hf --rows 128 64 48 --only gate_up --out $O/ts_plans_13b.json --detail $O/r03_tuning_13b_gate_up.json > $O/tune13.log 2>&1
grep "^tp" $O/tune13.log | cut -c1-400
timeout 600 python tools/ts_tune_tp.py --tp 1 --arch meta-llama/Llama-2-7b-hf --rows 128 --only gate_up --out $O/ts_plans_7b.json > $O/tune7.log 2>&1
grep "^tp" $O/tune7.log | cut -c1-400
