cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/l
mkdir -p $O
timeout 1200 python tools/ts_tune_tp.py --arch meta-llama/Llama-2-7b-hf --tp 8 4 2 --layers 16 --rows 64 1 128 --out $O/ts_plans_gfx950.json --detail $O/r03_ts_linear_tuning_tp_draft7b.json > $O/tune.log 2>&1; echo rc=$?
grep "^tp" $O/tune.log | cut -c1-100
tail -3 $O/tune.log
