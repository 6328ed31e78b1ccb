cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/g
mkdir -p $O
timeout 900 python tools/ts_tune_tp.py --tp 8 4 2 --out $O/ts_plans_gfx950.json --detail $O/r03_ts_linear_tuning_tp.json > $O/tune.log 2>&1; echo rc=$?
grep "^tp" $O/tune.log | cut -c1-110
tail -3 $O/tune.log
