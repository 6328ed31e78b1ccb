cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/f
mkdir -p $O
bash tools/pmc_r03.sh > $O/pmc.log 2>&1; tail -12 $O/pmc.log
cd $GRAFT_REPO_ROOT
SEQUOIA_BENCH_ONE_DEVICE=1 SEQUOIA_TS_EXCLUSIVE=1 timeout 900 python bench.py --gpus 2 --config E --backend gloo --steps 8 --warmup 2 --no-cpu-baseline --no-autoregressive --no-tuned-growmap --no-tp-extra > $O/bench_E_tp2.json 2> $O/bench_E_tp2.err; echo "rc=$?"; tail -c 2500 $O/bench_E_tp2.json; tail -8 $O/bench_E_tp2.err
