cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/i
mkdir -p $O
export SEQUOIA_BENCH_ONE_DEVICE=1 SEQUOIA_TS_EXCLUSIVE=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 -m sequoia_amd.growmap_tuning --config E --backend gloo --width 8 --accept-steps 24 --time-steps 3 --budgets 8 32 64 128 --max-depth 6 --out $O/MI355X-TP2-onegpu-7b-70b.json > $O/tune_E.log 2>&1; echo rc=$?
tail -5 $O/tune_E.log
ls -la $O
