cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/e
mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -s -k "7b" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "^FAILED|^ERROR|passed|failed|rc=|token-identical|margin" $O/tests.log | tail -25
grep -B2 -A12 "^E  " $O/tests.log | head -80
timeout 600 python bench.py --steps 100 --warmup 8 --cpu-steps 3 > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json; tail -3 $O/bench_default.err
