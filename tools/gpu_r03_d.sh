cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/d
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/tests.log | tail -25
grep -B2 -A25 "^E  " $O/tests.log | head -150
