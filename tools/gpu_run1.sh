cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2/tests6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/tests6.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2/tests6.log | tail -20
