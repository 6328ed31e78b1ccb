cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/c
mkdir -p $O
timeout 900 python -m pytest tests/test_xgmi_allreduce_gpu.py tests/test_draft_fused_gpu.py tests/test_step_pipeline_gpu.py tests/test_hip_kernels.py -q -x --durations=8 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
grep -E "^FAILED|^ERROR|passed|failed|rc=" $O/tests.log | tail -15
tail -60 $O/tests.log
SEQUOIA_DRAFT_FUSED=1 python tools/draft_level_bench.py 1 19 34 2>&1 | grep fused
