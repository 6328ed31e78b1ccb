cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/b
mkdir -p $O
timeout 300 python -m pytest tests/test_draft_fused_gpu.py -q -x 2>&1 | tail -5
python tools/draft_level_bench.py 1 19 34 > $O/draft_level.log 2>&1; cat $O/draft_level.log | grep fused
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o d -- python $GRAFT_REPO_ROOT/tools/draft_level_bench.py 34 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $(find $O/prof -name "*results.db" | head -1) 14 > $O/draft_kernel_stats.md; find $O/prof -name "*.db" -delete; cat $O/draft_kernel_stats.md
