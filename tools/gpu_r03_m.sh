cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3/m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_loop -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 8 --no-kernel-rooflines > $O/prof_loop.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $(find $O/prof_loop -name "*results.db" | head -1) 45 > $O/r03_bench_kernel_stats_loop_only.md; head -40 $O/r03_bench_kernel_stats_loop_only.md; tail -2 $O/prof_loop.log
find $O -name "*.db" -delete
