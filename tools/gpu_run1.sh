cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_samp -o s -- python $GRAFT_REPO_ROOT/tools/kbench.py samp > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $(find gpurun_out/r2/prof_samp -name "*results.db" | head -1) 12; find gpurun_out/r2/prof_samp -name "*.db" -delete
