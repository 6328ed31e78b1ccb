# Round-2 measurement pass on the GPU box: full GPU suite, default bench line, configs C / D short lines, rocprofv3 kernel
# statistics of the default bench command, step-time comparison.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2/final
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2/final/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/final/tests.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2/final/tests.log | tail -10
timeout 900 python bench.py > gpurun_out/r2/final/bench_default.json 2> gpurun_out/r2/final/bench_default.err; tail -c 600 gpurun_out/r2/final/bench_default.json
timeout 300 python tools/step_time.py > gpurun_out/r2/final/step_time.log 2>&1; tail -3 gpurun_out/r2/final/step_time.log
for c in C D; do timeout 600 python bench.py --config $c --steps 60 --warmup 5 --no-cpu-baseline --no-autoregressive > gpurun_out/r2/final/bench_config$c.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/final/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-tuned-growmap --no-autoregressive > $GRAFT_REPO_ROOT/gpurun_out/r2/final/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $(find gpurun_out/r2/final/prof -name "*results.db" | head -1) 45 > gpurun_out/r2/final/kernel_stats.md; find gpurun_out/r2/final/prof -name "*.db" -delete; head -30 gpurun_out/r2/final/kernel_stats.md
