# Round 4, second GPU session: the GPU suite on the new fixtures / kernels, the new bench line, where config D's step goes.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp2
mkdir -p $O
export PYTHONUNBUFFERED=1
echo "== GPU suite"
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; tail -25 $O/tests_gpu.log
echo "== bench (driver command)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 6000 $O/bench_driver_cmd.json; tail -5 $O/bench_driver_cmd.err
echo "== config D loop profile"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_d -o b -- python $GRAFT_REPO_ROOT/bench.py --config D --steps 60 --warmup 6 --no-kernel-rooflines --no-cpu-baseline --no-tuned-growmap --no-autoregressive --no-other-configs --no-reference-metric > $O/prof_d.log 2>&1)
python tools/rocprof_summary.py $(find $O/prof_d -name "*results.db" | head -1) 40 > $O/kernel_stats_configD_loop.md; find $O/prof_d -name "*.db" -delete
head -40 $O/kernel_stats_configD_loop.md; tail -2 $O/prof_d.log
