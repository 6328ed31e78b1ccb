# Round 4, third GPU session: pair tuning of the draft-size projections, the new C_7b fixture, a clean config D loop profile.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp3
mkdir -p $O
export PYTHONUNBUFFERED=1
echo "== new fixture / kernels"
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_step_pipeline_gpu.py tests/test_hip_kernels.py -m gpu -q -k "C_7b or kv_compact or top_p or B_topp09" > $O/tests_subset.log 2>&1; tail -8 $O/tests_subset.log
echo "== pairs: 1.3B draft"
timeout 500 python tools/ts_tune_pairs.py --arch princeton-nlp/Sheared-LLaMA-1.3B --layers 24 --rows 16 32 --names qkv o down --out $O/pairs_1p3b.json > $O/pairs_1p3b.log 2>&1; tail -6 $O/pairs_1p3b.log
echo "== pairs: 13B qkv @64, 7B qkv @128"
timeout 300 python tools/ts_tune_pairs.py --arch meta-llama/Llama-2-13b-hf --layers 4 --rows 64 --names qkv --out $O/pairs_13b_qkv.json > $O/pairs_13b_qkv.log 2>&1; tail -1 $O/pairs_13b_qkv.log
timeout 300 python tools/ts_tune_pairs.py --rows 128 --names qkv --out $O/pairs_7b_qkv.json > $O/pairs_7b_qkv.log 2>&1; tail -1 $O/pairs_7b_qkv.log
echo "== config D loop profile (no GEMM tuning in the trace)"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_d -o b -- python $GRAFT_REPO_ROOT/bench.py --config D --steps 60 --warmup 6 --no-gemm-tuning --no-kernel-rooflines --no-cpu-baseline --no-tuned-growmap --no-autoregressive --no-other-configs --no-reference-metric > $O/prof_d.log 2>&1)
python tools/rocprof_summary.py $(find $O/prof_d -name "*results.db" | head -1) 45 > $O/kernel_stats_configD_loop.md; find $O/prof_d -name "*.db" -delete
head -52 $O/kernel_stats_configD_loop.md; tail -2 $O/prof_d.log
