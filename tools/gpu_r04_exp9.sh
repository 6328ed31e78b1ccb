# measured logit distance of every committed trace on the GPU (printed by the pytest summary): evidence for the tolerances
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp9
mkdir -p $O
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_baselines_gpu.py -m gpu -q -k "reproduces_reference_tokens or follows_reference" > $O/tests_logit_excess.log 2>&1; grep -A30 "logit distance" $O/tests_logit_excess.log | cut -c1-200; tail -2 $O/tests_logit_excess.log
