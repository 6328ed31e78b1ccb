# the whole GPU suite + smoke on the round's last tree
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04/exp10
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; tail -26 $O/tests_gpu.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
