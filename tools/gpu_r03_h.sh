cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3/h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in qkv gate_up o; do for m in 0 1; do
PROBE_MODE=$m timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_${p}_$m -o d -- python $GRAFT_REPO_ROOT/tools/prefetch_probe.py $p 3 > $O/prof_${p}_$m.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find $O/prof_${p}_$m -name "*results.db" | head -1) 6 | grep -E "ts_linear|prefetch|rmsnorm" | cut -c1-110 | sed "s/^/$p mode $m: /"
done; done
find $O -name "*.db" -delete
