# PMC passes of round 2 (run on the GPU box through gpurun): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and
# MFMA counters of the projection kernel at the shipped 128-row plans and of the tree-attention kernel.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -iE "MFMA|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES" | head -40 > $OUT/counters_available.txt
run_ts() {   # tag shape tiles splits
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
    tagp=$(echo $pass | cut -d' ' -f1)
    TS_ONLY="$2" TS_TILES=$3 TS_SPLITS=$4 timeout 200 rocprofv3 --kernel-trace --pmc $pass -d $OUT/ts_$1_$tagp -o r -- $GRAFT_REPO_ROOT/tools/ts_bench 128 > $OUT/ts_$1_$tagp.log 2>&1
  done
}
run_ts qkv qkv 128 2
run_ts o "o+res" 64 4
run_ts gate_up "gate_up+silu" 230 1
run_ts down "down+res" 64 4
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  tagp=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT/attn_$tagp -o r -- python $GRAFT_REPO_ROOT/tools/kbench.py attn > $OUT/attn_$tagp.log 2>&1
done
cd $GRAFT_REPO_ROOT
args=""
for d in $OUT/*/; do db=$(find $d -name "*results.db" | head -1); [ -n "$db" ] && args="$args $(basename $d)=$db"; done
python tools/pmc_summary.py $OUT/pmc_r02_raw.json $args > /dev/null
find $OUT -name "*.db" -delete
ls -la $OUT | head -40; cat $OUT/counters_available.txt | head -20
