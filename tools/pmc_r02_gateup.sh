cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  tagp=$(echo $pass | cut -d' ' -f1)
  TS_ONLY="gate_up+silu" TS_TILES=230 TS_SPLITS=1 timeout 200 rocprofv3 --kernel-trace --pmc $pass -d $OUT/ts_gate_up_$tagp -o r -- $GRAFT_REPO_ROOT/tools/ts_bench 128 > $OUT/ts_gate_up_$tagp.log 2>&1
done
cd $GRAFT_REPO_ROOT
args=""
for d in $OUT/ts_gate_up_*/; do db=$(find $d -name "*results.db" | head -1); [ -n "$db" ] && args="$args $(basename $d)=$db"; done
python tools/pmc_summary.py $OUT/pmc_r02_gateup_raw.json $args | head -30
find $OUT -name "*.db" -delete
