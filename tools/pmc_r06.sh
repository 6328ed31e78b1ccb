# PMC passes of round 6 (the round-4 recipe on the round-6 kernel sources) (run on the GPU box through gpurun): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA
# counters of the projection kernel at the shipped plans -- 7B at 128 rows (configuration B) AND 13B at 64 rows (configuration
# D: the 4-unit fused SwiGLU plan, the 8-tile qkv plan, the pair-tuned o / down plans) -- and of the tree-attention kernel
# (incl. the 70B shard with the XCD span of round 4).  bench.py looks the dominant kernel's record up by launch plan.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06/pmc_ts
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE")
run_ts() {   # tag arch rows shape tiles splits
  for pass in "${PASSES[@]}"; do
    tagp=$(echo $pass | cut -d' ' -f1)
    TS_ARCH=$2 TS_ONLY="$4" TS_TILES=$5 TS_SPLITS=$6 timeout 200 rocprofv3 --kernel-trace --pmc $pass -d $OUT/ts_$1_$tagp -o r -- $GRAFT_REPO_ROOT/tools/ts_bench $3 > $OUT/ts_$1_$tagp.log 2>&1
  done
}
run_ts qkv 7b 128 qkv 128 2
run_ts o 7b 128 "o+res" 64 4
run_ts gate_up 7b 128 "gate_up+silu" 230 1
run_ts down 7b 128 "down+res" 64 4
run_ts d_qkv 13b 64 qkv 120 2
run_ts d_o 13b 64 "o+res" 80 3
run_ts d_gate_up 13b 64 "gate_up+silu" 216 1
run_ts d_down 13b 64 "down+res" 80 3
# configuration E at TP = 1: the full-width 70B gate_up at 129 rows on the 8-tile + extra-row build (4 units per workgroup)
run_ts e_gate_up 70b 129 "gate_up+silu" 448 1
run_ts e_down 70b 129 "down+res" 128 4
for pass in "${PASSES[@]}"; do
  tagp=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT/attn_$tagp -o r -- python $GRAFT_REPO_ROOT/tools/kbench.py attn > $OUT/attn_$tagp.log 2>&1
done
cd $GRAFT_REPO_ROOT
args=""
for d in $OUT/*/; do db=$(find $d -name "*results.db" | head -1); [ -n "$db" ] && args="$args $(basename $d)=$db"; done
python tools/pmc_summary.py $OUT/pmc_r06_raw.json $args > /dev/null
find $OUT -name "*.db" -delete
python tools/pmc_r04_summary.py $OUT/r06_pmc.json $OUT/pmc_r06_raw.json
