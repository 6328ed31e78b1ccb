#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
OUT=gpurun_out/r2/ts_cs6.log
: > $OUT
run() { SQ_TS_VARIANT=$1 TS_ONLY="$2" TS_CANDS="$3" timeout 120 tools/ts_bench 128 2>&1 | grep "M=128" | sed 's/max_err [0-9.e+-]* //' >> $OUT; }
echo "== variant 0 (K-split waves)" >> $OUT
run 0 qkv "128x2"; run 0 "o+res" "64x4"; run 0 "gate_up+silu" "230x1"; run 0 "down+res" "64x4"
echo "== variant 1 (column-split waves)" >> $OUT
run 1 qkv "64x4,64x3,96x2,48x5,128x2"
run 1 "o+res" "64x4,32x8,32x6,64x3"
run 1 "gate_up+silu" "230x1,172x1"
run 1 "down+res" "64x4,32x8,32x6,16x16"
for dbg in 8 40; do
echo "== variant 1 TS_DBG=$dbg" >> $OUT
export LD_LIBRARY_PATH=tools/_dbg/$dbg
run 1 qkv "64x4,128x2"; run 1 "o+res" "64x4"; run 1 "gate_up+silu" "172x1,230x1"; run 1 "down+res" "64x4,32x8"
done
cat $OUT
