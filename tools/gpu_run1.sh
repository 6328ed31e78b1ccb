#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
OUT=gpurun_out/r2/ts_merge1.log
: > $OUT
run() { TS_ONLY="$1" TS_CANDS="$2" timeout 120 tools/ts_bench $3 2>&1 | grep "M=" | sed 's/max_err [0-9.e+-]* //' >> $OUT; }
run qkv "256x1,128x2,96x2" 128; run "o+res" "64x4,32x8" 128; run "gate_up+silu" "230x1,256x1" 128; run "down+res" "64x4,32x8" 128; run lm_head "500x1" 128
run qkv "256x1,128x2" "16 48 64 100 129 144"; run "gate_up+silu" "230x1" "16 48 64 100 129 144"; run "down+res" "64x4" "16 48 64 129"
cat $OUT
timeout 900 python -m pytest tests/test_ts_linear_gpu.py -x -q 2>&1 | tail -5
