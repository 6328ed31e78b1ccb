cd $GRAFT_REPO_ROOT
run() { # lib shape tiles splits
  if [ "$1" = prod ]; then pre=""; else pre="LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_dbg/$1"; fi
  echo -n "$1 $2 $3x$4: "; env $pre TS_ARCH=7b TS_ONLY="$2" TS_TILES=$3 TS_SPLITS=$4 timeout 120 tools/ts_bench 128 2>&1 | grep "us " | head -1
}
for lib in prod d2 occ2; do
  run $lib "gate_up+silu" 230 1; run $lib "gate_up+silu" 344 1; run $lib "gate_up+silu" 688 1
  run $lib qkv 128 2; run $lib qkv 192 2; run $lib qkv 192 1; run $lib qkv 256 2; run $lib qkv 384 1
  run $lib "down+res" 64 4; run $lib "down+res" 64 8; run $lib "down+res" 128 4
  run $lib "o+res" 64 4; run $lib "o+res" 64 8; run $lib "o+res" 128 4
done
