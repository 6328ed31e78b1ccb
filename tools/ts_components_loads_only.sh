cd $GRAFT_REPO_ROOT
for bits in 0 64 65 66 67 96; do
  for spec in "qkv:128:2" "o+res:64:4" "gate_up+silu:230:1" "down+res:64:4"; do
    IFS=: read shape tiles splits <<< "$spec"
    if [ $bits = 0 ]; then pre=""; else pre="LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_dbg/$bits"; fi
    echo -n "TS_DBG=$bits $shape: "; env $pre TS_ARCH=7b TS_ONLY="$shape" TS_TILES=$tiles TS_SPLITS=$splits timeout 120 tools/ts_bench 128 2>&1 | grep "us " | head -1
  done
done
