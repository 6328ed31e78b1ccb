#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
cp sequoia_amd/lib/libsequoia_hip.so /tmp/new.so
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then cp tools/_dbg/att_old/libsequoia_hip.so sequoia_amd/lib/libsequoia_hip.so; else cp /tmp/new.so sequoia_amd/lib/libsequoia_hip.so; fi
  echo "== $v" ; timeout 300 python tools/kbench.py attn 2>&1 | grep attn_
done; done | tee gpurun_out/r2/kbench_attn_ab.log
cp /tmp/new.so sequoia_amd/lib/libsequoia_hip.so
