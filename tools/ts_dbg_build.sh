#!/bin/bash
# Builds experiment variants of the library (ts_linear.hip compiled with -DTS_DBG=<bits>) into tools/_dbg/<bits>/ ;
# (some switch combinations crash the compiler's AGPR rewrite pass on the widest builds)
# run them with  LD_LIBRARY_PATH=tools/_dbg/<bits> tools/ts_bench ...   (the other objects are the production ones)
set -e
cd "$(dirname "$0")/.."
L=sequoia_amd/lib
for bits in "$@"; do
  mkdir -p tools/_dbg/$bits
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form \
      -DTS_DBG=$bits -c sequoia_amd/csrc/ts_linear.hip -o tools/_dbg/$bits/ts_linear.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_dbg/$bits/libsequoia_hip.so $L/kv_ops.o $L/sampler.o $L/verify.o \
      $L/tree_attention.o $L/fused_ops.o tools/_dbg/$bits/ts_linear.o
  rm tools/_dbg/$bits/*.o
done
