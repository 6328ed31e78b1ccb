"""Build libsequoia_hip.so (gfx950) in-tree with hipcc.  `python -m sequoia_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libsequoia_hip.so")
SOURCES = ["kv_ops.hip", "sampler.hip", "verify.hip", "tree_attention.hip", "fused_ops.hip", "ts_linear.hip", "allreduce.hip", "draft_block.hip"]
# measurement aids (tools/prefetch_probe.py) and measured-negative experiments (draft_fused.hip: RMSNorm inside the projection,
# profiles/r03_draft_fused_not_adopted.md) stay out of the product library: SEQUOIA_BUILD_PROBES=1 adds them
PROBES = os.environ.get("SEQUOIA_BUILD_PROBES", "0") == "1"
if PROBES:
    SOURCES = SOURCES + ["ts_probe.hip", "draft_fused.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"] + (["-DSEQUOIA_BUILD_PROBES"] if PROBES else [])
# per-source additions.  ts_linear: keep the MFMA accumulators in VGPRs -- with the AGPR form the register
# allocator permutes the 48-128 accumulator registers on every trip of the ring loop (72 v_accvgpr_* moves
# per 48 MFMAs in the ISA)
EXTRA_FLAGS = {"ts_linear.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "attn_map.h"), os.path.join(os.path.dirname(PKG), "include", "sequoia_hip.h")]
    objs = []
    cc = hipcc()
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s).replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [cc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
