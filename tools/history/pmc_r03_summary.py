"""profiles/r03_pmc.json from the raw rocprofv3 PMC passes of tools/pmc_r03.sh (gpurun_out/r3/pmc/*.json), stamped with the sha of the kernel sources it was measured on (bench.py refuses a stale record): per kernel
HBM bytes per launch (gfx950 correction of MI355X_MICROARCH.md §HBM: 2 x FETCH_SIZE + WRITE_SIZE, KiB), the ratio to
the algorithmic bytes, and the MFMA counters (busy cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE summed
over the 8 XCDs) -> mfma_util = busy / (GUI_ACTIVE / 8 x 1024)."""
import json
import sys

raw = {}
for p in sys.argv[2:]:
    raw.update(json.load(open(p)))


def pick(tag, kernel_sub, grid=None):
    for k, v in raw.get(tag, {}).items():
        if kernel_sub in k and (grid is None or k.endswith(f"grid={grid}")):
            return v
    raise KeyError((tag, kernel_sub, grid))


def entry(prefix, kernel_sub, grid, shape, algorithmic):
    f = pick(prefix + "_FETCH_SIZE", kernel_sub, grid)["FETCH_SIZE"]["avg"]
    w = pick(prefix + "_WRITE_SIZE", kernel_sub, grid)["WRITE_SIZE"]["avg"]
    m = pick(prefix + "_SQ_VALU_MFMA_BUSY_CYCLES", kernel_sub, grid)
    hbm = (2 * f + w) * 1024
    gui = m["GRBM_GUI_ACTIVE"]["avg"] / 8
    return dict(shape=shape, kernel=kernel_sub, FETCH_SIZE_KiB=round(f, 1), WRITE_SIZE_KiB=round(w, 1),
                algorithmic_bytes=algorithmic, hbm_bytes_per_launch=int(hbm), traffic_over_algorithmic=round(hbm / algorithmic, 3),
                SQ_VALU_MFMA_BUSY_CYCLES=m["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"], SQ_INSTS_VALU_MFMA_MOPS_F16=m["SQ_INSTS_VALU_MFMA_MOPS_F16"]["avg"],
                mfma_flop=m["SQ_INSTS_VALU_MFMA_MOPS_F16"]["avg"] * 512, GRBM_GUI_ACTIVE_per_xcd=round(gui),
                SQ_WAVE_CYCLES=m["SQ_WAVE_CYCLES"]["avg"],
                mfma_util=round(m["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / (gui * 1024), 4))


M = 128
out = dict(note=("rocprofv3 --kernel-trace --pmc <counter> (FETCH_SIZE, WRITE_SIZE and the SQ/GRBM group in SEPARATE passes) over "
                 "tools/ts_bench 128 at the shipped 128-row launch plans (weights rotate over > 640 MB) and tools/kbench.py attn. "
                 "hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads). "
                 "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); profiled passes run at a "
                 "lower clock than the timed runs."),
           kernels={})
K = out["kernels"]
K["qkv@8:128x2"] = entry("ts_qkv", "ts_linear_kernel<8, 6, 3, false>", 65536, "7B qkv 12288x4096, 128 rows, tiles 128 x splits 2 (fp32 slabs)",
                         12288 * 4096 * 2 + M * 4096 * 2 + 2 * M * 12288 * 4)
K["o@8:64x4"] = entry("ts_o", "ts_linear_kernel<8, 4, 3, false>", 65536, "7B o_proj 4096x4096, 128 rows, tiles 64 x splits 4",
                      4096 * 4096 * 2 + M * 4096 * 2 + 4 * M * 4096 * 4)
K["gate_up@8:230x1"] = entry("ts_gate_up", "ts_linear_kernel<8, 6, 3, true>", 58880, "7B gate_up 2x11008x4096 + SwiGLU, 128 rows, tiles 230",
                             2 * 11008 * 4096 * 2 + M * 4096 * 2 + M * 11008 * 2)
K["down@8:64x4"] = entry("ts_down", "ts_linear_kernel<8, 4, 3, false>", 65536, "7B down_proj 4096x11008, 128 rows, tiles 64 x splits 4",
                         4096 * 11008 * 2 + M * 11008 * 2 + 4 * M * 4096 * 4)
K["tree_attention_target7b"] = entry("attn", "tree_attention_kernel<128, 1>", 131072,
                                     "7B verify layer: H=32, q=128, kv_len=287, D=128, implicit tree mask",
                                     2 * 32 * 287 * 128 * 2 + 2 * 32 * 128 * 128 * 2)
K["tree_attention_draft68m_level"] = entry("attn", "tree_attention_kernel<64, 1>", 24576, "68m draft level: H=12, q=34, kv_len=214, D=64",
                                           2 * 12 * 214 * 64 * 2 + 2 * 12 * 34 * 64 * 2)
K["tree_attention_target70b_shard"] = entry("attn", "tree_attention_kernel<128, 1>", 36864, "70B shard (TP=8): H=8, H_kv=1, q=129, kv_len=288",
                                            2 * 1 * 288 * 128 * 2 + 2 * 8 * 129 * 128 * 2)
import hashlib
import os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sha(*names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(REPO, "sequoia_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


out["source_sha"] = {"ts_linear.hip": _sha("ts_linear.hip", "common.h"), "tree_attention.hip": _sha("tree_attention.hip", "common.h")}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in K.items():
    print(f"{k:32s} hbm {v['hbm_bytes_per_launch'] / 1e6:7.2f} MB  x{v['traffic_over_algorithmic']:.3f}  mfma_util {v['mfma_util']:.4f}")
