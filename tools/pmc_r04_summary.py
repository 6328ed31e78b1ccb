"""profiles/r04_pmc.json from the raw rocprofv3 PMC passes of tools/pmc_r04.sh (gpurun_out/r04/pmc/*.json), stamped with the sha of the kernel sources it was measured on (bench.py refuses a stale record): per kernel
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
                 "tools/ts_bench at the shipped launch plans (7B shapes at 128 rows, 13B shapes -- keys D:* -- at 64 rows; weights "
                 "rotate over > 640 MB) and tools/kbench.py attn. "
                 "hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads). "
                 "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); profiled passes run at a "
                 "lower clock than the timed runs."),
           kernels={})
K = out["kernels"]


def add(key, *a):
    try:
        K[key] = entry(*a)
    except KeyError as e:
        print(f"{key}: no record ({e})")


add("qkv@8:128x2", "ts_qkv", "ts_linear_kernel<8, 6, 3, false>", 65536, "7B qkv 12288x4096, 128 rows, tiles 128 x splits 2 (fp32 slabs)",
    12288 * 4096 * 2 + M * 4096 * 2 + 2 * M * 12288 * 4)
add("o@8:64x4", "ts_o", "ts_linear_kernel<8, 4, 3, false>", 65536, "7B o_proj 4096x4096, 128 rows, tiles 64 x splits 4",
    4096 * 4096 * 2 + M * 4096 * 2 + 4 * M * 4096 * 4)
add("gate_up@8:230x1", "ts_gate_up", "ts_linear_kernel<8, 6, 3, true>", 58880, "7B gate_up 2x11008x4096 + SwiGLU, 128 rows, tiles 230",
    2 * 11008 * 4096 * 2 + M * 4096 * 2 + M * 11008 * 2)
add("down@8:64x4", "ts_down", "ts_linear_kernel<8, 4, 3, false>", 65536, "7B down_proj 4096x11008, 128 rows, tiles 64 x splits 4",
    4096 * 11008 * 2 + M * 11008 * 2 + 4 * M * 4096 * 4)
# configuration D: Llama-2-13b projections at the 64 rows of its verify forward (A100-CNN-160m-13b growmap, 64 nodes)
R = 64
add("D:qkv@4:120x2", "ts_d_qkv", "ts_linear_kernel<4, 8, 3, false>", 61440, "13B qkv 15360x5120, 64 rows, tiles 120 x splits 2 (8 column tiles / workgroup)",
    15360 * 5120 * 2 + R * 5120 * 2 + 2 * R * 15360 * 4)
add("D:o@4:80x3", "ts_d_o", "ts_linear_kernel<4, 4, 4, false>", 61440, "13B o_proj 5120x5120, 64 rows, tiles 80 x splits 3 (pair-tuned with its norm)",
    5120 * 5120 * 2 + R * 5120 * 2 + 3 * R * 5120 * 4)
add("D:gate_up@4:216x1", "ts_d_gate_up", "ts_linear_kernel<4, 8, 3, true>", 55296, "13B gate_up 2x13824x5120 + SwiGLU, 64 rows, tiles 216 (4 gate+up units / workgroup)",
    2 * 13824 * 5120 * 2 + R * 5120 * 2 + R * 13824 * 2)
add("D:down@4:80x3", "ts_d_down", "ts_linear_kernel<4, 4, 4, false>", 61440, "13B down_proj 5120x13824, 64 rows, tiles 80 x splits 3 (pair-tuned with its norm)",
    5120 * 13824 * 2 + R * 13824 * 2 + 3 * R * 5120 * 4)
# configuration E at TP = 1 (round 5): full-width Llama-2-70b projections at the 129 rows of the 64x2 tree
R9 = 129
add("E:gate_up@9:448x1", "ts_e_gate_up", "ts_linear_tail_kernel<8, 8, 2, true>", 114688, "70B gate_up 2x28672x8192 + SwiGLU, 129 rows, tiles 448 (8 MFMA row tiles + the extra row on the vector ALU, 4 gate+up units / workgroup)",
    2 * 28672 * 8192 * 2 + R9 * 8192 * 2 + R9 * 28672 * 2)
add("E:down@9:128x4", "ts_e_down", "ts_linear_tail_kernel<8, 4, 3, false>", 131072, "70B down_proj 8192x28672, 129 rows, tiles 128 x splits 4",
    8192 * 28672 * 2 + R9 * 28672 * 2 + 4 * R9 * 8192 * 4)
add("tree_attention_target7b", "attn", "tree_attention_kernel<128, 1>", 131072, "7B verify layer: H=32, q=128, kv_len=287, D=128, implicit tree mask",
    2 * 32 * 287 * 128 * 2 + 2 * 32 * 128 * 128 * 2)
add("tree_attention_draft68m_level", "attn", "tree_attention_kernel<64, 1>", 24576, "68m draft level: H=12, q=34, kv_len=214, D=64",
    2 * 12 * 214 * 64 * 2 + 2 * 12 * 34 * 64 * 2)
# 70B shard (TP = 8): 72 (query head, tile) items of the ONE KV head on 3 XCDs (round 4; round 3 dealt them over all 8)
add("tree_attention_target70b_shard", "attn", "tree_attention_kernel<128, 1>", 98304, "70B shard (TP=8): H=8, H_kv=1, q=129, kv_len=288, XCD span 3",
    2 * 1 * 288 * 128 * 2 + 2 * 8 * 129 * 128 * 2)
add("tree_attention_target70b_shard_tp2", "attn", "tree_attention_kernel<128, 1>", 147456, "70B shard (TP=2): H=32, H_kv=4, q=129, kv_len=288, XCD span 2",
    2 * 4 * 288 * 128 * 2 + 2 * 32 * 129 * 128 * 2)
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
