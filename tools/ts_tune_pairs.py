"""Time row-parallel projections TOGETHER with the kernel that consumes their output.

o_proj / down_proj end in a residual add + RMSNorm.  A split-K plan hands the norm `splits` fp32 slabs to sum
(8.4 MB at 128 x 4096 x 4 splits, written and read back), a plan without a K split can write the residual stream in
the projection's own "+ residual" epilogue and leave a norm-only launch -- fewer bytes, but every workgroup then
reads the whole activation image.  The autotuner times the projection alone; this tool times the pair, per layer,
under hipGraph replay with the weights rotating over the layers (nothing stays in the 256 MiB Infinity Cache):

    python tools/ts_tune_pairs.py [--arch meta-llama/Llama-2-7b-hf] [--rows 128] [--layers 8] [--out gpurun_out/pairs.json]

Forms per (tiles, splits):
    slab     splits > 1 : linear_ts -> slabs ; add_rmsnorm_slabs (sum, + x, norm -> fragment-major operand)
    rows     splits == 1: linear_ts -> rows  ; add_rmsnorm_frag  (+ x, norm)
    inplace  splits == 1: linear_ts(+ residual epilogue, x in place) ; rmsnorm_frag (norm only)
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequoia_amd.Engine import ts_linear  # noqa: E402
from sequoia_amd.Engine.Llama_model import KNOWN_ARCHS, LlamaDims, LlamaWeights  # noqa: E402
from sequoia_amd.ops import get_ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="meta-llama/Llama-2-7b-hf")
ap.add_argument("--rows", nargs="+", type=int, default=[128])
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--names", nargs="+", default=["o", "down"], help="o down qkv")
ap.add_argument("--out", default=None)
args = ap.parse_args()

dev = "cuda:0"
ops = get_ops()
dims = LlamaDims(vocab_size=32000, **dict(KNOWN_ARCHS[args.arch], num_hidden_layers=args.layers))
W = LlamaWeights.random(dims, torch.float16, dev, seed=1)
ts = ts_linear.TsLinearSet(W, dims)
eps = dims.rms_norm_eps
hidden = dims.hidden_size
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, n_layers, reps=16):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for i in range(2):
            fn(i % n_layers)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(reps):
                fn(i % n_layers)
        g.replay()
        best = 1e9
        for _ in range(3):
            e0.record(side)
            g.replay()
            e1.record(side)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    torch.cuda.current_stream(dev).wait_stream(side)
    return best


report = {}


def tune_qkv(q):
    """qkv + the RoPE / KV-write launch that consumes it: split-K slabs (sq_rope_kv_write_slabs_f16 sums them) against fp16
    rows (sq_rope_kv_write_f16)."""
    n_out, k, _ = ts.shapes["qkv"]
    H, Hkv, D, M = dims.local_heads, dims.local_kv_heads, dims.head_dim, 384
    for li in range(args.layers):
        ts.frag("qkv", li)
    a = ops.repack_rows((torch.randn(q, k, device=dev) * 0.5).half())
    rows = torch.empty((q, n_out), dtype=torch.float16, device=dev)
    q_out = torch.empty((H, q, D), dtype=torch.float16, device=dev)
    kc = torch.zeros((Hkv, M, D), dtype=torch.float16, device=dev); vc = torch.zeros_like(kc)
    cos = torch.randn(2048, D, device=dev).half()
    pos = torch.arange(130, 130 + q, device=dev)
    res = {}
    for tiles, splits in sorted(set(ts_linear.candidates(n_out, k, False, q, allow_split=True))):
        if splits > 1:
            def pair(li, tiles=tiles, splits=splits):
                ops.linear_ts(a, ts.frag("qkv", li), q, n_out, k, tiles=tiles, splits=splits, slab=ts._slab)
                ops.rope_kv_write_slabs(ts._slab, splits, n_out, q_out, kc, vc, cos, cos, pos, pos, H, Hkv, D)
            res[f"{tiles}x{splits}:slab"] = (None, round(timeit(pair, args.layers), 2))
        else:
            def pair(li, tiles=tiles):
                ops.linear_ts(a, ts.frag("qkv", li), q, n_out, k, out=rows, tiles=tiles, splits=1)
                ops.rope_kv_write(rows, q_out, kc, vc, cos, cos, pos, pos, H, Hkv, D)
            res[f"{tiles}x1:rows"] = (None, round(timeit(pair, args.layers), 2))
    order = sorted(res.items(), key=lambda kv: kv[1][1])
    key = f"qkv:{n_out}x{k}@{q}"
    report[key] = dict(best=order[0][0], best_pair_us=order[0][1][1], all={k_: v for k_, v in order})
    print(key, "best pair (projection + RoPE / KV write):", order[:6], flush=True)


for q in args.rows:
    for name in args.names:
        if name == "qkv":
            tune_qkv(q)
            continue
        n_out, k, _ = ts.shapes[name]
        assert n_out == hidden
        for li in range(args.layers):
            ts.frag(name, li)
        a = ops.repack_rows((torch.randn(q, k, device=dev) * 0.5).half())
        x = (torch.randn(q, hidden, device=dev) * 0.5).half()
        rows = torch.empty((q, hidden), dtype=torch.float16, device=dev)
        nxt = torch.empty(ops.frag_shape(q, hidden), dtype=torch.float16, device=dev)
        wn = W.layers[0].ln2
        units = n_out // 16
        res = {}
        # the projection alone, for reference, then the pair
        cands = ts_linear.candidates(n_out, k, False, q, allow_split=True)
        cands += [(t, s) for t in (units, units // 2) for s in (1,) if (t, s) not in cands]
        for tiles, splits in sorted(set(cands)):
            if splits > 1:
                def alone(li, tiles=tiles, splits=splits):
                    ops.linear_ts(a, ts.frag(name, li), q, n_out, k, tiles=tiles, splits=splits, slab=ts._slab)

                def pair(li, tiles=tiles, splits=splits):
                    ops.linear_ts(a, ts.frag(name, li), q, n_out, k, tiles=tiles, splits=splits, slab=ts._slab)
                    ops.add_rmsnorm_slabs(ts._slab, splits, x, x, wn, nxt, eps, out_frag=True)
                res[f"{tiles}x{splits}:slab"] = (round(timeit(alone, args.layers), 2), round(timeit(pair, args.layers), 2))
            else:
                def alone(li, tiles=tiles):
                    ops.linear_ts(a, ts.frag(name, li), q, n_out, k, out=rows, tiles=tiles, splits=1)

                def pair_rows(li, tiles=tiles):
                    ops.linear_ts(a, ts.frag(name, li), q, n_out, k, out=rows, tiles=tiles, splits=1)
                    ops.add_rmsnorm_frag(rows, x, x, wn, nxt, eps)

                def pair_inplace(li, tiles=tiles):
                    ops.linear_ts(a, ts.frag(name, li), q, n_out, k, out=x, res=x, tiles=tiles, splits=1)
                    ops.rmsnorm_frag(x, wn, nxt, eps)
                t_alone = round(timeit(alone, args.layers), 2)
                res[f"{tiles}x1:rows"] = (t_alone, round(timeit(pair_rows, args.layers), 2))
                res[f"{tiles}x1:inplace"] = (t_alone, round(timeit(pair_inplace, args.layers), 2))
        order = sorted(res.items(), key=lambda kv: kv[1][1])
        key = f"{name}:{n_out}x{k}@{q}"
        report[key] = dict(best=order[0][0], best_pair_us=order[0][1][1], all={k_: v for k_, v in order})
        print(key, "best pair:", order[:6], flush=True)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
