"""Launch plans of the tall-skinny projections for the TENSOR-PARALLEL shards of the 70B target (configuration E) at the
verify forward's 129 rows (9 row tiles) and the 128-row prefill chunks of the exclusive weight mode: measured on one
GPU (a shard's projections are plain GEMM shapes), merged into sequoia_amd/ts_plans_gfx950.json so that every rank of a
tensor-parallel job takes the same, measured plan (Engine/ts_linear.py: shipped plans are rank-independent).

    python tools/ts_tune_tp.py [--tp 8 4 2] [--layers 6] [--detail profiles/r03_ts_linear_tuning_tp.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequoia_amd.Engine import ts_linear  # noqa: E402
from sequoia_amd.Engine.Llama_model import KNOWN_ARCHS, LlamaDims, LlamaWeights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tp", nargs="+", type=int, default=[8, 4, 2])
ap.add_argument("--arch", default="meta-llama/Llama-2-70b-hf")
ap.add_argument("--layers", type=int, default=6, help="layers of the tuning shard (weights rotate over them: > 256 MiB per projection set)")
ap.add_argument("--rows", nargs="+", type=int, default=[129, 128])
ap.add_argument("--out", default=ts_linear.PLAN_FILE)
ap.add_argument("--detail", default=None)
ap.add_argument("--only", nargs="+", default=None, help="projection names (qkv o gate_up down lm_head)")
args = ap.parse_args()

with open(ts_linear.PLAN_FILE) as f:
    shipped = json.load(f)
ts_linear._SHIPPED = {}                       # measure, do not look up
detail = {}
for tp in args.tp:
    dims = LlamaDims(vocab_size=32000, tp_world=tp, tp_rank=0, **dict(KNOWN_ARCHS[args.arch], num_hidden_layers=args.layers))
    if tp == 1:
        dims = LlamaDims(vocab_size=32000, **dict(KNOWN_ARCHS[args.arch], num_hidden_layers=args.layers))
    W = LlamaWeights.random(dims, torch.float16, "cuda:0", seed=1)
    ts = ts_linear.TsLinearSet(W, dims)
    for q in args.rows:
        for name in (args.only or ts.NAMES):
            ts.autotune(name, q)
    for key, rec in ts.tuned.items():
        choice = rec["choice"]
        if choice == "torch" and not key.startswith("32000") and "x8192@" in key or choice == "torch":
            # the exclusive weight mode cannot fall back to PyTorch's GEMM for a layer projection: keep the best kernel plan
            best = min(rec["ts_us"].items(), key=lambda kv: kv[1]) if rec["ts_us"] else None
            if best is not None and not key.split("x")[0] in ("4000", "8000", "16000"):
                choice = [int(x) for x in best[0].split("x")]
        shipped["plans"][key] = choice
        detail[key] = dict(rec, tp=tp, shipped=choice)
        print(f"tp{tp} {key:18s} -> {str(choice):12s} {rec['us']:8.1f} us (torch {rec['torch_us']:8.1f})  {rec['ts_us']}", flush=True)
    del ts, W
    torch.cuda.empty_cache()
with open(args.out, "w") as f:
    json.dump(shipped, f, indent=0, sort_keys=True)
if args.detail:
    with open(args.detail, "w") as f:
        json.dump(detail, f, indent=1, sort_keys=True)
