"""Measure the launch plans of the tall-skinny projections (sequoia_amd/Engine/ts_linear.py) for the bundled
architectures on this GPU and write sequoia_amd/ts_plans_gfx950.json (plan key -> "torch" | [tiles, splits]).

    python tools/ts_tune.py [--archs JackFram/llama-68m meta-llama/Llama-2-7b-hf ...] [--out path]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequoia_amd.Engine import ts_linear  # noqa: E402
from sequoia_amd.Engine.Llama_model import load_weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--archs", nargs="+", default=["JackFram/llama-68m", "meta-llama/Llama-2-7b-hf",
                                               "princeton-nlp/Sheared-LLaMA-1.3B", "meta-llama/Llama-2-13b-hf"])
ap.add_argument("--out", default=ts_linear.PLAN_FILE)
ap.add_argument("--detail", default=None, help="also dump every timing to this JSON")
ap.add_argument("--merge", action="store_true", help="update the plans in --out instead of replacing the file")
ap.add_argument("--mtp", nargs="+", type=int, default=list(range(1, 9)), help="row-tile counts to tune (rows = 16 x mtp)")
ap.add_argument("--only", nargs="+", default=None, help="projection names (qkv o gate_up down lm_head)")
ap.add_argument("--layers", type=int, default=None, help="build the tuning model with this many layers (weights rotate over them)")
args = ap.parse_args()

ts_linear._SHIPPED = {}                       # measure everything
plans, detail = {}, {}
for arch in args.archs:
    if args.layers:
        from sequoia_amd.Engine.Llama_model import KNOWN_ARCHS, LlamaDims, LlamaWeights
        dims = LlamaDims(vocab_size=32000, **dict(KNOWN_ARCHS[arch], num_hidden_layers=args.layers))
        W = LlamaWeights.random(dims, torch.float16, "cuda:0", seed=1)
    else:
        W = load_weights(f"random:{arch}:seed=1", torch.float16, "cuda:0")
    ts = ts_linear.TsLinearSet(W, W.dims)
    for mtp in args.mtp:
        q = 16 * mtp
        for name in (args.only or ts.NAMES):
            ts.autotune(name, q)
    for key, rec in ts.tuned.items():
        plans[key] = rec["choice"]
        detail[key] = dict(rec, arch=arch)
        print(f"{arch:34s} {key:18s} -> {str(rec['choice']):12s} {rec['us']:8.1f} us (torch {rec['torch_us']:8.1f})", flush=True)
    del ts, W
    torch.cuda.empty_cache()
if args.merge and os.path.exists(args.out):
    with open(args.out) as f:
        old = json.load(f)
    plans = dict(old.get("plans", {}), **plans)
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
with open(args.out, "w") as f:
    json.dump({"device": torch.cuda.get_device_name(0), "plans": plans}, f, indent=0, sort_keys=True)
if args.detail:
    with open(args.detail, "w") as f:
        json.dump(detail, f, indent=1, sort_keys=True)
