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
args = ap.parse_args()

ts_linear._SHIPPED = {}                       # measure everything
plans, detail = {}, {}
for arch in args.archs:
    W = load_weights(f"random:{arch}:seed=1", torch.float16, "cuda:0")
    ts = ts_linear.TsLinearSet(W, W.dims)
    for mtp in range(1, 9):
        q = 16 * mtp
        for name in ts.NAMES:
            ts.autotune(name, q)
    for key, rec in ts.tuned.items():
        plans[key] = rec["choice"]
        detail[key] = dict(rec, arch=arch)
        print(f"{arch:34s} {key:18s} -> {str(rec['choice']):12s} {rec['us']:8.1f} us (torch {rec['torch_us']:8.1f})", flush=True)
    del ts, W
    torch.cuda.empty_cache()
with open(args.out, "w") as f:
    json.dump({"device": torch.cuda.get_device_name(0), "plans": plans}, f, indent=0, sort_keys=True)
if args.detail:
    with open(args.detail, "w") as f:
        json.dump(detail, f, indent=1, sort_keys=True)
