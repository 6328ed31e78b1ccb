"""Golden vectors for the growmap search: runs the reference's own `tree_search.py` (read-only, executed
with runpy; `torch.load` is given map_location='cpu' because the shipped acceptance vector was saved from a
CUDA tensor, `torch.save` is intercepted) on a few configs and stores config + resulting growmap + the
T.max table in tests/golden/tree_search.json.  Test infrastructure only; needs /root/reference.

    python oracle/gen_tree_search_golden.py
"""
import io
import json
import os
import runpy
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tree_search.json")

shipped = torch.load(f"{REF}/acceptance-rate-vector.pt", map_location="cpu").float().tolist()
rng = np.random.default_rng(17)
geo = np.float32(0.55) * np.float32(0.45) ** np.arange(16, dtype=np.float32)
geo = [0.0] + geo.tolist() + [float(1.0 - geo.sum())]
noisy = np.sort(rng.dirichlet(np.ones(9) * 0.7)).astype(np.float32)[::-1]
noisy = [0.0] + noisy[:8].tolist() + [float(noisy[8])]

CASES = {
    "demo": dict(p=shipped, max_depth=10, max_budget=128, draft_time=0.3, valid_budget=[1, 2, 4, 8, 16, 32],
                 target_time=[10, 10, 10, 12, 14, 18]),
    "flat_verify_128": dict(p=shipped, max_depth=10, max_budget=128, draft_time=0.1,
                            valid_budget=[1, 2, 4, 8, 16, 32, 64, 128], target_time=[5.8, 5.8, 5.8, 5.8, 5.9, 6.0, 6.1, 6.3]),
    "geometric_64": dict(p=geo, max_depth=6, max_budget=64, draft_time=0.5, valid_budget=[16, 32, 48, 64],
                         target_time=[8.0, 8.5, 9.0, 9.5]),
    "dirichlet_40": dict(p=noisy, max_depth=8, max_budget=40, draft_time=0.2, valid_budget=[8, 24, 40],
                         target_time=[3.0, 3.1, 3.3]),
}


def run_reference(case):
    saved = {}
    with tempfile.TemporaryDirectory() as d:
        torch.save(torch.tensor(case["p"], dtype=torch.float32), f"{d}/p.pt")
        cfg = {k: v for k, v in case.items() if k != "p"}
        cfg.update(acceptance_rate_vector=f"{d}/p.pt", dst=f"{d}/out.pt")
        with open(f"{d}/cfg.json", "w") as f:
            json.dump(cfg, f)
        load, save, argv = torch.load, torch.save, sys.argv
        torch.load = lambda f, *a, **k: load(f, map_location="cpu", weights_only=False)
        torch.save = lambda obj, path, *a, **k: saved.update(g=obj)
        sys.argv = ["tree_search.py", "--config", f"{d}/cfg.json"]
        try:
            with redirect_stdout(io.StringIO()):
                ns = runpy.run_path(f"{REF}/tree_search.py", run_name="__main__")
        finally:
            torch.load, torch.save, sys.argv = load, save, argv
    g = saved["g"]
    return {"roots": g["roots"], "branches": g["branches"], "Successors": g["Successors"], "size": g["size"],
            "depth": g["depth"].tolist(), "mask_rowsum": g["mask"].sum(1).tolist(), "pair": list(ns["pairs"]),
            "dec_time": float(ns["dec_time"]),
            "results": [[None if not np.isfinite(x) else float(x) for x in row] for row in ns["results"].tolist()]}


if __name__ == "__main__":
    out = {}
    for name, case in CASES.items():
        out[name] = {"config": case, "expect": run_reference(case)}
        print(name, out[name]["expect"]["pair"], out[name]["expect"]["size"], flush=True)
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
