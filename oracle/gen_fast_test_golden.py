"""Golden vector for the static acceptance estimator: runs the REFERENCE's own `evaluate` (tests/fast_test.py:36-108) on
synthetic logits and records inputs + output (test infrastructure; runs only where /root/reference exists).

tests/fast_test.py cannot be imported (it parses arguments and downloads models at import time), so its two function
definitions (`get_residual`, `evaluate`) are compiled from the file's AST -- the reference's code, executed unmodified --
and called with stand-in models that return recorded logits.

    python oracle/gen_fast_test_golden.py       # -> tests/golden/fast_test.npz
"""
import ast
import os
import types

import numpy as np
import torch

REF = os.environ.get("SEQUOIA_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_functions():
    src = open(os.path.join(REF, "tests", "fast_test.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("get_residual", "evaluate")]
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"torch": torch, "softmax": torch.nn.functional.softmax, "tqdm": lambda it, total=None: it,
          "LlamaForCausalLM": object, "DataLoader": object}
    exec(compile(mod, "fast_test.py", "exec"), ns)
    return ns["evaluate"]


class _Fake:
    def __init__(self, logits):
        self.logits = logits
        self.i = 0

    def __call__(self, **batch):
        out = types.SimpleNamespace(logits=self.logits[self.i].clone())
        self.i += 1
        return out


def main():
    evaluate = reference_functions()
    arrays = {}
    cases = [("plain", 6, 1.0, 1.1, 0.6), ("topp", 5, 0.9, 0.99, 0.6), ("hot", 4, 1.0, 1.1, 1.0)]
    for name, k, top_p, dtp, T in cases:
        gen = torch.Generator().manual_seed(len(name) * 101)
        V, L, rows = 512, 140, 2
        tl = [(torch.randn(1, L, V, generator=gen) * 3) for _ in range(rows)]
        dl = [(t + torch.randn(1, L, V, generator=gen) * 2) for t in tl]
        labels = [torch.randint(1, V, (1, L), generator=gen) for _ in range(rows)]
        labels[0][0, 130] = -100
        labels[1][0, 135] = 0
        loader = [dict(labels=lb) for lb in labels]
        torch.manual_seed(99)
        out = evaluate(_Fake(tl), _Fake(dl), loader, k=k, T=T, top_p=top_p, draft_top_p=dtp)
        arrays[f"{name}/target"] = torch.stack(tl).numpy()
        arrays[f"{name}/draft"] = torch.stack(dl).numpy()
        arrays[f"{name}/labels"] = torch.stack(labels).numpy()
        arrays[f"{name}/params"] = np.array([k, top_p, dtp, T], dtype=np.float64)
        arrays[f"{name}/out"] = out.numpy()
        print(name, out)
    path = os.path.join(REPO, "tests", "golden", "fast_test.npz")
    np.savez_compressed(path, **arrays)
    print("->", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
