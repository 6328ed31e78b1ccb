"""Growmap search: the Sequoia dynamic program that turns an acceptance-rate vector and a
(draft time, verify time per budget) table into the tree the speculation loop grows.

Replaces the reference's `tree_search.py` (:14-132).  Same config keys, same on-disk growmap
(`{roots, branches, Successors, mask, depth, size}`, tree_search.py:121-128), same result on the same
inputs — including every tie, because the recurrence is evaluated in the reference's float32
arithmetic and every argmax takes the first maximum — but as array sweeps with back-pointers
instead of per-element tensor loops and deep-copied child lists (budget 128 / depth 10: 36 s -> <0.1 s;
budget 1024 becomes practical).

    T[m, l, b] = expected number of accepted tokens of the best tree with m nodes, depth <= l, whose
                 root has exactly b children, when the k-th child drawn without replacement is
                 accepted with probability p[k]:
        T[1, l, 0] = 1
        T[m, l, 1] = 1 + p[1] * max_b' T[m-1, l-1, b']
        T[m, l, b] = max_{1 <= y < m}  T[y, l, b-1] + p[b] * max_b' T[m-y, l-1, b']

    python -m sequoia_amd.tree_search --config demo-config.json
"""
from __future__ import annotations

import argparse
import json
from dataclasses import dataclass

import numpy as np

NEG = np.float32(-np.inf)


@dataclass
class SearchTables:
    T: np.ndarray          # float32 [budget+1, depth+1, branch+1]
    best: np.ndarray       # float32 [budget+1, depth+1]      max over b
    best_b: np.ndarray     # int32   [budget+1, depth+1]      first argmax over b
    split: np.ndarray      # int32   [budget+1, depth+1, branch+1]  y of the recurrence (0 = none)


def load_acceptance_vector(path: str) -> np.ndarray:
    """The reference stores `[0, p_1 .. p_w, p_none]` as a torch tensor (tests/test_accept.py:86-89);
    `.json` / `.npy` with the same layout are accepted too."""
    if path.endswith(".json"):
        with open(path) as f:
            return np.asarray(json.load(f), dtype=np.float32)
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32)
    import torch
    return torch.load(path, map_location="cpu", weights_only=False).float().numpy()


def search_tables(p: np.ndarray, max_budget: int, max_depth: int) -> SearchTables:
    """p: float32 [max_branch + 1], p[0] unused (tree_search.py:14-15 drops the trailing reject-all entry
    before this point)."""
    p = np.asarray(p, dtype=np.float32)
    W = p.shape[0] - 1
    B, D = max_budget, max_depth
    T = np.full((B + 1, D + 1, W + 1), NEG, dtype=np.float32)
    best = np.full((B + 1, D + 1), NEG, dtype=np.float32)
    best_b = np.zeros((B + 1, D + 1), dtype=np.int32)
    split = np.zeros((B + 1, D + 1, W + 1), dtype=np.int32)
    if B >= 1:
        T[1, 1:, 0] = 1.0
        best[1, 1:] = 1.0
    for m in range(2, B + 1):
        # child-subtree values for every split y = 1 .. m-1:  best[m - y, l - 1]  (rows y, columns l)
        sub = best[m - 1:0:-1, 1:D]                                    # [m-1, D-1]  -> l = 2 .. D
        for b in range(1, W + 1):
            if b == 1:
                # T[m, l, 1] = 1 + p[1] * best[m-1, l-1]   (tree_search.py:31)
                with np.errstate(invalid="ignore"):
                    one = np.float32(1.0) + p[1] * best[m - 1, 1:D]
                T[m, 2:, 1] = np.where(np.isnan(one), NEG, one)
                split[m, 2:, 1] = 1
                continue
            prev = T[1:m, 2:, b - 1]                                    # [m-1, D-1]
            if not np.isfinite(prev).any():
                break                                                   # no tree with b-1 root children fits: none with b does
            with np.errstate(invalid="ignore"):
                cand = prev + p[b] * sub                                # float32: product rounded, then the sum
            cand = np.where(np.isnan(cand), NEG, cand)                  # 0 * -inf: the reference's '>' scan skips NaN
            y = np.argmax(cand, axis=0)                                 # first maximum, like the strict '>' scan (:37-41)
            val = cand[y, np.arange(D - 1)]
            T[m, 2:, b] = val
            split[m, 2:, b] = np.where(val > NEG, y + 1, 0)
        best[m] = T[m].max(axis=1)
        best_b[m] = T[m].argmax(axis=1)
    best_b[1] = 0
    return SearchTables(T, best, best_b, split)


def choose_budget_depth(tab: SearchTables, draft_time: float, target_time, valid_budget):
    """min over (budget, depth) of (depth * draft_time + target_time[budget]) / E[accepted]
    (tree_search.py:60-76), rounded as torch rounds `float / float32 tensor`; first strict minimum."""
    dec_time, pair = np.float32(np.inf), None
    for i, b in enumerate(valid_budget):
        for d in range(tab.best.shape[1]):
            ac = tab.best[b, d]
            if ac < 0:
                continue
            x = (np.float32(1.0) / ac) * np.float32(d * draft_time + target_time[i])   # torch's scalar / tensor = reciprocal * scalar
            if x < dec_time:
                dec_time, pair = x, (b, d)
    return float(dec_time), pair


def children_states(tab: SearchTables, m: int, l: int, b: int):
    """The b child states (budget, depth, branch) of state (m, l, b), in sampling order: state
    (m, l, b) = the first b-1 children packed into split[m, l, b] nodes, then the b-th child's subtree."""
    kids = []
    while b >= 1:
        y = 1 if b == 1 else int(tab.split[m, l, b])
        kids.append((m - y, l - 1, int(tab.best_b[m - y, l - 1])))
        m, b = y, b - 1
    return kids[::-1]


def build_growmap(tab: SearchTables, budget: int, depth: int) -> dict:
    """Breadth-first expansion of the chosen state into the growmap (tree_search.py:78-128)."""
    root = (budget, depth, int(tab.best_b[budget, depth]))
    states, parents, node_depth, successors = [root], [-1], [0], [[]]
    roots, branches = [], []
    frontier = [0]
    while frontier:
        roots.append(list(frontier))
        level_branch, nxt = [], []
        for i in frontier:
            m, l, b = states[i]
            level_branch.append(b)
            kids = children_states(tab, m, l, b)
            assert len(kids) == b
            for st in kids:
                j = len(states)
                states.append(st); parents.append(i); node_depth.append(node_depth[i] + 1); successors.append([])
                successors[i].append(j); nxt.append(j)
        branches.append(level_branch)
        frontier = nxt
    n = len(states)
    assert n == budget
    mask = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        if parents[i] >= 0:
            mask[i] = mask[parents[i]]
        mask[i, i] = 1
    return {"roots": roots, "branches": branches, "Successors": successors, "mask": mask,
            "depth": np.asarray(node_depth, dtype=np.int64), "size": n}


def search(config: dict) -> tuple[dict, dict]:
    """config: the reference's JSON keys (demo-config.json).  Returns (growmap, report)."""
    p = load_acceptance_vector(config["acceptance_rate_vector"]) if isinstance(config["acceptance_rate_vector"], str) \
        else np.asarray(config["acceptance_rate_vector"], dtype=np.float32)
    p = p[:-1]                                                          # drop P(reject all)  (tree_search.py:14)
    tab = search_tables(p, config["max_budget"], config["max_depth"])
    dec_time, pair = choose_budget_depth(tab, config["draft_time"], config["target_time"], config["valid_budget"])
    if pair is None:
        raise ValueError("no feasible (budget, depth) among valid_budget")
    g = build_growmap(tab, *pair)
    report = {"time_per_token": dec_time, "speedup_vs_autoregressive": config["target_time"][0] / dec_time,
              "budget": pair[0], "depth": pair[1], "expected_accepted": float(tab.best[pair])}
    return g, report


def save_growmap(g: dict, path: str) -> None:
    """`.pt` in the reference's format (torch tensors for mask / depth) or `.json` (successors only, the
    in-tree fixture format `GrowMap.load` reads)."""
    if path.endswith(".json"):
        with open(path, "w") as f:
            json.dump({"Successors": g["Successors"]}, f)
        return
    import torch
    out = dict(g)
    out["mask"] = torch.from_numpy(g["mask"]).long()
    out["depth"] = torch.from_numpy(g["depth"]).long()
    torch.save(out, path)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="demo-config.json")
    args = ap.parse_args(argv)
    with open(args.config) as f:
        config = json.load(f)
    g, report = search(config)
    print(json.dumps(report))
    save_growmap(g, config["dst"])


if __name__ == "__main__":
    main()
