"""Growmap (tree shape) handling.

The reference stores a tree as a dict {roots, branches, Successors, mask, depth, size}
(tree_search.py:121-128).  All of it follows from `Successors` (children of every node, BFS
order, children contiguous), so that is the only thing kept; the derived views are the ones
the native kernels consume: children CSR, ancestor bitmask, per-level sampler plan.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import numpy as np
import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
BUILTIN_DIR = os.path.join(_PKG, "growmaps")


@dataclass
class LevelPlan:
    """One draft-expansion level (SpecTree.collective_grow_static, Tree/SpecTree.py:87-134)."""
    row_ids: np.ndarray      # int32 tree-local ids of the parents (= roots[i])
    branch: np.ndarray       # int32 children per parent (= branches[i])
    k: int                   # max(branch): samples drawn per row
    out_off: np.ndarray      # int32 offset of each parent's children inside the level's new nodes
    total: int               # sum(branch): nodes added by this level
    first_child: int         # tree-local id of the first node added


@dataclass
class GrowMap:
    successors: list
    size: int = 0
    roots: list = field(default_factory=list)
    branches: list = field(default_factory=list)
    depth: np.ndarray = None          # int64 [n]
    child_off: np.ndarray = None      # int32 [n+1]
    child_ids: np.ndarray = None      # int32 [n-1]
    bitmask: np.ndarray = None        # uint64 [n, words]
    levels: list = field(default_factory=list)

    # ---- construction ---------------------------------------------------------------------------
    @staticmethod
    def from_successors(successors) -> "GrowMap":
        succ = [[int(c) for c in ch] for ch in successors]
        n = len(succ)
        if n < 1:
            raise ValueError("empty growmap")
        flat = [c for ch in succ for c in ch]
        if flat != list(range(1, n)):
            raise ValueError("Successors must enumerate nodes 1..n-1 in BFS order (children contiguous)")
        g = GrowMap(successors=succ, size=n)
        depth = np.zeros(n, dtype=np.int64)
        parent = np.full(n, -1, dtype=np.int64)
        for p, ch in enumerate(succ):
            for c in ch:
                depth[c] = depth[p] + 1
                parent[c] = p
        g.depth = depth
        n_levels = int(depth.max()) + 1
        g.roots = [[int(i) for i in np.nonzero(depth == d)[0]] for d in range(n_levels)]
        g.branches = [[len(succ[i]) for i in lvl] for lvl in g.roots]
        off = np.zeros(n + 1, dtype=np.int32)
        for i, ch in enumerate(succ):
            off[i + 1] = off[i] + len(ch)
        g.child_off = off
        g.child_ids = np.asarray(flat, dtype=np.int32)
        words = (n + 63) // 64
        bm = np.zeros((n, words), dtype=np.uint64)
        bm[0, 0] = 1
        for c in range(1, n):
            bm[c] = bm[parent[c]]
            bm[c, c // 64] |= np.uint64(1) << np.uint64(c % 64)
        g.bitmask = bm
        first = 1
        for lvl in range(n_levels - 1):
            br = np.asarray(g.branches[lvl], dtype=np.int32)
            total = int(br.sum())
            out_off = np.concatenate([[0], np.cumsum(br)[:-1]]).astype(np.int32)
            g.levels.append(LevelPlan(row_ids=np.asarray(g.roots[lvl], dtype=np.int32), branch=br,
                                      k=int(br.max()) if len(br) else 0, out_off=out_off, total=total,
                                      first_child=first))
            first += total
        return g

    @staticmethod
    def from_reference_dict(d) -> "GrowMap":
        return GrowMap.from_successors(d["Successors"])

    @staticmethod
    def load(path_or_name) -> "GrowMap":
        """A reference .pt growmap, one of our .json files, or the name of a bundled growmap."""
        if isinstance(path_or_name, GrowMap):
            return path_or_name
        if isinstance(path_or_name, dict):
            return GrowMap.from_reference_dict(path_or_name)
        p = str(path_or_name)
        if not os.path.exists(p):
            cand = os.path.join(BUILTIN_DIR, p if p.endswith(".json") else p + ".json")
            if os.path.exists(cand):
                p = cand
            else:
                raise FileNotFoundError(path_or_name)
        if p.endswith(".json"):
            with open(p) as f:
                return GrowMap.from_successors(json.load(f)["Successors"])
        return GrowMap.from_reference_dict(torch.load(p, weights_only=False))

    # ---- views --------------------------------------------------------------------------------
    @property
    def draft_step(self) -> int:
        return len(self.roots)

    def dense_mask(self) -> np.ndarray:
        """growmap['mask']: int64 [n, n], 1 where column is an ancestor-or-self of the row."""
        n = self.size
        m = np.zeros((n, n), dtype=np.int64)
        for i in range(n):
            for w in range(self.bitmask.shape[1]):
                bits = int(self.bitmask[i, w])
                while bits:
                    b = bits & -bits
                    m[i, w * 64 + b.bit_length() - 1] = 1
                    bits ^= b
        return m

    def to_reference_dict(self) -> dict:
        """The reference's on-disk dict (tree_search.py:121-128)."""
        return dict(roots=[list(r) for r in self.roots], branches=[list(b) for b in self.branches],
                    Successors=[list(s) for s in self.successors], mask=torch.from_numpy(self.dense_mask()),
                    depth=torch.from_numpy(self.depth.copy()), size=self.size)

    def device_tensors(self, device):
        """Everything the kernels need, uploaded once per (growmap, device)."""
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        levels = []
        for lv in self.levels:
            levels.append(dict(row_ids=t(lv.row_ids), branch=t(lv.branch), out_off=t(lv.out_off), k=lv.k,
                               total=lv.total, first_child=lv.first_child, n_rows=len(lv.row_ids)))
        return dict(child_off=t(self.child_off), child_ids=t(self.child_ids) if self.size > 1 else None,
                    bitmask=t(self.bitmask.view(np.int64)), depth=t(self.depth), depth32=t(self.depth.astype(np.int32)),
                    max_depth=int(self.depth.max()), levels=levels)
