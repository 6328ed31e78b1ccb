"""GreedySTree — the "greedy draft, sampled target" baseline (reference: Tree/GreedySTree.py:8-268): the draft tree
is GreedyTree's (top-k children), but each node's target token is SAMPLED from softmax(filtered logits / T)
(:188-190) instead of taken as the argmax; the walk accepts a child iff it equals its parent's sampled token.

The reference samples with `multinomial(1)` on the device generator; here the draw of node t is the exact inverse
CDF at an explicit 24-bit uniform (fresh per step, CPU generator) — reproducible and checkable against the oracle.
"""
from __future__ import annotations

import torch

from .GreedyTree import GreedyTree


class GreedySTree(GreedyTree):
    stochastic = False
    _compact_when_terminal = False

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        n = self.tree_size
        self.target_u24 = torch.zeros((n, 1), dtype=torch.int32, device=self.device)
        self.target_token = torch.zeros(n, dtype=torch.long, device=self.device)
        self.target_uniforms = None          # tests may pin the per-step uniforms: list of int arrays [n]

    def _verify_native(self, gt: int):
        n = self.tree_size
        if self.target_uniforms is not None:
            u = torch.as_tensor(self.target_uniforms[self.step_idx % len(self.target_uniforms)], dtype=torch.int32)
        else:
            u = torch.randint(0, 1 << 24, (n,), dtype=torch.int32)
        self.target_u24.copy_(u.reshape(n, 1), non_blocking=True)
        if self.top_p < 1.0:
            self.ops.top_p_filter(self.target_logits, self.top_p, self.temperature)
        self.ops.sample_iid(self.target_logits, self.target_u24, None, 1, self.temperature, self.target_token)
        self.ops.verify_tokens(self.target_token, self.tokens, self.gdev["child_off"], self.gdev["child_ids"], n, gt,
                               self.verify_ws, self.result)
