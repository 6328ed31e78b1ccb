"""SpecTree — Sequoia's stochastic tree speculation (reference: Tree/SpecTree.py:7-281) on the
native kernels.  Same constructor and step API; the per-child host loop of the reference
(accept_step, one device->host sync per child) is one verifier launch here.
"""
from __future__ import annotations

import torch

from ._native_tree import NativeTree


class SpecTree(NativeTree):
    stochastic = True
    _compact_when_terminal = True      # the reference rolls the KV back even on a terminal step (:226-227)

    def _sample_level(self, i: int, lv: dict):
        """Children of level i: sampling without replacement from softmax(draft_logits/T) via the
        exponential-race keys log(u)/q (utils.py:10-18), gathered per parent (SpecTree.py:103-104)."""
        fn = None if self.sampling_callables is None else self.sampling_callables.get(i)
        if fn is not None and not getattr(fn, "_sequoia_native", False):
            # foreign callable injected by the caller: honour the reference contract literally
            idx = lv["row_ids"].long()
            new_tokens_set = fn(self.draft_logits[idx], self.rand[idx])
            gather = self.sample_gather_indices[i]
            self.tokens[self.num_nodes:self.num_nodes + lv["total"]] = new_tokens_set[gather]
            return
        self.ops.sample_wor(self.draft_logits, self.rand, lv["row_ids"], lv["k"], self.temperature,
                            self.tokens[self.num_nodes:], branch=lv["branch"], out_off=lv["out_off"])

    def _verify_native(self, gt: int):
        if self.top_p < 1.0:
            # get_sampling_logits (utils.py:65-77) on the target rows, in place like the reference (:196)
            self.ops.top_p_filter(self.target_logits, self.top_p, self.temperature)
        self.ops.verify_stochastic(self.target_logits, self.draft_logits, self.tokens, self.r, self.gdev["child_off"],
                                   self.gdev["child_ids"], self.tree_size, gt, self.temperature,
                                   self._bonus_uniform(), self.verify_ws, self.result)
