"""GreedyTree — greedy tree speculation (reference: Tree/GreedyTree.py:6-264): children are the
top-k draft tokens, a child is accepted iff it equals the target's argmax at its parent, the
bonus token is the target argmax at the last accepted node.  Tokens are integer work: the
native path is bit-exact against the reference.
"""
from __future__ import annotations

from ._native_tree import NativeTree


class GreedyTree(NativeTree):
    stochastic = False
    _compact_when_terminal = False     # GreedyTree only gathers the KV when not terminal (:206-209)

    def _sample_level(self, i: int, lv: dict):
        fn = None if self.sampling_callables is None else self.sampling_callables.get(i)
        if fn is not None and not getattr(fn, "_sequoia_native", False):
            new_tokens_set = fn(self.draft_logits[lv["row_ids"].long()])
            self.tokens[self.num_nodes:self.num_nodes + lv["total"]] = new_tokens_set[self.sample_gather_indices[i]]
            return
        self.ops.topk(self.draft_logits, lv["row_ids"], lv["k"], self.tokens[self.num_nodes:], branch=lv["branch"],
                      out_off=lv["out_off"])

    def _verify_native(self, gt: int):
        self.ops.verify_greedy(self.target_logits, self.tokens, self.gdev["child_off"], self.gdev["child_ids"],
                               self.tree_size, gt, self.verify_ws, self.result)
