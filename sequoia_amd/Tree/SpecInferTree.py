"""SpecInferTree — the SpecInfer baseline of the paper (reference: Tree/SpecInferTree.py:7-291) on the native
kernels: children are k i.i.d. draws WITH replacement from the draft distribution, a child is accepted iff
p[tok] >= r q[tok], a rejection replaces p by the residual and leaves q unchanged.  Same constructor, step API
and growmaps as SpecTree.

The reference draws with `multinomial(replacement=True)` on the device generator (:108); here every draw is the
exact inverse CDF at an explicit 24-bit uniform (fresh uniforms per speculation step, CPU generator), so the step is
reproducible across devices and checkable against the oracle — the same deliberate deviation as the bonus draw.
"""
from __future__ import annotations

import torch

from ._native_tree import NativeTree


class SpecInferTree(NativeTree):
    stochastic = True
    _compact_when_terminal = True

    def _init_draft_noise(self, n: int, vocab_size: int):
        # keeps the reference's generator order (r, then the [n, V] noise matrix it also allocates, :60,84)
        self.rand = torch.empty((n, vocab_size), dtype=self.dtype).uniform_().to(self.device)
        self._kmax = max([lv["k"] for lv in self.gdev["levels"]] + [1])
        self.draw_u24 = torch.zeros((n, self._kmax), dtype=torch.int32, device=self.device)
        self.draw_uniforms = None            # tests may pin the per-step uniforms: list of int arrays [n, kmax]

    def construct_grow_map(self, benchmark=False):
        if self.draw_uniforms is not None:
            u = torch.as_tensor(self.draw_uniforms[self.step_idx % len(self.draw_uniforms)], dtype=torch.int32)
        else:
            u = torch.randint(0, 1 << 24, (self.tree_size, self._kmax), dtype=torch.int32)
        self.draw_u24.copy_(u.reshape(self.tree_size, self._kmax), non_blocking=True)
        return super().construct_grow_map(benchmark=benchmark)

    def _sample_level(self, i: int, lv: dict):
        # rows of this level use their own rows of the uniform table: gather them in level order
        u = self.draw_u24[lv["row_ids"].long(), :lv["k"]].contiguous()
        self.ops.sample_iid(self.draft_logits, u, lv["row_ids"], lv["k"], self.temperature, self.tokens[self.num_nodes:],
                            branch=lv["branch"], out_off=lv["out_off"])

    def _verify_native(self, gt: int):
        if self.top_p < 1.0:
            self.ops.top_p_filter(self.target_logits, self.top_p, self.temperature)
        self.ops.verify_specinfer(self.target_logits, self.draft_logits, self.tokens, self.r, self.gdev["child_off"],
                                  self.gdev["child_ids"], self.tree_size, gt, self.temperature,
                                  self._bonus_uniform(), self.verify_ws, self.result)
