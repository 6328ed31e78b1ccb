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


_STAR: dict = {}


def _star_growmap(width: int) -> dict:
    """One growmap dict per width for the life of the process (the probe classes are rebuilt every step)."""
    g = _STAR.get(width)
    if g is None:
        g = _STAR[width] = {"Successors": [list(range(1, width + 1))] + [[] for _ in range(width)]}
    return g


class SpecTreeTest(SpecTree):
    """The acceptance-rate probe of tests/test_accept.py (reference: Tree/SpecTree.py:283-481): a one-level tree of
    `max_width` children, built AND grown by the constructor, rebuilt for every step with the KV lengths carried over;
    verify() returns (valid_tokens, a, a, b, terminal) with b the index of the accepted child (-1: none).

    The probe is NOT SpecTree's arithmetic: its noise is fp32 (`torch.rand(M)`, `torch.empty((w + 1, V)).uniform_()`,
    :327,347), so the sampling keys rand.log() / q are fp32 and the acceptance test is `p >= r q` evaluated in fp32
    (:412) -- sq_sample_wor_f32noise_f16 / sq_verify_probe_f16 reproduce exactly that; the accepted tokens are gathered
    before the bonus token is appended (:472-474) and the KV caches are rolled back to the accepted path
    (gather_kv, :476-477) with no next-root forward (the next probe's constructor feeds the bonus token)."""
    _prepare_next = False
    _compact_when_terminal = False       # gather_kv only when not terminal (:476-479)

    def __init__(self, draft_model_engine, target_model_engine, prefix, temperature: float = 0.6, top_p: float = 0.9,
                 draft_kv_len=0, target_kv_len=0, max_length=256, max_width=32, device="cpu", attn_mask=None,
                 sequence=None, new_tokens_buffer=None, parents_buffer=None, position_ids=None, bonus_uniforms=None):
        self.max_width = max_width
        vocab = draft_model_engine.engine.model.vocab_size
        super().__init__(draft_model_engine=draft_model_engine, target_model_engine=target_model_engine, prefix=prefix,
                         temperature=temperature, top_p=top_p, draft_kv_len=draft_kv_len, target_kv_len=target_kv_len,
                         max_length=max_length, device=device, max_target_seq=max_length, vocab_size=vocab,
                         grow_map=_star_growmap(max_width), attn_mask=attn_mask, sequence=sequence,
                         new_tokens_buffer=new_tokens_buffer, parents_buffer=parents_buffer, position_ids=position_ids,
                         bonus_uniforms=bonus_uniforms, step_graph=False, commit_order="lossless")
        self.construct_grow_map()

    def _draw_r(self, m: int):
        return torch.rand(m)                                            # fp32 (:327)

    def _init_draft_noise(self, n: int, vocab_size: int):
        self.rand = torch.empty((n, vocab_size)).uniform_().to(self.device)     # fp32 [max_width + 1, V] (:347)

    def _sample_level(self, i: int, lv: dict):
        self.ops.sample_wor_f32noise(self.draft_logits, self.rand, lv["row_ids"], lv["k"], self.temperature,
                                     self.tokens[self.num_nodes:], branch=lv["branch"], out_off=lv["out_off"])

    def _verify_native(self, gt: int):
        if self.top_p < 1.0:
            self.ops.top_p_filter(self.target_logits, self.top_p, self.temperature)
        self.ops.verify_probe(self.target_logits, self.draft_logits, self.tokens, self.r, self.gdev["child_off"],
                              self.gdev["child_ids"], self.tree_size, gt, self.temperature, self._bonus_uniform(),
                              self.verify_ws, self.result)

    def verify(self, benchmark=False):
        from ..native import SQ_RES_LAST_NODE, SQ_RES_N_TREE
        valid, a, _, terminal = super().verify()
        res = self.last_result
        b = int(res[SQ_RES_LAST_NODE]) - 1 if int(res[SQ_RES_N_TREE]) > 0 else -1
        return valid, a, a, b, terminal
