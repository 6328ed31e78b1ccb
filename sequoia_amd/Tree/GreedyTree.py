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


class GreedyTreeTest(GreedyTree):
    """GreedyTree on a star growmap of `max_width` children (reference: Tree/GreedyTree.py:267-456, used by
    tests/test_accept.py to measure which of the top-k draft tokens the target picks); verify() also returns b.  Like the
    reference's probe it is rebuilt every step: the KV caches are rolled back to the accepted path and no next-root
    forward runs (the next probe's constructor feeds the bonus token)."""
    _prepare_next = False

    def __init__(self, draft_model_engine, target_model_engine, prefix, temperature: float = 0.6, top_p: float = 0.9,
                 draft_kv_len=0, target_kv_len=0, max_length=256, max_width=32, device="cpu", attn_mask=None,
                 sequence=None, new_tokens_buffer=None, parents_buffer=None, position_ids=None):
        self.max_width = max_width
        vocab = draft_model_engine.engine.model.vocab_size
        from .SpecTree import _star_growmap
        gm = _star_growmap(max_width)
        super().__init__(draft_model_engine=draft_model_engine, target_model_engine=target_model_engine, prefix=prefix,
                         temperature=temperature, top_p=top_p, draft_kv_len=draft_kv_len, target_kv_len=target_kv_len,
                         max_length=max_length, device=device, max_target_seq=max_length, vocab_size=vocab, grow_map=gm,
                         attn_mask=attn_mask, sequence=sequence, new_tokens_buffer=new_tokens_buffer,
                         parents_buffer=parents_buffer, position_ids=position_ids, step_graph=False)
        self.construct_grow_map()

    def verify(self, benchmark=False):
        from ..native import SQ_RES_LAST_NODE, SQ_RES_N_TREE
        valid, a, _, terminal = super().verify()
        res = self.last_result
        b = int(res[SQ_RES_LAST_NODE]) - 1 if int(res[SQ_RES_N_TREE]) > 0 else -1
        return valid, a, a, b, terminal
