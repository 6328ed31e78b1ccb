"""Device-driven speculation step: construct_grow_map() + verify() + the next step's preparation
(Tree/SpecTree.py:245-281, tests/testbed.py:80-87) as ONE launch sequence in which no launch argument depends on the
step.  The ground-truth length lives in a device block (include/sequoia_hip.h, SQ_STEP_*): the samplers, the input
staging of every forward, the verifier, both KV compactions and the 1-token draft forward read it there, the verifier
writes the next one.  On a HIP device the sequence is captured once per (engines, growmap, sampling parameters) into a
hipGraph and replayed per step; the verifier writes each step's 256-byte result record into a ring in pinned host
memory, which the host polls one step late -- the GPU never waits for the host inside or between steps.

The static buffers (tokens, draft logits, noise, step block, staging buffers) belong to a StepState that outlives the
per-prompt tree objects of the harness: a new tree adopts them (per-prompt constructor, outside the timed region).
"""
from __future__ import annotations

import os

import torch

from ..Engine.Llama_modules import TreeContext
from ..Engine.ts_linear import MAX_ROWS as TS_MAX_ROWS
from ..native import (SQ_RES_N_TREE, SQ_RESULT_INTS, SQ_RESULT_RING, SQ_STEP_ACTIVE, SQ_STEP_GT, SQ_STEP_INTS,
                      SQ_VERIFY_GATHER_FIRST)
from ..ops import get_ops
from .Tree import _content_key

N_BONUS = 1024
# "1": a forward on the tall-skinny path stages its own inputs in its first launch (sq_embed_stage_rmsnorm_f16) -- 7 launches per
# step fewer; "0": sq_stage_tree_inputs in front of every forward
FUSE_STAGE = os.environ.get("SEQUOIA_FUSE_STAGE", "1") == "1"
# The draft forward over the LAST tree level: its nodes are leaves -- no child is ever sampled from their draft rows
# (Tree/SpecTree.py:103) -- so the step needs only their KV rows (a leaf can be accepted and then belongs to the next
# step's context), not their logits.
#   KV_ONLY_LAST_LEVEL ("1", default): that forward stops after its last layer's RoPE + KV write (no attention / o_proj /
#       MLP of the last layer, no final norm, no lm_head, no row statistics: TreeContext.need_logits = False).  Committed
#       tokens are identical by construction (the draft rows of leaves are never read); "0" runs the full forward.
#   OVERLAP_LAST_LEVEL ("0", default): "1" additionally captures that forward on a forked stream, concurrent with the
#       target's verify forward, joined before the verifier; "2" the same with the step captured on a high-priority stream
#       and the fork on a low-priority one.  MEASURED NEGATIVE on MI355X (profiles/r05_overlap_last_level_not_adopted.md:
#       config B 5.29 -> 5.40 ms / step, config D 12.03 -> 12.22): the target's projections need all 256 CUs in ONE
#       resident wave of workgroups, and every draft kernel that holds a CU when such a launch starts stretches its tail
#       by more than the draft forward saves.  Kept as an opt-in for other tree / model shapes.
KV_ONLY_LAST_LEVEL = os.environ.get("SEQUOIA_KV_ONLY_LAST_LEVEL", "1") == "1"
OVERLAP_LAST_LEVEL = os.environ.get("SEQUOIA_OVERLAP_LAST_LEVEL", "0")

class _Fwd:
    """Static inputs of one forward of fixed q_len (the analogue of a _GraphRunner's buffers)."""

    def __init__(self, q_len, n_tree, bitmask, device, independent_rows=False):
        self.q_len = q_len
        self.ids = torch.zeros((1, q_len), dtype=torch.long, device=device)
        self.pos = torch.zeros((1, q_len), dtype=torch.long, device=device)
        self.sto = torch.zeros(q_len, dtype=torch.long, device=device)
        self.ctx = torch.tensor([0, 1, q_len], dtype=torch.int32, device=device)
        self.tree = TreeContext(q_slot0=0, gt=1, n_tree=n_tree, bitmask=bitmask, kv_len=q_len, ctx=self.ctx,
                                contiguous_slots=True, independent_rows=independent_rows)


class StepState:
    """Static buffers + the captured step of one (draft engine, target engine, growmap, sampling parameters)."""

    @staticmethod
    def get(tree) -> "StepState":
        # The states live ON the target engine object (not in a module-level table): a StepState references both engines
        # (weights, KV caches), a captured graph with its private pool and [n, V] buffers -- it has to die with the
        # engines it serves (`del draft, target` frees everything; StepState.release(target) drops the states early).
        key = (id(tree.draft_model_engine), _content_key(tree.grow_map), type(tree).__name__,
               float(tree.temperature), float(tree.top_p), int(tree.max_length), int(tree.vocab_size), str(tree.device),
               tree.commit_order)
        table = tree.target_model_engine.__dict__.setdefault("_step_states", {})
        st = table.get(key)
        if st is None or st.draft is not tree.draft_model_engine:
            st = table[key] = StepState(tree)
        return st

    @staticmethod
    def release(target_engine):
        """Drop every step state (graphs, static buffers) captured for this target engine."""
        target_engine.__dict__.pop("_step_states", None)

    def __init__(self, tree):
        dev, n, V, M = tree.device, tree.tree_size, tree.vocab_size, tree.max_length
        self.device, self.n, self.V, self.M = dev, n, V, M
        self.ops = get_ops()
        self.stochastic = tree.stochastic
        self.temperature, self.top_p = tree.temperature, tree.top_p
        self.commit_order = tree.commit_order
        self.gdev = tree.gdev
        self.draft, self.target = tree.draft_model_engine, tree.target_model_engine
        f16 = torch.float16
        self.tokens = torch.zeros(M, dtype=torch.long, device=dev)
        self.draft_logits = torch.zeros((n, V), dtype=f16, device=dev)
        self.r = torch.zeros(M, dtype=f16, device=dev)
        self.rand = torch.zeros((n, V), dtype=f16, device=dev) if self.stochastic else None
        self.stats = torch.zeros(self.ops.stats_shape(n, V), dtype=torch.float32, device=dev) if self.stochastic else None
        self.step = torch.zeros(SQ_STEP_INTS, dtype=torch.int32, device=dev)
        self.bonus = torch.zeros(N_BONUS, dtype=torch.int32, device=dev)
        self.result = torch.zeros(SQ_RESULT_INTS + n, dtype=torch.int32, device=dev)
        self.cuda = str(dev).startswith("cuda")
        # result ring: pinned host memory on a HIP device -- the walker writes each step's record straight into it and the
        # host polls the record's step-index word (no copy, no event wait: an event wait costs about a millisecond of
        # wake-up latency per step on this runtime)
        self.ring = torch.zeros(SQ_RESULT_RING * SQ_RESULT_INTS, dtype=torch.int32, device="cpu" if self.cuda else dev)
        if self.cuda:
            self.ring = self.ring.pin_memory()
        self.ring_np = self.ring.numpy().reshape(SQ_RESULT_RING, SQ_RESULT_INTS) if self.ring.device.type == "cpu" else None
        self.verify_ws = self.ops.verify_workspace(n, dev)
        bm = self.gdev["bitmask"]
        # the new nodes of one tree level never see each other (none is another's ancestor): TreeContext.independent_rows
        self.fwd_levels = [_Fwd(lv["total"], n, bm, dev, independent_rows=True) for lv in self.gdev["levels"]]
        self.fwd_target = _Fwd(n, n, bm, dev)
        self.fwd_one = _Fwd(1, n, bm, dev)
        self.graph = None

    # ---- the launch sequence of one step --------------------------------------------------------------------
    def body(self):
        ops, g, n, T = self.ops, self.gdev, self.n, self.temperature
        gt_dev = self.step[SQ_STEP_GT:SQ_STEP_GT + 1]
        depth = g["depth32"]
        dm, tm = self.draft.engine, self.target.engine
        n_levels = len(g["levels"])
        kv_only = KV_ONLY_LAST_LEVEL and self.cuda and getattr(dm.model, "ts", None) is not None
        # (a sharded draft runs collectives: its forward stays on the step's one stream, in the ranks' common order)
        overlap = kv_only and OVERLAP_LAST_LEVEL in ("1", "2") and getattr(dm.model, "reduce_fn", None) is None
        side = None
        for li, (lv, f) in enumerate(zip(g["levels"], self.fwd_levels)):
            first = lv["first_child"]
            out = self.tokens[first - 1:]                 # + gt on the device: tokens[gt + first - 1 + out_off[r] + s]
            if self.stochastic:
                ops.sample_wor(self.draft_logits, self.rand, lv["row_ids"], lv["k"], T, out, branch=lv["branch"],
                               out_off=lv["out_off"], out_base=gt_dev, stats=self.stats)
            else:
                ops.topk(self.draft_logits, lv["row_ids"], lv["k"], out, branch=lv["branch"], out_off=lv["out_off"],
                         out_base=gt_dev)
            if kv_only and not overlap and li == n_levels - 1 and lv["total"] <= TS_MAX_ROWS:
                f.tree.need_logits = False                     # leaves: KV rows only
                self._stage(f, first - 1, first - 1 + lv["total"])
                dm.model_run(input_ids=f.ids, storage_ids=f.sto, position_ids=f.pos, attention_mask=None, tree=f.tree)
                continue
            if overlap and li == n_levels - 1 and lv["total"] <= TS_MAX_ROWS:
                # leaves: KV rows only, off the critical path.  The inputs are staged on the step's stream (the fork below
                # orders the side stream behind them; the walker that rewrites `tokens` / the step block runs after the join)
                f.tree.stage = None
                f.tree.need_logits = False
                self.ops.stage_tree_inputs(f.ids, f.pos, f.sto, f.ctx, self.tokens, g["depth32"], self.n, first - 1,
                                           first - 1 + lv["total"], self.step, False)
                main = torch.cuda.current_stream()
                side = self._side_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    dm.model_run(input_ids=f.ids, storage_ids=f.sto, position_ids=f.pos, attention_mask=None, tree=f.tree)
                continue
            f.tree.need_logits = True
            self._stage(f, first - 1, first - 1 + lv["total"])
            logits = dm.model_run(input_ids=f.ids, storage_ids=f.sto, position_ids=f.pos, attention_mask=None, tree=f.tree)
            self._adopt_rows(logits[0], first, lv["total"])
        f = self.fwd_target
        self._stage(f, -1, n - 1)
        target_logits = tm.model_run(input_ids=f.ids, storage_ids=f.sto, position_ids=f.pos, attention_mask=None,
                                     tree=f.tree)[0]
        self.target_logits = target_logits
        if side is not None:                                   # join: the leaves' draft KV rows are in place
            torch.cuda.current_stream().wait_stream(side)
        flags = SQ_VERIFY_GATHER_FIRST if self.commit_order == "lossless" else 0
        if self.stochastic:
            if self.top_p < 1.0:
                ops.top_p_filter(target_logits, self.top_p, T)
            ops.verify_stochastic(target_logits, self.draft_logits, self.tokens, self.r, g["child_off"], g["child_ids"], n, 0,
                                  T, flags, self.verify_ws, self.result, step=self.step, bonus_table=self.bonus,
                                  result_ring=self.ring)
        else:
            ops.verify_greedy(target_logits, self.tokens, g["child_off"], g["child_ids"], n, 0, self.verify_ws, self.result,
                              step=self.step, result_ring=self.ring)
        slots, count = self.result[SQ_RESULT_INTS:], self.result[SQ_RES_N_TREE:SQ_RES_N_TREE + 1]
        # both caches roll back to the accepted path in one launch (Tree/SpecTree.py:226-227)
        ops.kv_compact2(dm.kv_cache, tm.kv_cache, slots, count, g["max_depth"], 0, dst_offset_dev=gt_dev)
        f = self.fwd_one                                   # next root: the bonus token at slot new_gt - 1
        self._stage(f, -1, 0, advance=True)
        logits = dm.model_run(input_ids=f.ids, storage_ids=f.sto, position_ids=f.pos, attention_mask=None, tree=f.tree)
        self._adopt_rows(logits[0], 0, 1)

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device, priority=0)      # (0 = the lowest priority)
        return self._side

    def _stage(self, f, rel_slot0, rel_kv_len, advance=False):
        """Inputs of the forward that follows: handed to the model through the forward's TreeContext -- on the tall-skinny
        path (CUDA) the forward's first launch stages them itself (ops.embed_stage_rmsnorm), elsewhere
        ops.stage_tree_inputs runs here as a launch of its own."""
        args = (f.ids, f.pos, f.sto, f.ctx, self.tokens, self.gdev["depth32"], self.n, rel_slot0, rel_kv_len, self.step, advance)
        if self.cuda and FUSE_STAGE:
            f.tree.stage = args                      # Engine/Llama_model.py::forward falls back to the separate launch
        else:
            f.tree.stage = None
            self.ops.stage_tree_inputs(*args)

    def _adopt_rows(self, rows, first, total):
        """draft_logits[first:first+total] = the forward's last `total` rows (Tree/SpecTree.py:121,279), with the
        per-part softmax statistics of those rows for the next sampler (one launch instead of a copy)."""
        src = rows[-total:]
        if self.stochastic:
            self.ops.logits_stats(src, self.temperature, self.stats[first:first + total],
                                  copy_dst=self.draft_logits[first:first + total])
        else:
            self.draft_logits[first:first + total].copy_(src)

    # ---- capture --------------------------------------------------------------------------------------------
    def _kv_marks(self):
        return [(e.engine.kv_cache, e.engine.kv_cache.kv_offset, e.engine.kv_cache.dirty_end) for e in (self.draft, self.target)]

    @torch.inference_mode()
    def ensure_captured(self):
        """Warm the sequence up (launch plans, weight images, workspaces) and capture it.  Runs real steps on scratch
        state (gt = 1, empty caches): call before a prompt is prefilled; both KV caches are cleared afterwards."""
        if self.graph is not None or not self.cuda:
            return
        from ..Engine.Engine import prime_graph_rng
        prime_graph_rng(self.device)
        marks = self._kv_marks()
        self.step.zero_()
        self.step[SQ_STEP_GT] = 1
        self.step[SQ_STEP_ACTIVE] = 1
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self.step[SQ_STEP_GT] = 1
                self.body()
            s.synchronize()
        torch.cuda.current_stream().wait_stream(s)
        self.step[SQ_STEP_GT] = 1
        graph = torch.cuda.CUDAGraph()
        kw = dict(stream=torch.cuda.Stream(device=self.device, priority=-1)) if OVERLAP_LAST_LEVEL == "2" else {}
        with torch.cuda.graph(graph, **kw):
            self.body()
        self.graph = graph
        for kv, off, dirty in marks:
            kv.kv_offset, kv.dirty_end = off, dirty
        self.draft.clear_kv(); self.target.clear_kv()
        self.reset()

    def reset(self):
        self.tokens.zero_()
        self.draft_logits.zero_()
        self.step.zero_()
        self.ring.zero_()

    def launch(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.body()
