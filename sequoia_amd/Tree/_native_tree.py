"""Shared machinery of SpecTree / GreedyTree on the native kernels.

One speculation step = construct_grow_map() (draft expansion, Tree/SpecTree.py:245-259) +
verify() (target forward, accept walk, bonus token, KV rollback, next-step prep,
Tree/SpecTree.py:159-281).  State and index algebra follow SURVEY.md §3.7:
slot == token index == KV slot; tree node t sits at slot gt-1+t; positions of tree nodes are
depth + gt - 1.

Host <-> device traffic per step: one 256-byte read of the verifier's result record (the step
API must return the accepted length to the caller); everything else is stream-ordered.
"""
from __future__ import annotations

import os
import time

import torch

from ..Engine.Llama_modules import TreeContext
from ..native import (SQ_REASON_SKIPPED, SQ_VERIFY_GATHER_FIRST, SQ_RES_ACCEPT_LEN, SQ_RES_N_TREE, SQ_RES_SLOTS, SQ_RES_TERMINAL,
                      SQ_RESULT_INTS, SQ_RESULT_RING)
from ..ops import get_ops
from .Tree import Tree, growmap_on_device


from ..Engine import xgmi_allreduce as _xgmi       # noqa: E402  (the per-step fault check of tensor-parallel jobs)


def _sync(device):
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()


# SpecTree / SpecInferTree commit order.
#   "lossless" (DEFAULT): the accepted tokens are gathered first, then the bonus token is stored -- the committed text is
#       exactly the accepted path plus the bonus token, the algorithm as published.
#   "reference": the bonus token is stored BEFORE the gather, like Tree/SpecTree.py:222-224 -- an accepted node that sits
#       at slot gt + n_accepted (the root's (n_accepted + 1)-th child leading the accepted path: the root's second child
#       accepted alone, its third child with one accepted descendant, ...) is then committed with the BONUS token's id
#       while its KV rows belong to the token that was accepted.  An upstream bug, and not a rare one: 6 of the 20 timed
#       steps of bench.py's driver run on the 128-node growmap (profiles/r04_*).  Reproducing it is what makes the token
#       stream identical to the reference's, so every replay of a reference trace selects it explicitly
#       (tests/helpers.py::make_tree) and anyone who needs the reference's stream sets SEQUOIA_COMMIT_ORDER=reference or
#       passes commit_order="reference" per tree.  It is not the product default: a generation API should not emit ids the
#       verifier never accepted (rounds 2-3 shipped "reference" as the default).
#       In this mode the quirk is observable, not silent: every such step is counted (QUIRK_STEPS, tree.quirk_steps;
#       bench.py prints config.commit_order_quirk_steps) and the first one of a process raises a warning.  The KV rows
#       are left alone in both modes: they are the accepted token's -- the context later steps condition on is the
#       accepted path; rewriting them to match the glitched id would make the model state follow the bug.
# bench.py prints the order it ran with (`config.commit_order`) and times its CPU baseline with the same one.
COMMIT_ORDER = os.environ.get("SEQUOIA_COMMIT_ORDER", "lossless")
QUIRK_STEPS = [0]
if COMMIT_ORDER not in ("reference", "lossless"):
    raise ValueError(f"SEQUOIA_COMMIT_ORDER must be 'reference' or 'lossless', got {COMMIT_ORDER!r}")
# Device-driven step (Tree/step_graph.py): "1" = every tree adopts the static step buffers and can run pipelined steps
# (enqueue_step / collect_step); the synchronous reference API (construct_grow_map / verify) works either way.
STEP_GRAPH = os.environ.get("SEQUOIA_STEP_GRAPH", "0") == "1"


class NativeTree(Tree):
    stochastic = True

    def __init__(self, draft_model_engine, target_model_engine, prefix: torch.LongTensor, temperature: float = 0.6,
                 top_p: float = 0.9, draft_kv_len=0, target_kv_len=0, max_length=256, device: str = "cpu",
                 max_target_seq=256, vocab_size=32000, grow_map=None, attn_mask=None, sequence=None,
                 new_tokens_buffer=None, parents_buffer=None, position_ids=None, residual_graph=None,
                 sampling_callables=None, sample_gather_indices=None, bonus_uniforms=None, step_graph=None,
                 commit_order=None) -> None:
        super().__init__(device=device, max_length=max_length)
        assert self.max_length == draft_model_engine.engine.max_length
        self.ops = get_ops()
        self.max_target_seq = max_target_seq
        self.draft_model_engine = draft_model_engine
        self.target_model_engine = target_model_engine
        self.temperature = temperature
        self.top_p = top_p
        self.residual_graph = residual_graph            # accepted for signature compatibility
        self.grow_map = grow_map
        self.sampling_callables = sampling_callables
        self.sample_gather_indices = sample_gather_indices
        self.vocab_size = vocab_size

        self.gm, self.gdev = growmap_on_device(grow_map, device)
        self.draft_step = self.gm.draft_step
        self.Successors = self.gm.successors
        self.tree_size = self.gm.size
        self.commit_order = commit_order or COMMIT_ORDER
        # device-driven step: adopt the static buffers of this (engines, growmap, parameters) combination; the
        # launch sequence is captured on the first fresh prompt (both caches empty), before the prefill below
        self.state = None
        want_state = STEP_GRAPH if step_graph is None else bool(step_graph)
        if want_state and sampling_callables is None and hasattr(target_model_engine.engine, "model_run"):
            from .step_graph import StepState
            st = StepState.get(self)
            if draft_kv_len == 0 and target_kv_len == 0:
                st.ensure_captured()
            if st.graph is not None or not st.cuda:
                st.reset()
                self.state = st
                self.tokens = st.tokens
        self.initialize(attn_mask, sequence, new_tokens_buffer, parents_buffer, position_ids, None)
        self.set_prefix(prefix=prefix)
        n, gt = self.tree_size, len(prefix)
        if gt + n - 1 > self.max_length:
            raise ValueError(f"prefix ({gt}) + tree ({n}) does not fit max_length {self.max_length} (README.md:47)")
        self.ground_truth_len = gt
        if self.stochastic:
            # same CPU-generator draws, in the same order, as the reference (Tree/SpecTree.py:60,84)
            self.r = self._draw_r(len(position_ids)).to(self.device)
            if self.state is not None:
                # (the draw above has the reference's length, len(position_ids), for RNG-order parity; slots past
                # max_length never exist, so the static buffer keeps the first M values)
                m = min(len(self.r), self.state.r.numel())
                self.state.r[:m].copy_(self.r[:m])
                self.r = self.state.r
        self.depth = self.gdev["depth"][1:]
        self.position_ids[gt: gt + n - 1] = self.depth + (gt - 1)
        self.storage_ids = torch.arange(self.max_length, device=self.device)
        self._arange = self.storage_ids
        # graphs the harness asked for with initialize_cuda_graph([...]) are captured for this growmap
        # on first use (the per-prompt constructor is outside the timed region, tests/testbed.py:67-79)
        if str(self.device).startswith("cuda") and draft_kv_len == 0:
            for eng in (draft_model_engine, target_model_engine):
                if getattr(eng, "requested_graph_lengths", None) and hasattr(eng, "ensure_tree_graphs"):
                    eng.ensure_tree_graphs(self.gdev["bitmask"], n)
        self.draft_logits = (self.state.draft_logits if self.state is not None else
                             torch.zeros((max(n, 1), vocab_size), dtype=self.dtype, device=self.device))

        logits = self.draft_model_engine.inference(
            input_ids=self.tokens[draft_kv_len:self.num_nodes].unsqueeze(0),
            storage_ids=self.storage_ids[draft_kv_len:self.num_nodes],
            position_ids=self.position_ids[draft_kv_len:self.num_nodes].unsqueeze(0),
            attn_mask=None, tree=self._ctx(draft_kv_len, self.num_nodes))
        self.draft_logits[0] = logits[0, -1]
        self.draft_kv_len = self.num_nodes
        self.target_kv_len = target_kv_len
        if self.stochastic:
            self._init_draft_noise(n, vocab_size)
            if bonus_uniforms is None:
                bonus_uniforms = torch.randint(0, 1 << 24, (self.max_length + 1,))
            self.bonus_u24 = [int(x) for x in bonus_uniforms]
        self.step_idx = 0
        self._no_room = None
        if self.state is not None:
            self.verify_ws, self.result = self.state.verify_ws, self.state.result
            if self.stochastic:
                nb = self.state.bonus.numel()
                tab = [self.bonus_u24[i % len(self.bonus_u24)] for i in range(nb)]
                self.state.bonus.copy_(torch.tensor(tab, dtype=torch.int32))
        else:
            self.verify_ws = self.ops.verify_workspace(n, self.device)
            self.result = torch.zeros(SQ_RESULT_INTS + n, dtype=torch.int32, device=self.device)
        self._pipe = None
        self.seq_to_use = list(range(self.max_length))
        self.target_logits = None

    # ---- helpers --------------------------------------------------------------------------------
    def _init_draft_noise(self, n: int, vocab_size: int):
        """Per-prompt noise of the draft expansion: the [n, V] uniforms of sampling without replacement
        (Tree/SpecTree.py:84), drawn on the CPU generator right after `r` like the reference."""
        rand = torch.empty((n, vocab_size), dtype=self.dtype).uniform_()
        if self.state is not None and self.state.rand is not None:
            self.state.rand.copy_(rand)
            self.rand = self.state.rand
        else:
            self.rand = rand.to(self.device)

    def _draw_r(self, m: int):
        """The acceptance uniforms r[slot] (Tree/SpecTree.py:60: fp16, drawn on the CPU generator)."""
        return torch.rand(m, dtype=self.dtype)

    def _ctx(self, q_slot0: int, kv_len: int, independent_rows: bool = False) -> TreeContext:
        # storage_ids = arange(M) (Tree/SpecTree.py:63): the queries' KV slots are q_slot0 + arange(q_len)
        return TreeContext(q_slot0=q_slot0, gt=self.ground_truth_len, n_tree=self.tree_size,
                           bitmask=self.gdev["bitmask"], kv_len=kv_len, contiguous_slots=True,
                           independent_rows=independent_rows)

    def _sample_level(self, i: int, lv: dict):
        raise NotImplementedError

    def _verify_native(self, gt: int):
        raise NotImplementedError

    def _bonus_uniform(self) -> int:
        u = self.bonus_u24[self.step_idx % len(self.bonus_u24)]
        return u | SQ_VERIFY_GATHER_FIRST if self.commit_order == "lossless" else u

    # ---- draft expansion --------------------------------------------------------------------------
    @torch.inference_mode()
    def collective_grow_static(self, idx_list, n_branch_list, benchmark=False, grow_step=None):
        lv = self.gdev["levels"][grow_step]
        total_branch = lv["total"]
        x1 = x2 = 0.0
        if benchmark:
            _sync(self.device); t1 = time.time()
        self._sample_level(grow_step, lv)
        if benchmark:
            _sync(self.device); t2 = time.time(); x1 += t2 - t1
        start_pos = self.num_nodes
        self.num_nodes = self.num_nodes + total_branch
        end_pos = self.num_nodes
        logits = self.draft_model_engine.graph_inference(
            input_ids=self.tokens[self.draft_kv_len:end_pos].unsqueeze(0),
            position_ids=self.position_ids[start_pos:end_pos].unsqueeze(0),
            attn_mask=None, storage_ids=self.storage_ids[self.draft_kv_len:end_pos],
            # exactly the level's new nodes (the usual case: the draft cache is current up to start_pos): rows that never
            # see each other
            tree=self._ctx(self.draft_kv_len, end_pos, independent_rows=self.draft_kv_len == start_pos), borrow=True)
        self.draft_kv_len = end_pos
        first = start_pos - self.ground_truth_len + 1
        self.draft_logits[first:first + total_branch] = logits[0][-total_branch:]
        if benchmark:
            _sync(self.device); t3 = time.time(); x2 += t3 - t2
            return n_branch_list, x1, x2
        return n_branch_list

    def construct_grow_map(self, benchmark=False):
        if self._no_room:
            raise ValueError(self._no_room)
        sample_time = compute_time = 0.0
        for i in range(self.draft_step - 1):
            out = self.collective_grow_static(self.gm.roots[i], self.gm.branches[i], benchmark=benchmark, grow_step=i)
            if benchmark:
                sample_time += out[1]; compute_time += out[2]
        return (sample_time, compute_time) if benchmark else None

    # ---- verification -----------------------------------------------------------------------------
    @torch.inference_mode()
    def verify(self, benchmark=False):
        gt, n = self.ground_truth_len, self.tree_size
        new_node_num = self.num_nodes - gt + 1
        start_pos = self.target_kv_len
        end_pos = self.num_nodes
        if benchmark:
            _sync(self.device); t1 = time.time()
        # a captured verify graph (q_len == tree size) is used when the engine has one; otherwise eager,
        # like the reference's GraphInferenceEngineTG.inference (Engine/Engine.py:274-282)
        tgt = self.target_model_engine
        run = getattr(tgt, "graph_inference", None)
        kw = dict(borrow=True) if run is not None else {}
        out = (run or tgt.inference)(
            input_ids=self.tokens[start_pos:end_pos].unsqueeze(0),
            position_ids=self.position_ids[start_pos:end_pos].unsqueeze(0), attn_mask=None,
            storage_ids=self.storage_ids[start_pos:end_pos], tree=self._ctx(start_pos, end_pos), **kw)
        if benchmark:
            _sync(self.device); t2 = time.time()
        self.target_logits = out[0][gt - 1 - start_pos:] if start_pos == 0 else out[0][-new_node_num:]
        assert len(self.target_logits) == new_node_num == n

        self._verify_native(gt)
        res = self.result.cpu()                      # the step's only device->host synchronisation
        if _xgmi._LIVE:
            _xgmi.raise_on_fault()                   # a collective that timed out: stop here, do not decode on stale sums
        accept_length = int(res[SQ_RES_ACCEPT_LEN])
        n_acc = int(res[SQ_RES_N_TREE])
        terminal = bool(res[SQ_RES_TERMINAL])
        self.last_result = res
        accept_list = self.seq_to_use[:gt] + [int(s) for s in res[SQ_RESULT_INTS:SQ_RESULT_INTS + n_acc]]
        self._note_commit_quirk(accept_length, accept_list[gt:], terminal)
        if benchmark:
            _sync(self.device); t3 = time.time()
        self.step_idx += 1

        if self._compact_when_terminal or not terminal:
            slots, count = self.result[SQ_RESULT_INTS:], self.result[SQ_RES_N_TREE:SQ_RES_N_TREE + 1]
            max_count = n_acc
            self.draft_model_engine.engine.kv_cache.compact_from_device(slots, count, max_count, gt, accept_length)
            self.target_model_engine.engine.kv_cache.compact_from_device(slots, count, max_count, gt, accept_length)

        if not terminal:
            if benchmark:
                _sync(self.device); t4 = time.time()
            if self._prepare_next:
                self.prepare_for_next_iter(accept_list, self.tokens[:accept_length + 1])
            if benchmark:
                return self.tokens[:accept_length + 1], accept_length, accept_length, t2 - t1, t3 - t2, t4 - t3, terminal
            return self.tokens[:accept_length + 1], accept_length, accept_length, terminal
        if benchmark:
            _sync(self.device); t4 = time.time()
            return self.tokens[:accept_length], accept_length, accept_length, t2 - t1, t3 - t2, t4 - t3, terminal
        return self.tokens[:accept_length], accept_length, accept_length, terminal

    def prepare_for_next_iter(self, accept_list, valid_tokens: torch.LongTensor):
        a = len(accept_list)
        if a + 1 > self.max_target_seq:
            return
        new_gt = len(valid_tokens)          # a + 1
        n = self.tree_size
        if new_gt + n - 1 > self.max_length:
            # the sequence is complete for callers that stop at max_length - tree size (tests/testbed.py:80 stops at
            # 256 of M = 384); another speculation step has no room for its tree -> refuse it there, not here
            self._no_room = f"{new_gt} committed tokens + tree ({n}) exceed max_length {self.max_length} (README.md:47)"
            self.ground_truth_len = new_gt
            self.num_nodes = new_gt
            return
        # accepted node j of the path sat at depth j+1, i.e. position gt+j == its new index, so the
        # compacted prefix is simply 0..a (Tree/SpecTree.py:264-266 evaluates to the same values)
        self.position_ids[:new_gt] = self._arange[:new_gt]
        torch.add(self.depth, new_gt - 1, out=self.position_ids[new_gt:new_gt + n - 1])
        self.ground_truth_len = new_gt
        self.num_nodes = new_gt
        logits = self.draft_model_engine.graph_inference(
            input_ids=self.tokens[a:new_gt].unsqueeze(0), storage_ids=self.storage_ids[a:new_gt],
            position_ids=self.position_ids[a:new_gt].unsqueeze(0), attn_mask=None, tree=self._ctx(a, new_gt),
            borrow=True)
        self.draft_logits[0] = logits[0, -1]
        self.draft_kv_len = new_gt
        self.target_kv_len = a

    quirk_steps = 0

    def _note_commit_quirk(self, a: int, slots, terminal: bool):
        """commit_order == "reference": the bonus token was stored at slot a BEFORE tokens[gt:a] = tokens[accepted slots]
        (Tree/SpecTree.py:222-224); when an accepted node sits at slot a its committed id is the bonus token's."""
        if terminal or not self.stochastic or self.commit_order != "reference" or a not in slots:
            return
        self.quirk_steps += 1
        QUIRK_STEPS[0] += 1
        if QUIRK_STEPS[0] == 1:
            import warnings
            warnings.warn("sequoia_amd: this step committed the bonus token's id over an accepted token (the reference's "
                          "store-before-gather order, Tree/SpecTree.py:222-224, reproduced for token parity; the KV cache holds "
                          "the accepted token).  SEQUOIA_COMMIT_ORDER=lossless / commit_order='lossless' commits the accepted "
                          "path itself.  Further occurrences are counted in sequoia_amd.Tree._native_tree.QUIRK_STEPS.", stacklevel=3)

    _compact_when_terminal = True
    _prepare_next = True       # the acceptance probes rebuild the tree every step: no next-root forward (Tree/SpecTree.py:283)

    # ---- device-driven steps (Tree/step_graph.py) ------------------------------------------------------------------
    def begin_pipeline(self):
        """Hand the loop to the device: the step block takes over the ground-truth length.  Legal once the target is
        prefilled (after at least one synchronous step) and when the next step has room."""
        st = self.state
        if st is None:
            raise RuntimeError("this tree has no step state (construct it with step_graph=True / SEQUOIA_STEP_GRAPH=1)")
        if self.target_kv_len == 0 or self._no_room:
            raise RuntimeError("begin_pipeline() needs a prefilled target (run one synchronous step) and room for a step")
        import collections
        gt = self.ground_truth_len
        block = torch.zeros(st.step.numel(), dtype=torch.int32)
        block[0], block[1], block[2], block[3] = gt, gt, self.step_idx, 1
        st.step.copy_(block)
        if self.stochastic:      # statistics of the root's draft row (the synchronous path does not keep them)
            self.ops.logits_stats(self.draft_logits[0:1], self.temperature, st.stats[0:1])
        self._pipe = dict(inflight=collections.deque(), gt_known=gt, next_idx=self.step_idx, dead=False)

    def can_enqueue(self, horizon: int) -> bool:
        """True when one more step may be enqueued without knowing the results of the steps still in flight: even if
        each of them commits a full path the tree and the bonus token fit the buffers and the sequence stays below
        `horizon` tokens (the caller's stopping length)."""
        p = self._pipe
        per_step = self.gdev["max_depth"] + 1
        gt_max = p["gt_known"] + per_step * len(p["inflight"])
        n = self.tree_size
        return (not p["dead"] and gt_max < horizon and gt_max + n - 1 <= self.max_length
                and gt_max + per_step <= self.max_length)

    def enqueue_step(self):
        st, p = self.state, self._pipe
        idx = p["next_idx"]
        slot = idx % SQ_RESULT_RING
        st.ring_np[slot, 7] = -1                 # the walker stores the step index here LAST: the record's arrival flag
        st.launch()
        p["inflight"].append(idx)
        p["next_idx"] = idx + 1

    def collect_step(self, timeout_s: float = 30.0):
        """Result of the oldest step in flight: (accept_length, n_accepted, bonus_token, terminal).  Polls that step's
        256-byte record in pinned host memory: it arrives when the step's verification is done, while the step's KV
        compaction and next-root forward (and the following step, if enqueued) are still running."""
        st, p = self.state, self._pipe
        idx = p["inflight"].popleft()
        rec = st.ring_np[idx % SQ_RESULT_RING]
        if st.cuda:
            t0 = time.time()
            spins = 0
            while int(rec[7]) != idx:
                spins += 1
                if (spins & 0xfff) == 0 and time.time() - t0 > timeout_s:
                    if _xgmi._LIVE:
                        _xgmi.raise_on_fault()       # a collective that gave up explains the missing record: say THAT
                    raise RuntimeError(f"step {idx}: no result record after {timeout_s} s")
        rec = rec.copy()
        if _xgmi._LIVE:
            _xgmi.raise_on_fault()                   # (the fault word is pinned host memory: no device read)
        a, n_acc, bonus, terminal = int(rec[SQ_RES_ACCEPT_LEN]), int(rec[SQ_RES_N_TREE]), int(rec[2]), bool(rec[SQ_RES_TERMINAL])
        assert int(rec[7]) == idx, f"result ring slot {idx % SQ_RESULT_RING} holds step {int(rec[7])}, expected {idx}"
        if p["dead"]:
            # a step that was in flight behind the terminal one: the device skipped its commit (SQ_STEP_ACTIVE == 0,
            # csrc/verify.hip) -- the text, the caches and the host mirrors stay as the terminal step left them
            assert terminal and n_acc == 0 and int(rec[4]) == SQ_REASON_SKIPPED, f"step {idx} ran behind a terminal step but committed"
            return self.ground_truth_len, 0, -1, True
        self.last_result = rec
        self.step_idx = idx + 1
        self._note_commit_quirk(a, [int(x) for x in rec[SQ_RES_SLOTS:SQ_RES_SLOTS + min(n_acc, SQ_RESULT_INTS - SQ_RES_SLOTS)]], terminal)
        if terminal:
            p["dead"] = True
        else:
            p["gt_known"] = a + 1
        self._sync_host_state(a, terminal)
        return a, n_acc, bonus, terminal

    def _sync_host_state(self, a: int, terminal: bool):
        """Host mirrors of what the device step did (the synchronous API keeps working after a pipelined stretch)."""
        new_gt = a if terminal else a + 1
        n = self.tree_size
        self.ground_truth_len = new_gt
        self.num_nodes = new_gt
        self.draft_kv_len = new_gt
        self.target_kv_len = new_gt - 1
        dkv, tkv = self.draft_model_engine.engine.kv_cache, self.target_model_engine.engine.kv_cache
        dkv.kv_offset, tkv.kv_offset = new_gt, new_gt - 1
        dkv.dirty_end = max(dkv.dirty_end, new_gt + n - 1); tkv.dirty_end = max(tkv.dirty_end, new_gt + n - 1)
        if new_gt + n - 1 > self.max_length:
            self._no_room = f"{new_gt} committed tokens + tree ({n}) exceed max_length {self.max_length} (README.md:47)"

    def end_pipeline(self):
        """Drain the steps still in flight and restore the host-side buffers the synchronous path reads."""
        p = self._pipe
        if p is None:
            return
        while p["inflight"]:
            self.collect_step()
        if self.state.cuda:
            torch.cuda.current_stream().synchronize()      # a record arrives before its step's tail (compaction, next root)
        gt, n = self.ground_truth_len, self.tree_size
        self.position_ids[:gt] = self._arange[:gt]
        if gt + n - 1 <= self.max_length:
            torch.add(self.depth, gt - 1, out=self.position_ids[gt:gt + n - 1])
        self._pipe = None
