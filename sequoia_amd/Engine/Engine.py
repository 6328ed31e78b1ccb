"""Inference engines with the reference's API surface (Engine/Engine.py:7-282).

InferenceEngine / GraphInferenceEngine drive the draft model, InferenceEngineTG /
GraphInferenceEngineTG the target.  Additions over the reference (all optional, the reference
call patterns keep working):
  * `tree=` keyword on inference()/graph_inference(): the tree-causal mask is evaluated inside
    the attention kernel from the growmap's ancestor bitmask instead of a dense [q, M] tensor;
  * hipGraph capture works for both flavours because the step-dependent scalars
    (q_slot0, gt, kv_len) are read from a 3-int device block, not baked into the launch.
"""
from __future__ import annotations

import gc
from typing import List, Optional

import torch

from ..ops import get_ops
from .Llama_KV import KV_Cache
from .Llama_model import KVConfigView, LlamaForCausalLM_FI, LlamaForCausalLM_TG, load_weights
from .Llama_modules import TreeContext


class InferenceEngine:
    model_cls = LlamaForCausalLM_FI

    def __init__(self, max_length: int, model_name_or_path, dtype=torch.float16, device="cuda:0", **model_kw) -> None:
        self.device = device
        self.dtype = dtype
        self.max_length = max_length
        weights = load_weights(model_name_or_path, dtype, device, **model_kw)
        self.model = self.model_cls(weights)
        self.model.eval()
        self.model.ensure_rope(max_length)          # HF regrows its rotary cache past max_position_embeddings
        self.model_config = self.model.config
        self.kv_cache = KV_Cache(config=KVConfigView(weights.dims), max_length=max_length, device=device, dtype=dtype)

    @torch.inference_mode()
    def model_run(self, input_ids: torch.LongTensor, storage_ids: torch.LongTensor,
                  attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.LongTensor] = None,
                  debug: bool = False, tree: Optional[TreeContext] = None):
        if debug:
            _, input_length = input_ids.shape
            assert storage_ids.shape[0] == input_length
            assert position_ids.shape[1] == input_length
            if attention_mask is not None:
                assert attention_mask.shape[-2] == input_length
        return self.model(input_ids=input_ids, max_length=self.max_length, storage_ids=storage_ids,
                          attention_mask=attention_mask, position_ids=position_ids, kv_cache=self.kv_cache,
                          debug=debug, tree=tree)

    def clear_kv(self):
        self.kv_cache.clear()

    def set_kv_len(self, kv_len: int):
        self.kv_cache.set_kv_len(kv_len)

    def initialize_kv(self, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: int):
        self.kv_cache.initialize_kv(k_cache, v_cache, kv_len)

    def gather_kv(self, indices):
        self.kv_cache.gather_kv(indices)

    def get_kv_cache(self, in_place=False):
        if not in_place:
            return self.kv_cache.k_cache.clone(), self.kv_cache.v_cache.clone()
        return self.kv_cache.k_cache, self.kv_cache.v_cache


class InferenceEngineTG(InferenceEngine):
    model_cls = LlamaForCausalLM_TG

    def __init__(self, max_length: int, model_name_or_path, dtype=torch.float16, device="cuda:0", offloading=False,
                 **model_kw) -> None:
        if offloading:
            raise NotImplementedError(
                "host offloading is replaced by tensor parallelism on MI355X (288 GB HBM per GPU): use "
                "Engine.offload_engine.OffloadEngine / TPEngine under torchrun instead")
        super().__init__(max_length, model_name_or_path, dtype, device, **model_kw)


_PRIMED: dict = {}


def prime_graph_rng(device):
    """The first hipGraph capture of a process registers the default generator and allocates its capture-state tensors
    in whatever mode is active then; if that is inference mode (all captures of this package are), every later capture
    OUTSIDE inference mode -- user code, a benchmark harness -- fails with "Inplace update to inference tensor".  One
    empty graph captured in normal mode first, and kept alive, pins those tensors as ordinary ones."""
    key = str(device)
    if key in _PRIMED or not str(device).startswith("cuda"):
        return
    with torch.inference_mode(False):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                pass
        torch.cuda.current_stream(device).wait_stream(s)
    _PRIMED[key] = g


class _GraphRunner:
    """One captured forward of fixed q_len: static input buffers + a hipGraph (the analogue of
    the reference's capture_graph closure, Engine/Engine.py:127-166)."""

    def __init__(self, engine: InferenceEngine, q_len: int, mempool, n_warmups: int, mode: str, n_tree: int = 1,
                 bitmask=None, independent_rows: bool = False):
        dev, M = engine.device, engine.max_length
        prime_graph_rng(dev)
        self.engine, self.q_len, self.mode = engine, q_len, mode
        self.input_ids = torch.zeros((1, q_len), dtype=torch.long, device=dev)
        self.position_ids = torch.zeros((1, q_len), dtype=torch.long, device=dev)
        self.storage_ids = torch.arange(q_len, dtype=torch.long, device=dev)
        self.tree = None
        self.mask = None
        if mode == "dense":
            self.mask = torch.zeros((1, 1, q_len, M), dtype=engine.dtype, device=dev)
        else:
            self.ctx = torch.tensor([0, 1, q_len], dtype=torch.int32, device=dev)
            self.tree = TreeContext(q_slot0=0, gt=1, n_tree=n_tree, bitmask=bitmask, kv_len=q_len, ctx=self.ctx,
                                    contiguous_slots=True, independent_rows=independent_rows)
        kv = engine.kv_cache
        saved = (kv.kv_offset, kv.dirty_end)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(n_warmups):
                kv.kv_offset = 0
                self.logits = self._run()
            s.synchronize()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        kv.kv_offset = 0
        with torch.cuda.graph(self.graph, pool=mempool):
            self.logits = self._run()
        kv.kv_offset, kv.dirty_end = saved

    def _run(self):
        return self.engine.model_run(input_ids=self.input_ids, storage_ids=self.storage_ids,
                                     position_ids=self.position_ids, attention_mask=self.mask, tree=self.tree)

    def replay(self, input_ids, storage_ids, position_ids, attn_mask=None, tree: TreeContext | None = None,
               borrow: bool = False):
        # a captured forward always produces its logits: TreeContext.need_logits = False (KV rows only) is a property of the
        # device-driven step's own launch sequence (Tree/step_graph.py), not of a graph runner
        assert tree is None or tree.need_logits, "graph runners compute logits: need_logits = False is not supported here"
        kv = self.engine.kv_cache
        ops = get_ops()
        fast = (hasattr(ops, "stage_inputs") and input_ids.is_contiguous() and storage_ids.is_contiguous()
                and position_ids.is_contiguous() and input_ids.dtype == storage_ids.dtype == position_ids.dtype == torch.int64)
        if fast:                             # one launch instead of three copies (+ the context store)
            if self.mode == "dense":
                ops.stage_inputs(self.input_ids, input_ids, self.position_ids, position_ids, self.storage_ids, storage_ids)
                self.mask.copy_(attn_mask)
            else:
                ops.stage_inputs(self.input_ids, input_ids, self.position_ids, position_ids, self.storage_ids, storage_ids,
                                 ctx=self.ctx, q_slot0=tree.q_slot0, gt=tree.gt, kv_len=tree.kv_len)
        else:
            self.input_ids.copy_(input_ids)
            self.storage_ids.copy_(storage_ids)
            self.position_ids.copy_(position_ids)
            if self.mode == "dense":
                self.mask.copy_(attn_mask)
            else:
                ops.store_i32(self.ctx, [tree.q_slot0, tree.gt, tree.kv_len])
        self.graph.replay()
        kv.note_written(self.q_len)          # host-side bookkeeping the captured forward cannot replay
        # `borrow`: the caller consumes the static output before the next replay (stream order), no clone
        return self.logits if borrow else self.logits.clone()


class GraphInferenceEngine:
    """Draft engine with per-length hipGraphs (Engine/Engine.py:168-244)."""
    inner_cls = InferenceEngine

    def __init__(self, max_length: int, model_name_or_path, dtype=torch.float16, device="cuda:0", **model_kw) -> None:
        self.device = device
        self.dtype = dtype
        self.max_length = max_length
        self.engine = self.inner_cls(max_length=max_length, model_name_or_path=model_name_or_path, dtype=dtype,
                                     device=device, **model_kw)
        self.callables = {}        # dense-mask graphs, keyed by decoding length (reference contract)
        self.tree_callables = {}   # implicit-mask graphs, keyed by (decoding length, id(bitmask))
        self.requested_graph_lengths = set()   # lengths the harness asked for (tests/testbed.py:266-268)
        self.mempool = None

    @torch.inference_mode()
    def initialize_cuda_graph(self, decoding_seqlens: List[int], n_warmups=3, tree_bitmask=None, n_tree: int = 1,
                              clear_kv: bool = True):
        """Capture one graph per decoding length (reference: Engine/Engine.py:181-195).

        Without `tree_bitmask` (the reference's call) the lengths are recorded and the graphs for
        the implicit-mask path are captured the first time a tree with a concrete growmap uses
        this engine (NativeTree.__init__, outside any timed region).  Callers that drive
        graph_inference() with a dense attn_mask tensor run eagerly unless they capture dense graphs
        explicitly with capture_dense_graphs().  With `tree_bitmask` (the growmap's device bitmask)
        the implicit-mask graphs are captured now."""
        gc.collect()
        if self.mempool is None:
            self.mempool = torch.cuda.graphs.graph_pool_handle()
        for q_len in decoding_seqlens:
            if q_len == 0:
                continue
            if tree_bitmask is None:
                self.requested_graph_lengths.add(int(q_len))
            else:
                key = (q_len, tree_bitmask.data_ptr())
                if key not in self.tree_callables:
                    self.tree_callables[key] = _GraphRunner(self.engine, q_len, self.mempool, n_warmups, "tree",
                                                            n_tree=n_tree, bitmask=tree_bitmask)
                    # a second capture for forwards whose rows never see each other (one tree level): small drafts run
                    # the attention half of a layer as one launch there (Engine/ts_linear.py::attn_block_ok)
                    from .ts_linear import block_capable
                    model = getattr(self.engine, "model", None)
                    if q_len > 1 and model is not None and block_capable(model, q_len):
                        self.tree_callables[key + ("independent",)] = _GraphRunner(
                            self.engine, q_len, self.mempool, n_warmups, "tree", n_tree=n_tree, bitmask=tree_bitmask,
                            independent_rows=True)
        if clear_kv:
            self.engine.clear_kv()

    @torch.inference_mode()
    def capture_dense_graphs(self, decoding_seqlens: List[int], n_warmups=3):
        """The reference's capture_graph behaviour (static [q, M] mask buffer per length).  Only legal
        while the KV cache is empty: the capture runs the forward on scratch slots."""
        if self.engine.kv_cache.kv_offset != 0:
            raise RuntimeError("capture_dense_graphs() must run before any token is cached")
        if self.mempool is None:
            self.mempool = torch.cuda.graphs.graph_pool_handle()
        for q_len in decoding_seqlens:
            if q_len and q_len not in self.callables:
                self.callables[q_len] = _GraphRunner(self.engine, q_len, self.mempool, n_warmups, "dense")
        self.engine.clear_kv()

    def ensure_tree_graphs(self, tree_bitmask, n_tree: int, extra_lengths=()):
        """Capture implicit-mask graphs for every requested length not yet captured for this growmap."""
        want = [q for q in sorted(self.requested_graph_lengths | set(extra_lengths))
                if (q, tree_bitmask.data_ptr()) not in self.tree_callables]
        if want:
            kv = self.engine.kv_cache
            saved = (kv.kv_offset, kv.dirty_end)
            self.initialize_cuda_graph(want, tree_bitmask=tree_bitmask, n_tree=n_tree, clear_kv=False)
            kv.kv_offset, kv.dirty_end = saved

    @torch.inference_mode()
    def graph_inference(self, input_ids: torch.LongTensor, storage_ids: torch.LongTensor,
                        position_ids: Optional[torch.LongTensor] = None, attn_mask: Optional[torch.Tensor] = None,
                        debug: bool = False, tree: Optional[TreeContext] = None, borrow: bool = False):
        dec_length = input_ids.shape[1]
        if debug:
            assert input_ids.shape[0] == 1
            assert storage_ids.shape[0] == dec_length
            assert position_ids.shape[0] == 1 and position_ids.shape[1] == dec_length
            if attn_mask is not None:
                assert attn_mask.shape[2] == dec_length and attn_mask.shape[3] == self.engine.max_length
        if tree is not None:
            # the captured forwards assume storage_ids == q_slot0 + arange(q_len) (fused RoPE + attention)
            runner = None
            if tree.contiguous_slots:
                key = (dec_length, tree.bitmask.data_ptr())
                runner = (self.tree_callables.get(key + ("independent",)) if tree.independent_rows else None) \
                    or self.tree_callables.get(key)
            if runner is not None:
                return runner.replay(input_ids, storage_ids, position_ids, tree=tree, borrow=borrow)
            return self.inference(input_ids, storage_ids, position_ids, attn_mask, tree=tree)
        runner = self.callables.get(dec_length)
        if runner is not None:
            return runner.replay(input_ids, storage_ids, position_ids, attn_mask=attn_mask)
        return self.inference(input_ids, storage_ids, position_ids, attn_mask)

    def clear_kv(self):
        self.engine.clear_kv()

    def initialize_kv(self, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: int):
        self.engine.initialize_kv(k_cache, v_cache, kv_len)

    def get_kv_cache(self, in_place=False):
        return self.engine.get_kv_cache(in_place=in_place)

    def gather_kv(self, indices):
        self.engine.gather_kv(indices)

    def set_kv_len(self, kv_len: int):
        self.engine.set_kv_len(kv_len)

    @torch.inference_mode()
    def inference(self, input_ids: torch.LongTensor, storage_ids: torch.LongTensor,
                  position_ids: Optional[torch.LongTensor] = None, attn_mask: Optional[torch.Tensor] = None,
                  tree: Optional[TreeContext] = None):
        return self.engine.model_run(input_ids=input_ids, storage_ids=storage_ids, attention_mask=attn_mask,
                                     position_ids=position_ids, tree=tree)


class GraphInferenceEngineTG(GraphInferenceEngine):
    """Target engine (Engine/Engine.py:247-282).  The reference runs it eagerly; here
    initialize_cuda_graph() is available too (the verify forward always has q_len = tree size)."""
    inner_cls = InferenceEngineTG

    def __init__(self, max_length: int, model_name_or_path, dtype=torch.float16, device="cuda:0", offloading=False,
                 **model_kw) -> None:
        super().__init__(max_length, model_name_or_path, dtype, device, offloading=offloading, **model_kw)
