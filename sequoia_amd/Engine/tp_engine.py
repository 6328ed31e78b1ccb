"""Tensor-parallel target engine: one process per GPU, torch.distributed (backend "nccl" = RCCL
on ROCm, xGMI links between the 8 GPUs of a node).  Replaces the reference's host-offload path
for Llama-2-70B (Engine/offload_engine.py).

Sharding (Megatron-style, chosen for xGMI's point-to-point links: 2 all-reduces per layer of a
[q, hidden] fp16 activation, 2.1 MB for the 129-node 64x2 tree):
  q / k / v / gate / up  column-parallel (heads and MLP columns split across ranks),
  o / down               row-parallel  -> all-reduce(sum),
  KV cache               split by KV head: rank r owns heads [r*Hkv/W, (r+1)*Hkv/W) — RoPE, KV
                         write, tree attention and accepted-path compaction need no exchange,
  lm_head                column-parallel over the vocabulary -> all-gather of the logits.
The draft model, the samplers and the verifier run replicated on every rank with identical
noise (CPU generator, same seed) and an explicit bonus uniform, so all ranks take the same
decisions and no broadcast is needed.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .Engine import GraphInferenceEngineTG, InferenceEngineTG

FORCE_HOOKS = os.environ.get("SEQUOIA_TP_FORCE_HOOKS", "0") == "1"
# all-reduce of the row-parallel projections: "xgmi" = the two-shot kernel over peer-mapped buffers (csrc/allreduce.hip,
# Engine/xgmi_allreduce.py) with RCCL as the fallback when its setup or self-check fails; "rccl" = torch.distributed only
ALLREDUCE = os.environ.get("SEQUOIA_TP_ALLREDUCE", "xgmi")
# "1": the all-reduce of o_proj / down_proj also applies the skip connection and the following RMSNorm (row-aligned two-shot,
# sq_allreduce_add_rmsnorm_f16) -- two launches per layer fewer; "0": all-reduce, then sq_add_rmsnorm_*.  Same bits either way.
FUSED_NORM = os.environ.get("SEQUOIA_TP_FUSED_NORM", "1") == "1"


class _TPInner(InferenceEngineTG):
    def __init__(self, max_length, model_name_or_path, dtype, device, group, world, rank):
        super().__init__(max_length, model_name_or_path, dtype, device, tp_world=world, tp_rank=rank)
        self.group = group
        self.world = world
        self.collectives = 0                 # all-reduce / all-gather calls issued (eager count; tests)
        self.xgmi = None
        if world > 1 and ALLREDUCE == "xgmi":
            from .ts_linear import MAX_ROWS
            from .xgmi_allreduce import XgmiAllReduce
            self.xgmi = XgmiAllReduce.create(group, device, max_elems=MAX_ROWS * self.model.dims.hidden_size,
                                             max_gather_elems=MAX_ROWS * self.model.dims.vocab_size)
            if self.xgmi is None and str(device).startswith("cuda") and os.environ.get("SEQUOIA_TP_REQUIRE_XGMI", "0") == "1":
                raise RuntimeError("SEQUOIA_TP_REQUIRE_XGMI=1 but the xGMI all-reduce could not be set up (see stderr)")
        if world > 1 or FORCE_HOOKS:
            # (world 1 with SEQUOIA_TP_FORCE_HOOKS=1: a single-GPU box still runs every hook, RCCL call and capture)
            self.model.reduce_fn = self._all_reduce
            self.model.gather_logits_fn = self._gather_vocab
            if self.xgmi is not None:
                self.model.reduce_slabs_fn = self._all_reduce_slabs
                if FUSED_NORM:
                    self.model.reduce_norm_fn = self._all_reduce_norm

    def _all_reduce(self, x):
        self.collectives += 1
        if self.xgmi is not None and self.xgmi.fits(x):
            return self.xgmi(x)              # one kernel: peer stores over xGMI, fp32 sum in rank order
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x

    def _all_reduce_slabs(self, slab, splits, rows):
        """rows <- all-reduce of h(sum of this rank's split-K partials): the slab -> rows pass and the all-reduce in one kernel."""
        if self.xgmi is None or not self.xgmi.fits(rows):
            return None
        self.collectives += 1
        return self.xgmi.reduce_slabs(slab, splits, rows)

    def _all_reduce_norm(self, partial, splits, x, weight, out, eps, out_frag):
        """x <- x + all-reduce(partial), out <- RMSNorm(x) * weight in one kernel (row-aligned two-shot); None = does not fit."""
        if self.xgmi is None or not self.xgmi.fits_rows(x.shape[0], x.shape[1]):
            return None
        self.collectives += 1
        return self.xgmi.reduce_add_rmsnorm(partial, x, weight, out, eps, out_frag, splits=splits)

    @property
    def collectives_capturable(self):
        """True when a tree forward issues no torch.distributed call at all (both collectives on the xGMI kernels)."""
        return self.xgmi is not None

    @property
    def allreduce_kind(self):
        return "xgmi two-shot (peer-mapped buffers)" if self.xgmi is not None else "rccl"

    def _gather_vocab(self, logits):
        """[q, V / world] per rank (rank r owns vocabulary rows [r V / world, (r + 1) V / world)) -> [q, V]."""
        self.collectives += 1
        if not (dist.is_available() and dist.is_initialized()):
            return logits
        q, v = logits.shape
        logits = logits.contiguous()
        if self.xgmi is not None and self.xgmi.fits_gather(logits):
            return self.xgmi.gather_cols(logits)          # one kernel: every rank stores its columns into every peer
        if logits.device.type == "cuda":
            out = torch.empty((self.world * q, v), dtype=logits.dtype, device=logits.device)   # rank-major rows (the
            dist.all_gather_into_tensor(out, logits, group=self.group)                         # shape every backend takes)
            return out.view(self.world, q, v).permute(1, 0, 2).reshape(q, self.world * v)
        parts = [torch.empty_like(logits) for _ in range(self.world)]
        dist.all_gather(parts, logits, group=self.group)
        return torch.cat(parts, dim=-1)


class TPEngine(GraphInferenceEngineTG):
    """Same method set as GraphInferenceEngineTG / OffloadEngine, including initialize_cuda_graph / graph_inference:
    the all-reduces are captured into the forward's hipGraph (RCCL calls are stream-ordered)."""

    def __init__(self, max_length: int, model_name_or_path, dtype=torch.float16, device="cuda:0",
                 process_group=None) -> None:
        self.device, self.dtype, self.max_length = device, dtype, max_length
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
        else:
            self.world, self.rank = 1, 0
        self.engine = _TPInner(max_length, model_name_or_path, dtype, device, process_group, self.world, self.rank)
        self.callables = {}
        self.tree_callables = {}
        self.requested_graph_lengths = set()
        self.mempool = None
