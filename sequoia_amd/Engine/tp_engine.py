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

from typing import Optional

import torch
import torch.distributed as dist

from .Engine import InferenceEngineTG
from .Llama_modules import TreeContext


class _TPInner(InferenceEngineTG):
    def __init__(self, max_length, model_name_or_path, dtype, device, group, world, rank):
        super().__init__(max_length, model_name_or_path, dtype, device, tp_world=world, tp_rank=rank)
        self.group = group
        self.world = world
        if world > 1:
            self.model.reduce_fn = self._all_reduce
            self.model.gather_logits_fn = self._gather_vocab

    def _all_reduce(self, x):
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x

    def _gather_vocab(self, logits):
        parts = [torch.empty_like(logits) for _ in range(self.world)]
        dist.all_gather(parts, logits.contiguous(), group=self.group)
        return torch.cat(parts, dim=-1)


class TPEngine:
    """Same method set as GraphInferenceEngineTG / OffloadEngine."""

    def __init__(self, max_length: int, model_name_or_path, dtype=torch.float16, device="cuda:0",
                 process_group=None) -> None:
        self.device, self.dtype, self.max_length = device, dtype, max_length
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
        else:
            self.world, self.rank = 1, 0
        self.engine = _TPInner(max_length, model_name_or_path, dtype, device, process_group, self.world, self.rank)

    def clear_kv(self):
        self.engine.clear_kv()

    def initialize_kv(self, k_cache, v_cache, kv_len: int):
        self.engine.initialize_kv(k_cache, v_cache, kv_len)

    def get_kv_cache(self, in_place=False):
        return self.engine.get_kv_cache(in_place=in_place)

    def gather_kv(self, indices):
        self.engine.gather_kv(indices)

    def set_kv_len(self, kv_len: int):
        self.engine.set_kv_len(kv_len)

    @torch.no_grad()
    def inference(self, input_ids: torch.LongTensor, storage_ids: torch.LongTensor,
                  position_ids: Optional[torch.LongTensor] = None, attn_mask: Optional[torch.Tensor] = None,
                  tree: Optional[TreeContext] = None):
        return self.engine.model_run(input_ids=input_ids, storage_ids=storage_ids, attention_mask=attn_mask,
                                     position_ids=position_ids, tree=tree)
