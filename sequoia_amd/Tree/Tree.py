"""Base state of a speculation tree (reference: Tree/Tree.py:3-27).

The reference materialises a doubled [2M, 2M] additive mask and slides a window over it; here
the mask is implicit (growmap ancestor bitmask + ground-truth length, evaluated inside the
attention kernel), so the base class only owns the token / position buffers.
"""
from __future__ import annotations

import torch

from ..growmap import GrowMap

_GROWMAP_CACHE: dict = {}


def growmap_on_device(grow_map, device):
    """GrowMap + its device tensors, uploaded once per (growmap object, device)."""
    key = (id(grow_map), str(device))
    hit = _GROWMAP_CACHE.get(key)
    if hit is not None and hit[0] is grow_map:
        return hit[1], hit[2]
    g = GrowMap.load(grow_map)
    dev = g.device_tensors(device)
    _GROWMAP_CACHE[key] = (grow_map, g, dev)
    return g, dev


class Tree:
    def __init__(self, device: str = "cpu", max_length=512, dtype=torch.float16) -> None:
        self.tokens = torch.zeros(max_length, device=device).long()
        self.Successors: list[list[int]] = []
        self.num_nodes = 0
        self.device = device
        self.max_length = max_length
        self.dtype = dtype

    def initialize(self, attn_mask, sequence, new_tokens_buffer, parents_buffer, position_ids, active_mark):
        # caller-owned scratch, kept for signature compatibility (tests/testbed.py:53-57)
        self.full_attn_mask = attn_mask
        self.sequence = sequence
        self.new_tokens_buffer = new_tokens_buffer
        self.parents_buffer = parents_buffer
        self.position_ids = position_ids
        self.active_mark = active_mark

    def set_prefix(self, prefix: torch.LongTensor):
        n = len(prefix)
        self.tokens[:n] = prefix.to(self.device)
        self.position_ids[:n] = torch.arange(n, device=self.position_ids.device)
        self.num_nodes = n

    def verbose(self):
        print(self.tokens)
        print(self.Successors)
