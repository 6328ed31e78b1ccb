"""Static per-slot KV cache with the reference's API (Engine/Llama_KV.py:4-104) on top of the
HIP slot kernels.  Slot id == token index in the tree's `tokens` buffer; layout
[L, 1, H_kv, M, D] fp16, identical to the reference so cache contents can be compared byte
for byte.
"""
from __future__ import annotations

import os

import torch

from ..ops import get_ops


class KV_Cache:
    # What gather_kv_incremental zeroes after the accepted rows:
    #   "full"  rows [offset+len, M)   - the reference's behaviour (Llama_KV.py:65-66)
    #   "dirty" rows [offset+len, dirty_end) - only slots a tree could have written
    #   "none"  nothing                - the attention kernels never read past the live range
    ZERO_POLICY = os.environ.get("SEQUOIA_KV_ZERO", "none")

    def __init__(self, config, batch_size: int = 1, max_length: int = 256, device: str = "cuda:0",
                 dtype=torch.float16) -> None:
        if batch_size != 1:
            raise ValueError("Sequoia's KV cache is batch-1 (Engine/Llama_KV.py:8)")
        if dtype != torch.float16:
            raise ValueError("the native KV kernels are fp16, like the reference's hard-wired dtype (Tree/Tree.py:4)")
        self.config = config
        self.max_length = max_length
        self.device = device
        self.dtype = dtype
        self.num_layers = config.num_hidden_layers
        self.num_kv_heads = config.num_key_value_heads
        self.head_dim = config.hidden_size // config.num_attention_heads
        shape = (self.num_layers, batch_size, self.num_kv_heads, max_length, self.head_dim)
        self.k_cache = torch.zeros(shape, device=device, dtype=dtype)
        self.v_cache = torch.zeros(shape, device=device, dtype=dtype)
        self.kv_offset = 0
        self.dirty_end = 0          # host-side high-water mark of written slots
        self._slots_buf = torch.zeros(max_length, dtype=torch.int32, device=device)

    # ---- reference API ------------------------------------------------------------------------
    def initialize_kv(self, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: int):
        self.k_cache[..., :kv_len, :] = k_cache[..., :kv_len, :]
        self.v_cache[..., :kv_len, :] = v_cache[..., :kv_len, :]
        self.kv_offset = kv_len
        self.dirty_end = max(self.dirty_end, kv_len)

    def _zero_end(self, new_len: int) -> int:
        pol = self.ZERO_POLICY
        if pol == "full":
            return self.max_length
        if pol == "dirty":
            return max(self.dirty_end, new_len)
        return 0

    def gather_kv_incremental(self, indices, offset: int):
        """Move the accepted tree slots `indices` (ascending) to [offset, offset+len) and roll the
        cache back to that length.  `indices` is a Python list (reference signature)."""
        n = len(indices)
        if n:
            slots = torch.tensor(list(indices), dtype=torch.int32).to(self._slots_buf.device)
            self._slots_buf[:n] = slots
        self.compact_from_device(self._slots_buf, None, n, offset)

    def compact_from_device(self, slots_dev, count_dev, max_count: int, offset: int, new_len: int | None = None):
        """Sync-free variant: accepted slots / count already live on the device (the verifier's
        result record).  `new_len` is the host's knowledge of offset+count once it has it."""
        end_guess = offset + max_count if new_len is None else new_len
        get_ops().kv_compact(self.k_cache, self.v_cache, slots_dev, count_dev, max_count, offset,
                             self._zero_end(end_guess))
        if new_len is not None:
            self.kv_offset = new_len
            if self.ZERO_POLICY != "none":
                self.dirty_end = new_len
        elif count_dev is None:
            self.kv_offset = offset + max_count
            if self.ZERO_POLICY != "none":
                self.dirty_end = self.kv_offset

    def gather_kv(self, indices):
        idx = list(indices)
        ascending = all(idx[i] < idx[i + 1] for i in range(len(idx) - 1)) and all(s >= j for j, s in enumerate(idx))
        if ascending:
            self.gather_kv_incremental(idx, 0)
        else:  # arbitrary permutation: not on the hot path (Engine/Llama_KV.py:50-58)
            sel = torch.tensor(idx, dtype=torch.long, device=self.k_cache.device)
            k = self.k_cache[..., sel, :].clone()
            v = self.v_cache[..., sel, :].clone()
            self.k_cache[..., :len(idx), :] = k
            self.v_cache[..., :len(idx), :] = v
            if self.ZERO_POLICY != "none":
                self.k_cache[..., len(idx):, :] = 0.0
                self.v_cache[..., len(idx):, :] = 0.0
        self.kv_offset = len(idx)

    def update_kv_cache(self, new_k_cache: torch.Tensor, new_v_cache: torch.Tensor, layer_idx: int,
                        storage_ids: torch.LongTensor, debug: bool = False):
        """new_*: [1, H_kv, q, D] (reference layout).  Scatter into the slots of `storage_ids`."""
        input_length = len(storage_ids)
        if debug:
            assert input_length == new_k_cache.shape[-2]
            assert input_length == new_v_cache.shape[-2]
        get_ops().kv_scatter(self.k_cache[layer_idx, 0], self.v_cache[layer_idx, 0],
                             new_k_cache.reshape(self.num_kv_heads, input_length, self.head_dim).contiguous(),
                             new_v_cache.reshape(self.num_kv_heads, input_length, self.head_dim).contiguous(),
                             storage_ids)
        if layer_idx == self.num_layers - 1:
            self.note_written(input_length)
        return self.k_cache[layer_idx], self.v_cache[layer_idx]

    def note_written(self, input_length: int):
        """kv_offset protocol of the reference: advanced once per forward, on the last layer."""
        self.kv_offset += input_length
        self.dirty_end = max(self.dirty_end, self.kv_offset)

    def clear(self):
        self.k_cache.zero_()
        self.v_cache.zero_()
        self.kv_offset = 0
        self.dirty_end = 0

    def get_usable_length(self, layer_idx: int, input_length: int):
        if layer_idx == self.num_layers - 1:
            return self.kv_offset
        return self.kv_offset + input_length

    def set_kv_len(self, kv_len: int):
        self.kv_offset = kv_len
