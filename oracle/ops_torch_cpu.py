"""CPU ops for bench.py's `cpu_baseline` leg: the reference's own PyTorch op sequences (test / measurement
infrastructure, never the product path).  OracleOps (numpy, ops_adapter.py) restates every op element by element for
checking; timing THAT against the GPU would flatter the GPU, so the heavy row-wise ops are restated here the way the
reference runs them on a CPU -- fp16 torch tensors, the same calls in the same order:

  attention      Engine/Llama_modules.py:226-256   matmul / sqrt(d) + mask, softmax in fp32, cast, matmul
  RoPE           Engine/offload_engine.py:42-67     q cos + rotate_half(q) sin on fp16 tensors
  RMSNorm        Engine/Llama_modules.py:284-288    fp32 variance, rsqrt, cast, weight multiply
  SwiGLU gate    Engine/Llama_modules.py:270-271    silu(gate) * up
  samplers       utils.py:10-18, 29-32              softmax, rand.log() / q, topk

profiles/r02_cpu_reference_vs_port.json records how close this port runs to the imported reference on the same weights
(oracle/ref_cpu_baseline.py).  Verification stays on the numpy oracle (a few rows, sequential)."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops_np as O
from .ops_adapter import OracleOps, _np


def _rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


class TorchCpuOps(OracleOps):
    name = "torch-cpu"

    def rope_kv_write(self, qkv, q_out, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads, h_kv, d):
        q_len = qkv.shape[0]
        q = qkv[:, :n_heads * d].view(q_len, n_heads, d).transpose(0, 1)
        k = qkv[:, n_heads * d:(n_heads + h_kv) * d].view(q_len, h_kv, d).transpose(0, 1)
        v = qkv[:, (n_heads + h_kv) * d:].view(q_len, h_kv, d).transpose(0, 1)
        c, s = cos[position_ids].unsqueeze(0), sin[position_ids].unsqueeze(0)
        q_out.copy_((q * c) + (_rotate_half(q) * s))
        k_layer[:, storage_ids] = (k * c) + (_rotate_half(k) * s)
        v_layer[:, storage_ids] = v

    def tree_attention(self, q, k_layer, v_layer, out, kv_len, scale, dense_mask=None, q_slot0=0, gt=0, n_tree=0,
                       bitmask=None, ctx=None):
        if ctx is not None:
            q_slot0, gt, kv_len = (int(x) for x in ctx[:3])
        n_heads, q_len, d = q.shape
        if dense_mask is not None:
            mask = dense_mask.reshape(dense_mask.shape[-2], dense_mask.shape[-1])[:, :kv_len]
        else:
            bm = _np(bitmask).view(np.uint64) if bitmask is not None else np.ones((1, 1), np.uint64)
            mask = torch.from_numpy(O.tree_mask_dense(q_slot0, q_len, kv_len, gt, n_tree, bm))
        k, v = k_layer[:, :kv_len], v_layer[:, :kv_len]
        rep = n_heads // k.shape[0]
        if rep > 1:
            k, v = k.repeat_interleave(rep, dim=0), v.repeat_interleave(rep, dim=0)
        w = torch.matmul(q, k.transpose(1, 2)) / math.sqrt(d) + mask
        w = torch.nn.functional.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        out.copy_(torch.matmul(w, v).transpose(0, 1).reshape(q_len, n_heads * d))
        return out

    def rmsnorm(self, x, weight, out, eps):
        h = x.to(torch.float32)
        h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
        out.copy_(weight * h.to(x.dtype))
        return out

    def add_rmsnorm(self, x, residual, sum_out, weight, out, eps):
        sum_out.copy_(x + residual)
        return self.rmsnorm(sum_out, weight, out, eps)

    def silu_mul(self, gate_up, out):
        inter = out.shape[-1]
        out.copy_(torch.nn.functional.silu(gate_up[:, :inter]) * gate_up[:, inter:])
        return out

    def sample_wor(self, logits, rand, row_ids, k, temperature, out, branch=None, out_off=None, out_base=None, stats=None):
        idx = row_ids.long() if row_ids is not None else torch.arange(logits.shape[0])
        q = torch.softmax(logits[idx] / temperature, dim=-1)
        pos = (rand[idx].log() / q).topk(k=k).indices
        self._emit(pos.numpy(), out, branch, out_off, out_base)
        return out

    def topk(self, logits, row_ids, k, out, branch=None, out_off=None, out_base=None):
        idx = row_ids.long() if row_ids is not None else torch.arange(logits.shape[0])
        self._emit(logits[idx].topk(k=k).indices.numpy(), out, branch, out_off, out_base)
        return out
