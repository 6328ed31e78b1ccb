"""Building blocks of the Llama forward used by both engines.

The reference implements these as nn.Modules on HF internals (Engine/Llama_modules.py); here
they are plain functions over a fused-weight layer record, with the hot ops in HIP:
RMSNorm(+residual) -> sq_(add_)rmsnorm_f16, RoPE + KV slot write -> sq_rope_kv_write_f16,
tree-batched attention -> sq_tree_attention_f16, SwiGLU gate -> sq_silu_mul_f16.  The dense
projections in THIS module are PyTorch's (hipBLASLt): it serves prompt prefill (> 128 rows) and
tensor-parallel shards; tree forwards of <= 128 rows run Engine/ts_linear.py instead.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from ..ops import get_ops

# "1" (default): a forward whose rows never see each other -- the draft forward over one tree level, any one-row forward --
# runs RoPE + KV write + tree attention of a layer as one launch (csrc/draft_block.hip::level_attention_kernel) instead of two
LEVEL_ATTENTION = os.environ.get("SEQUOIA_LEVEL_ATTENTION", "1") == "1"


def rope_tables(head_dim: int, max_pos: int, base: float, device, dtype=torch.float16):
    """cos/sin caches exactly as LlamaRotaryEmbedding_FI builds them (Engine/Llama_modules.py:
    17-45): fp32 on the CPU, then cast — so every device sees the same fp16 table."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype).to(device).contiguous(), emb.sin().to(dtype).to(device).contiguous()


@dataclass
class TreeContext:
    """Implicit tree-causal mask of one forward: the queries sit at slots
    [q_slot0, q_slot0+q_len); slots < gt are committed text (causal), slots >= gt are tree
    nodes gt-1+t whose visibility is the growmap's ancestor bitmask.  `ctx` (optional, device
    int32[3] = {q_slot0, gt, kv_len}) makes the launch replayable from a hipGraph."""
    q_slot0: int
    gt: int
    n_tree: int
    bitmask: torch.Tensor          # int64 [n_tree, words] (uint64 bit patterns)
    kv_len: int
    ctx: torch.Tensor | None = None
    contiguous_slots: bool = False     # storage_ids == q_slot0 + arange(q_len): enables the fused RoPE+attention launch
    # device-driven step (Tree/step_graph.py): the arguments of ops.stage_tree_inputs for this forward's static input buffers;
    # a forward on the tall-skinny path stages its inputs inside its first launch (ops.embed_stage_rmsnorm)
    stage: tuple | None = None
    # False: the caller needs this forward's KV rows only, not its logits -- the draft forward over the LAST tree level
    # (its nodes are leaves: no child is ever sampled from their rows, Tree/SpecTree.py:103).  The tall-skinny forward
    # then stops after the last layer's RoPE + KV write (Engine/ts_linear.py::forward_ts) and returns None.  Consequence for
    # readers of the tree's buffers: the `draft_logits` rows (and sampler statistics) of LEAF nodes are not refreshed by a
    # device-driven step -- they hold an earlier step's values; nothing on the path reads them (no child is sampled from a leaf,
    # the verifier reads the draft rows of internal nodes only).  Honoured by forward_ts alone; the graph runners
    # (Engine/Engine.py::_GraphRunner) and the chunked forward assert it is True.
    need_logits: bool = True
    # True: no query of this forward may see another query's key (only its own) -- the draft forward over ONE tree level
    # (the new nodes of a level are siblings / cousins, none is another's ancestor: Tree/SpecTree.py:87-134).  Small drafts
    # then run the attention half of every layer as one launch (csrc/draft_block.hip, Engine/ts_linear.py::attn_block_ok);
    # a one-row forward qualifies without the flag.
    independent_rows: bool = False


@dataclass
class LayerWeights:
    ln1: torch.Tensor              # [hidden]
    wqkv: torch.Tensor             # [(H + 2 H_kv) D, hidden]   packed q | k | v
    wo: torch.Tensor               # [hidden, H D]
    ln2: torch.Tensor
    w_gate_up: torch.Tensor        # [2 I, hidden]              packed gate | up
    w_down: torch.Tensor           # [hidden, I]


def attention_core(qkv, layer_idx: int, dims, kv_cache, cos, sin, position_ids, storage_ids, dense_mask,
                   tree: TreeContext | None, out_frag: bool = False, qkv_slab=None, kv_only: bool = False):
    """qkv: [q, (H + 2 H_kv) D] packed projections -- or qkv_slab = (fp32 slab, splits, q_len): the split-K partials of the
    tall-skinny projection, summed by the RoPE kernel.  RoPE + KV slot write + tree-batched attention;
    returns the attention output [q, H D] (the o_proj input), or with out_frag its fragment-major image
    (the operand layout of the tall-skinny o_proj, Engine/ts_linear.py).  kv_only: RoPE + KV slot write alone (returns None)."""
    ops = get_ops()
    q_len = qkv.shape[0] if qkv_slab is None else qkv_slab[2]
    n_heads, h_kv, d = dims.local_heads, dims.local_kv_heads, dims.head_dim
    k_layer, v_layer = kv_cache.k_cache[layer_idx, 0], kv_cache.v_cache[layer_idx, 0]
    dt, dev = k_layer.dtype, k_layer.device
    attn = torch.empty(ops.frag_shape(q_len, n_heads * d) if out_frag else (q_len, n_heads * d), dtype=dt, device=dev)
    scale = 1.0 / math.sqrt(d)
    frag_kw = dict(out_frag=True) if out_frag else {}
    if (LEVEL_ATTENTION and not kv_only and tree is not None and (q_len == 1 or tree.independent_rows)
            and tree.contiguous_slots and d in (64, 128) and hasattr(ops, "level_attention")):
        # rows that never see each other (one tree level / one row): RoPE + KV write + attention are ONE launch
        ops.level_attention(qkv, attn, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads, h_kv, d, scale,
                            tree.q_slot0, tree.gt, tree.n_tree, bitmask=tree.bitmask, ctx=tree.ctx, out_frag=out_frag,
                            qkv_slab=None if qkv_slab is None else (qkv_slab[0], qkv_slab[1], q_len, (n_heads + 2 * h_kv) * d))
        return attn
    q_rot = torch.empty((n_heads, q_len, d), dtype=dt, device=dev)
    if qkv_slab is not None:
        ops.rope_kv_write_slabs(qkv_slab[0], qkv_slab[1], (n_heads + 2 * h_kv) * d, q_rot, k_layer, v_layer, cos, sin,
                                position_ids, storage_ids, n_heads, h_kv, d)
    else:
        ops.rope_kv_write(qkv, q_rot, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads, h_kv, d)
    if kv_only:
        return None
    if tree is not None:
        ops.tree_attention(q_rot, k_layer, v_layer, attn, tree.kv_len, scale, q_slot0=tree.q_slot0, gt=tree.gt,
                           n_tree=tree.n_tree, bitmask=tree.bitmask, ctx=tree.ctx, **frag_kw)
    else:
        if dense_mask is None:
            raise ValueError("attention needs either a dense additive mask or a TreeContext")
        kv_len = dense_mask.shape[-1]
        if kv_len > kv_cache.max_length:
            raise ValueError(f"Attention mask should cover at most {kv_cache.max_length} key slots, got {kv_len}")
        ops.tree_attention(q_rot, k_layer, v_layer, attn, kv_len, scale, dense_mask=dense_mask, **frag_kw)
    return attn


def attention_block(h, lw: LayerWeights, layer_idx: int, dims, kv_cache, cos, sin, position_ids, storage_ids,
                    dense_mask, tree: TreeContext | None, reduce_fn=None):
    """h: [q, hidden] normalised input.  Returns the o_proj output [q, hidden].

    Replaces LlamaAttention_FI/TG.forward (Engine/Llama_modules.py:87-140, 182-258)."""
    qkv = F.linear(h, lw.wqkv)                                         # hipBLASLt
    attn = attention_core(qkv, layer_idx, dims, kv_cache, cos, sin, position_ids, storage_ids, dense_mask, tree)
    out = F.linear(attn, lw.wo)
    if reduce_fn is not None:
        out = reduce_fn(out)                                           # TP: row-parallel all-reduce
    return out


def mlp_block(h, lw: LayerWeights, dims, reduce_fn=None):
    """SwiGLU MLP (LlamaMLP_FI, Engine/Llama_modules.py:259-272) with gate|up fused."""
    ops = get_ops()
    gu = F.linear(h, lw.w_gate_up)
    act = torch.empty((h.shape[0], lw.w_down.shape[1]), dtype=h.dtype, device=h.device)
    ops.silu_mul(gu, act)
    out = F.linear(act, lw.w_down)
    if reduce_fn is not None:
        out = reduce_fn(out)
    return out
