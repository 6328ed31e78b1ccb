"""CPU oracle for the Sequoia tree-speculation hot path — TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the reference's algorithm for every op that
libsequoia_hip.so implements.  It is the checker for the HIP kernels; it is never the thing
that is shipped or measured.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; nothing under sequoia_amd/ does.

Pinning: the reference ships no golden vectors, so the oracle is pinned
against traces produced by running the *reference itself* in the build container
(oracle/gen_golden.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py), and, wherever the
reference checkout is present, against the live reference: its utils.py functions and tree_search.py on seeded
random inputs and fresh traces of its tree classes for seeds outside the committed set
(tests/test_oracle_live_reference_cpu.py).

Arithmetic contract (what "the reference computes" means here).  The reference is fp16 torch
code; every torch op on an fp16 tensor computes in fp32 and rounds the result to fp16 once
(probed against torch 2.10 CPU: `x/T` == fp16(f32(x)/f32(T)) bit-for-bit; softmax ==
fp16(exp(y-max)/sum) up to the exp implementation, 4e-5 of elements differ by one fp16 ulp).
The functions below restate each op with exactly those rounding points.  `h()` is the
round-to-nearest-even fp32 -> fp16 conversion.  Two reductions are defined *exactly*
(order-independent integer arithmetic on the fp16 grid, every fp16 in [0, 1] is a multiple of
2^-24) so that the HIP kernels can match them bit-for-bit: the residual normaliser and the
inverse-CDF of the bonus draw.

Reference citations are relative to the reference checkout (/root/reference in the build
container).
"""
from __future__ import annotations

import numpy as np

F16_MIN = np.float16(-65504.0)  # torch.finfo(torch.float16).min, the reference's "masked" value
EOS_IDS = (0, 2)                # Tree/SpecTree.py:208


def h(x):
    """fp32 -> fp16, round to nearest even (what every fp16 torch op does to its result);
    overflow goes to +-inf like torch's cast."""
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=np.float32).astype(np.float16)


def f(x):
    return np.asarray(x).astype(np.float32)


# --------------------------------------------------------------------------------------------
# a1. tree-causal mask (Tree/Tree.py:13-27, Tree/SpecTree.py:45-58,270-271)
# --------------------------------------------------------------------------------------------
def bitmask_from_successors(successors):
    """Ancestor-or-self bitmask of each node == growmap['mask'] (tree_search.py:121-128).

    Returns uint64[n][words]; bit j of row i set iff j is an ancestor of i or j == i."""
    n = len(successors)
    words = (n + 63) // 64
    out = np.zeros((n, words), dtype=np.uint64)
    parent = [-1] * n
    for p, ch in enumerate(successors):
        for c in ch:
            parent[c] = p
    for i in range(n):
        j = i
        while j >= 0:
            out[i, j // 64] |= np.uint64(1) << np.uint64(j % 64)
            j = parent[j]
    return out


def bit(bitmask, i, j):
    return bool((int(bitmask[i, j // 64]) >> (j % 64)) & 1)


def visible(slot, col, gt, n_tree, bitmask):
    """May the query at absolute slot `slot` attend key slot `col`?

    Restates the strided window of the doubled mask (Tree/SpecTree.py:57-58): rows < gt are
    causal (Tree/Tree.py:27); tree rows see the whole prefix (incl. the root at gt-1) plus their
    ancestors (SpecTree.py:54); columns >= gt+n-1 fall in the masked quadrant."""
    tot = gt + n_tree - 1
    if col >= tot:
        return False
    if slot < gt:
        return col <= slot
    t = slot - (gt - 1)
    if t >= n_tree:
        return False
    if col < gt:
        return True
    return bit(bitmask, t, col - (gt - 1))


def tree_mask_dense(q_slot0, q_len, n_cols, gt, n_tree, bitmask):
    """Rows [q_slot0, q_slot0+q_len) x cols [0, n_cols) of the reference's additive mask."""
    out = np.full((q_len, n_cols), F16_MIN, dtype=np.float16)
    for i in range(q_len):
        for c in range(n_cols):
            if visible(q_slot0 + i, c, gt, n_tree, bitmask):
                out[i, c] = 0
    return out


# --------------------------------------------------------------------------------------------
# a5. KV slots (Engine/Llama_KV.py:60-94)
# --------------------------------------------------------------------------------------------
def kv_scatter(k_layer, v_layer, new_k, new_v, storage_ids):
    """update_kv_cache (Llama_KV.py:84-85): cache[h, storage_ids[i]] = new[h, i]."""
    ids = np.asarray(storage_ids, dtype=np.int64)
    k_layer[:, ids, :] = new_k
    v_layer[:, ids, :] = new_v


def kv_compact(k_cache, v_cache, slots, dst_offset, zero_end):
    """gather_kv_incremental (Llama_KV.py:60-68) on caches shaped [L, H, M, D] (batch squeezed).

    zero_end = M reproduces the reference's full-tail clear."""
    slots = [int(s) for s in slots]
    n = len(slots)
    for c in (k_cache, v_cache):
        if n:
            c[..., dst_offset:dst_offset + n, :] = c[..., slots, :].copy()
        if zero_end > dst_offset + n:
            c[..., dst_offset + n:zero_end, :] = 0


def kv_clear(k_cache, v_cache, used_rows):
    k_cache[..., :used_rows, :] = 0
    v_cache[..., :used_rows, :] = 0


# --------------------------------------------------------------------------------------------
# a3/a4. RoPE + tree-batched attention (Engine/Llama_modules.py:87-140,182-258;
#        Engine/offload_engine.py:35-67 for the 4.36 apply_rotary_pos_emb semantics)
# --------------------------------------------------------------------------------------------
def rope_tables(dim, max_pos, base=10000.0):
    """LlamaRotaryEmbedding_FI (Llama_modules.py:17-45): fp32 tables cast to fp16."""
    inv_freq = (1.0 / (np.float32(base) ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))).astype(np.float32)
    t = np.arange(max_pos, dtype=np.float32)
    freqs = np.outer(t, inv_freq).astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return h(np.cos(emb)), h(np.sin(emb))


def rope_apply(x, cos, sin, position_ids):
    """x: fp16 [H, q, D].  q_embed = (q*cos) + (rotate_half(q)*sin), fp16 after every op."""
    c = cos[np.asarray(position_ids)][None, :, :]
    s = sin[np.asarray(position_ids)][None, :, :]
    half = x.shape[-1] // 2
    rot = np.concatenate([-x[..., half:], x[..., :half]], axis=-1)
    return h(f(h(f(x) * f(c))) + f(h(f(rot) * f(s))))


def rope_kv_write(qkv, n_heads, h_kv, d, cos, sin, position_ids, storage_ids, k_layer, v_layer):
    """Split packed qkv [q, (H+2Hkv)*D], rotate q and k, scatter k/v; returns q_rot [H, q, D]."""
    q_len = qkv.shape[0]
    q = qkv[:, :n_heads * d].reshape(q_len, n_heads, d).transpose(1, 0, 2)
    k = qkv[:, n_heads * d:(n_heads + h_kv) * d].reshape(q_len, h_kv, d).transpose(1, 0, 2)
    v = qkv[:, (n_heads + h_kv) * d:].reshape(q_len, h_kv, d).transpose(1, 0, 2)
    q_rot = rope_apply(q, cos, sin, position_ids)
    k_rot = rope_apply(k, cos, sin, position_ids)
    kv_scatter(k_layer, v_layer, k_rot, v, storage_ids)
    return q_rot


def tree_attention(q, k_layer, v_layer, kv_len, scale, mask_add):
    """softmax(q k^T * scale + mask) v in fp32, output fp16 [q_len, H*D].

    q: fp16 [H, q_len, D]; k/v_layer: fp16 [H_kv, M, D]; mask_add: [q_len, kv_len] additive
    (0 / -65504).  fp32 reference of Llama_modules.py:228-248 (TG) and :127-134 (FI/SDPA)."""
    n_heads, q_len, d = q.shape
    h_kv = k_layer.shape[0]
    grp = n_heads // h_kv
    out = np.zeros((q_len, n_heads * d), dtype=np.float32)
    m = f(mask_add)
    for hh in range(n_heads):
        k = f(k_layer[hh // grp, :kv_len])
        v = f(v_layer[hh // grp, :kv_len])
        s = (f(q[hh]) @ k.T) * np.float32(scale) + m
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p = p / p.sum(-1, keepdims=True)
        out[:, hh * d:(hh + 1) * d] = p @ v
    return h(out)


# --------------------------------------------------------------------------------------------
# shared fp16 softmax (torch.nn.functional.softmax on an fp16 tensor)
# --------------------------------------------------------------------------------------------
def scaled_softmax_f16(x16, temperature):
    """softmax(x / T, dim=-1) for fp16 x: y = h(x/T); out = h(exp(y - max) / sum)."""
    y = h(f(x16) / np.float32(temperature))
    yf = f(y)
    m = yf.max(-1, keepdims=True)
    with np.errstate(invalid="ignore"):
        e = np.exp(yf - m).astype(np.float32)
    z = e.sum(-1, keepdims=True, dtype=np.float32)
    return h(e / z)


def _desc_order_with_ties(keys16):
    """indices sorted by key descending, ties broken by lower index (NaN treated as largest,
    which is torch.topk's rule; cannot occur for valid inputs)."""
    k = f(keys16).copy()
    k[np.isnan(k)] = np.inf
    return np.lexsort((np.arange(k.shape[0]), -k))


# --------------------------------------------------------------------------------------------
# a2. draft expansion samplers (utils.py:10-18, 29-32; Tree/SpecTree.py:103-104)
# --------------------------------------------------------------------------------------------
def sample_keys(logits16, rand16, temperature):
    """key = rand.log() / softmax(logits/T)  (utils.py:16-17), all fp16-rounded."""
    qd = scaled_softmax_f16(logits16, temperature)
    with np.errstate(divide="ignore", invalid="ignore"):
        lu = h(np.log(f(rand16)))
        return h(f(lu) / f(qd))


def sample_wor(logits16, rand16, k, temperature):
    """Rows [n, V] -> int64 [n, k]: the k largest keys per row, descending."""
    keys = sample_keys(logits16, rand16, temperature)
    out = np.zeros((keys.shape[0], k), dtype=np.int64)
    for r in range(keys.shape[0]):
        out[r] = _desc_order_with_ties(keys[r])[:k]
    return out


def topk_ids(logits16, k):
    """sampling_argmax (utils.py:29-32): top-k of the raw logits, descending."""
    out = np.zeros((logits16.shape[0], k), dtype=np.int64)
    for r in range(logits16.shape[0]):
        out[r] = _desc_order_with_ties(logits16[r])[:k]
    return out


def gather_branches(samples, branches):
    """new_tokens_set[sample_gather_indices] (SpecTree.py:104; tests/testbed.py:277-285):
    the first branches[j] samples of row j, rows concatenated."""
    return np.concatenate([samples[j, :b] for j, b in enumerate(branches)]) if len(branches) else np.zeros(0, np.int64)


# --------------------------------------------------------------------------------------------
# a6/a7. stochastic verification (Tree/SpecTree.py:136-157,196-227; utils.py:5-8)
# --------------------------------------------------------------------------------------------
_TWO24 = np.float64(1 << 24)


def _grid_int(p16):
    """fp16 values in [0, 1] as exact integers on the 2^-24 grid."""
    return np.round(p16.astype(np.float64) * _TWO24).astype(np.int64)


def round_int_to_f16(total):
    """h(total * 2^-24) computed exactly (total is an exact integer, float64 holds it)."""
    return np.float16(np.float64(total) / _TWO24)


def residual_f16(p16, q16):
    """get_residual (utils.py:5-8) with the normaliser summed exactly.
    Returns (residual fp16[V], had_nan)."""
    d = h(f(p16) - f(q16))
    d = np.where(d > 0, d, np.float16(0)).astype(np.float16)
    s = round_int_to_f16(int(_grid_int(d).sum()))
    with np.errstate(divide="ignore", invalid="ignore"):
        res = h(f(d) / np.float32(s))
    return res, bool(np.isnan(res).any())


def inverse_cdf(p16, u24):
    """Exact inverse CDF on the fp16 grid: smallest i with cumsum(w)[i] > (u24*total)>>24."""
    w = _grid_int(np.where(np.isnan(p16), np.float16(0), p16))
    total = int(w.sum())
    if total <= 0:
        return -1
    thr = (int(u24) * total) >> 24
    c = np.cumsum(w)
    return int(np.searchsorted(c, thr, side="right"))


def accept_children(p16, draft_row16, child_tokens, child_r16, temperature, margins=None, replace=False, probe=False):
    """accept_step (Tree/SpecTree.py:136-157) for one parent; replace=True is SpecInfer's rule
    (Tree/SpecInferTree.py:141-164): p >= r q, and a rejection leaves q and the draft logits untouched.

    p16: target distribution at the parent (fp16[V]); draft_row16: the parent's draft logits
    (fp16[V], mutated in place like the reference: rejected tokens get -65504);
    child_tokens / child_r16: token id and r[slot] of each child in Successors order.
    Returns (index of the first accepted child or -1, p16 after the rejections, n_rejected)."""
    n_rej = 0
    for j, (tok, r) in enumerate(zip(child_tokens, child_r16)):
        q16 = scaled_softmax_f16(draft_row16, temperature)
        if probe:
            # SpecTreeTest.accept_step (Tree/SpecTree.py:412): r is fp32, r * q[token] and the comparison are fp32
            rq32 = np.float32(r) * np.float32(q16[tok])
            if margins is not None:
                margins.append(float(np.float32(p16[tok]) - rq32))
            ok = np.float32(p16[tok]) >= rq32
        else:
            rq = h(f(np.float16(r)) * f(q16[tok]))
            if margins is not None:
                margins.append(float(f(p16[tok]) - f(rq)))
            ok = (p16[tok] >= rq) if replace else (p16[tok] > rq)
        if ok:
            return j, p16, n_rej
        p16, _ = residual_f16(p16, q16)
        if not replace:
            draft_row16[tok] = F16_MIN
        n_rej += 1
    return -1, p16, n_rej


def verify_stochastic(target_logits16, draft_logits16, tokens, r16, successors, gt, temperature,
                      u24, margins=None, replace=False, gather_first=False, probe=False):
    """SpecTree.verify from the softmax to the token compaction (Tree/SpecTree.py:196-224).

    target_logits16: [n, V]; draft_logits16: [>=n, V] tree-local rows (mutated); tokens: int64[M]
    (mutated: compaction + bonus); r16: fp16[M].  Returns the SQ_RES_* record as a dict."""
    n = len(successors)
    p_all = scaled_softmax_f16(target_logits16[:n], temperature)
    node = 0
    slots = []
    terminal, reason = False, 0
    while True:
        ch = successors[node]
        if len(ch) == 0:
            p = p_all[node]
            break
        ctoks = [int(tokens[c + gt - 1]) for c in ch]
        crs = [r16[c + gt - 1] for c in ch]
        j, p, _ = accept_children(p_all[node], draft_logits16[node], ctoks, crs, temperature, margins, replace, probe)
        if j < 0:
            break
        node = ch[j]
        slots.append(node + gt - 1)
        if int(tokens[node + gt - 1]) in EOS_IDS:
            terminal, reason = True, 1
            break
    a = gt + len(slots)
    bonus = -1
    if not terminal:
        if np.isnan(p).any():
            terminal, reason = True, 2
        else:
            bonus = inverse_cdf(p, u24)
            if bonus < 0:
                terminal, reason = True, 2
    # the reference stores the bonus token BEFORE the gather (Tree/SpecTree.py:222-224): an accepted node sitting at
    # slot a is therefore committed with the bonus token's id (quirk reproduced for token parity)
    # (gather_first = the lossless order, SQ_VERIFY_GATHER_FIRST)
    if not terminal and not gather_first:
        tokens[a] = bonus
    if slots:
        tokens[gt:a] = tokens[np.asarray(slots)].copy()
    if not terminal and gather_first:
        tokens[a] = bonus
    return dict(accept_len=a, n_tree=len(slots), bonus=bonus, terminal=int(terminal), reason=reason,
                gt=gt, last_node=node, slots=slots, final_p=p)


# --------------------------------------------------------------------------------------------
# a6. nucleus (top-p) filter on the target logits (utils.py:65-77, called at Tree/SpecTree.py:196)
# --------------------------------------------------------------------------------------------
def top_p_filter(logits16, top_p, temperature):
    """Rows [n, V] fp16 -> copy with the removed tokens set to -inf.

    Reference (utils.py:65-77): sort descending; cumulative_probs = cumsum(softmax(sorted / T)); the token at sorted
    rank k is removed iff cumulative_probs[k-1] > top_p.  Restated with torch's CPU arithmetic for fp16 tensors:
      * the softmax runs over the SORTED row (same values, fp32 sum in that order) and is rounded to fp16;
      * torch.cumsum on a CPU fp16 tensor accumulates SEQUENTIALLY in fp32 (at::acc_type<Half, false> = float,
        aten/src/ATen/native/cpu/ReduceOpsKernel.cpp cumsum_cpu_kernel) and rounds every prefix to fp16.  Every fp16
        value is a multiple of 2^-24, so every prefix below 1 is exactly representable in fp32: the sequential fp32
        cumsum IS the exact sum up to the cut (and beyond 1 every comparison is true whatever the rounding).  The
        "exact mass" rule of the native kernel (sq_top_p_filter_f16) is therefore the reference's rule, not an
        approximation of it;
      * `cumulative_probs > top_p` compares in fp16 (the python scalar is cast to the tensor dtype);
      * torch.sort: equal logits are ordered by ascending token id here.  torch's CPU sort of fp16 is NOT stable
        (x86-simd-sort on AVX-512 FP16 hosts; `stable=False` is the default), so which of several exactly equal logits
        sit before the cut is implementation-defined in the reference.  Equal logits carry equal probability: the
        filtered distribution is the same up to relabelling those tokens.
    Against the reference itself (tests/test_oracle_golden.py, tests/test_oracle_live_reference_cpu.py): identical on every
    row except for the identity of equal-logit tokens at the cut.  A probability whose exp() differs from torch's in the
    last fp32 ulp could additionally move the cut by that element's fp16 ulp of mass; not observed on the test rows."""
    out = logits16.copy()
    th16 = np.float16(top_p)
    V = logits16.shape[1]
    for r in range(logits16.shape[0]):
        xf = f(logits16[r]).copy()
        xf[np.isnan(xf)] = np.inf
        order = np.lexsort((np.arange(V), -xf))                       # stable descending sort
        p16 = scaled_softmax_f16(logits16[r][order], temperature)      # softmax of the sorted row
        p32 = np.where(np.isnan(p16), np.float16(0), p16).astype(np.float32)
        c16 = np.cumsum(p32, dtype=np.float32).astype(np.float16)      # sequential fp32 accumulation, fp16 prefixes
        filt = c16 > th16
        remove = np.zeros(V, dtype=bool)
        remove[1:] = filt[:-1]
        out[r, order[remove]] = np.float16(-np.inf)
    return out


def top_p_mass_difference(logits16, a16, b16, temperature):
    """Per row: the probability mass (under softmax(logits / T)) of the tokens that exactly one of the two filtered
    copies a16 / b16 removed -- the measure in which two nucleus filters are compared (a token count says nothing: the
    cut sits in a tail of ~1e-5-mass tokens)."""
    p = scaled_softmax_f16(logits16, temperature).astype(np.float64)
    diff = np.isinf(f(a16)) != np.isinf(f(b16))
    return np.where(diff, p, 0.0).sum(axis=1), diff.sum(axis=1)


# --------------------------------------------------------------------------------------------
# a8. greedy verification (Tree/GreedyTree.py:131-146,186-209)
# --------------------------------------------------------------------------------------------
def argmax_first(x16):
    """argmax(-1) with ties -> lowest index."""
    xf = f(x16)
    return np.argmax(xf, axis=-1).astype(np.int64)


def verify_greedy(target_logits16, tokens, successors, gt):
    n = len(successors)
    tgt = argmax_first(target_logits16[:n])
    node = 0
    slots = []
    terminal, reason = False, 0
    while True:
        nxt = -1
        for c in successors[node]:
            if int(tokens[c + gt - 1]) == int(tgt[node]):
                nxt = c
                break
        if nxt < 0:
            break
        node = nxt
        slots.append(node + gt - 1)
        if int(tokens[node + gt - 1]) in EOS_IDS:
            terminal, reason = True, 1
            break
    a = gt + len(slots)
    if slots:
        tokens[gt:a] = tokens[np.asarray(slots)].copy()
    bonus = -1
    if not terminal:
        bonus = int(tgt[node])
        tokens[a] = bonus
    return dict(accept_len=a, n_tree=len(slots), bonus=bonus, terminal=int(terminal), reason=reason,
                gt=gt, last_node=node, slots=slots, target_token=tgt)


# ---- tall-skinny linear layers (sequoia_amd/csrc/ts_linear.hip) ------------------------------------------
# Restates nn.Linear as the reference uses it (Engine/Llama_modules.py:104-112,138,256,262-271: fp16 weights and
# activations, fp32 accumulation inside the GEMM, fp16 output) plus the operand images the kernel reads.
def frag_rows(x16, mtp=None):
    """[m, k] -> fragment-major image [k/32, mtp, 64, 8]: lane = (k/8 % 4) * 16 + row % 16; rows >= m are zero."""
    m, k = x16.shape
    mtp = mtp or (m + 15) // 16
    pad = np.zeros((mtp * 16, k), dtype=x16.dtype)
    pad[:m] = x16
    # [mt, r, kb, g, j] -> [kb, mt, g, r, j]
    return np.ascontiguousarray(pad.reshape(mtp, 16, k // 32, 4, 8).transpose(2, 0, 3, 1, 4)).reshape(k // 32, mtp, 64, 8)


def unfrag_rows(xf, m, k):
    mtp = xf.shape[1]
    full = xf.reshape(k // 32, mtp, 4, 16, 8).transpose(1, 3, 0, 2, 4).reshape(mtp * 16, k)
    return np.ascontiguousarray(full[:m])


def frag_weight(w16):
    """[n, k] -> [n/16, k/32, 64, 8] (weights: tile-major)."""
    n, k = w16.shape
    return np.ascontiguousarray(w16.reshape(n // 16, 16, k // 32, 4, 8).transpose(0, 2, 3, 1, 4)).reshape(n // 16, k // 32, 64, 8)


def silu_mul_f16(g16, u16):
    """h(h(silu(g)) * u)  (LlamaMLP_FI: act_fn(gate_proj(x)) * up_proj(x), fp16 tensors)."""
    gf = f(g16)
    s = h(gf / (np.float32(1.0) + np.exp(-gf, dtype=np.float32)))
    return h(f(s) * f(u16))


def linear_f16(a16, w16, res16=None, silu=False):
    """a [m, k] . w [n, k]^T with fp32 accumulation and fp16 output; silu: w = [gate rows | up rows];
    res: h(h(acc) + res)."""
    acc = f(a16) @ f(w16).T
    if silu:
        n = w16.shape[0] // 2
        return silu_mul_f16(h(acc[:, :n]), h(acc[:, n:]))
    out = h(acc)
    if res16 is not None:
        out = h(f(out) + f(res16))
    return out


def add_rmsnorm_slabs(slabs32, res16, weight16, eps):
    """slabs [splits, m, n] fp32 -> (sum_out, normed): x = h(((s0 + s1) + s2) ...), sum = h(x + res), RMSNorm as
    LlamaRMSNorm_FI (Engine/Llama_modules.py:282-288)."""
    acc = slabs32[0].astype(np.float32)
    for s in range(1, slabs32.shape[0]):
        acc = acc + slabs32[s]
    total = h(f(h(acc)) + f(res16))
    xf = f(total)
    var = (xf * xf).mean(-1, keepdims=True, dtype=np.float32)
    nrm = h(xf * (np.float32(1.0) / np.sqrt(var + np.float32(eps))))
    return total, h(f(weight16) * f(nrm))


# ---- the paper's comparison baselines ---------------------------------------------------
def sample_iid(logits16, u24, k, temperature):
    """SpecInferTree.collective_grow_static (Tree/SpecInferTree.py:104-109): k draws with replacement per row from
    softmax(logits / T); draw j of row i = exact inverse CDF at u24[i][j] / 2^24 (explicit uniforms in place of
    torch's device multinomial stream).  Returns int64 [n, k]."""
    q = scaled_softmax_f16(logits16, temperature)
    out = np.zeros((q.shape[0], k), dtype=np.int64)
    for i in range(q.shape[0]):
        for j in range(k):
            out[i, j] = inverse_cdf(q[i], int(u24[i][j]))
    return out


def verify_specinfer(target_logits16, draft_logits16, tokens, r16, successors, gt, temperature, u24, margins=None):
    """SpecInferTree.verify (Tree/SpecInferTree.py:167-247): SpecTree's walk with the with-replacement accept rule."""
    return verify_stochastic(target_logits16, draft_logits16, tokens, r16, successors, gt, temperature, u24, margins,
                             replace=True)


def verify_tokens(target_tokens, tokens, successors, gt):
    """GreedySTree.verify after the per-node target draw (Tree/GreedySTree.py:196-214): walk by token equality against
    target_tokens[node]; bonus = target token of the last accepted node."""
    node, slots, terminal, reason = 0, [], False, 0
    while True:
        nxt = -1
        for c in successors[node]:
            if int(tokens[c + gt - 1]) == int(target_tokens[node]):
                nxt = c
                break
        if nxt < 0:
            break
        node = nxt
        slots.append(node + gt - 1)
        if int(tokens[node + gt - 1]) in EOS_IDS:
            terminal, reason = True, 1
            break
    a = gt + len(slots)
    bonus = -1 if terminal else int(target_tokens[node])
    if slots:
        tokens[gt:a] = tokens[np.asarray(slots)].copy()
    if not terminal:
        tokens[a] = bonus
    return dict(accept_len=a, n_tree=len(slots), bonus=bonus, terminal=int(terminal), reason=reason, gt=gt,
                last_node=node, slots=slots)


# ---- the acceptance-rate probe: SpecTreeTest, Tree/SpecTree.py:283-481 -------------------------------
def sample_wor_f32noise(logits16, rand32, k, temperature):
    """SpecTreeTest.collective_grow_static (Tree/SpecTree.py:349-360): the noise is fp32, so torch evaluates
    rand.log() / softmax(logits / T) in fp32 (fp32 / fp16 promotes); the k largest keys per row, descending."""
    q = scaled_softmax_f16(logits16, temperature)
    with np.errstate(divide="ignore", invalid="ignore"):
        keys = np.log(rand32.astype(np.float32)).astype(np.float32) / q.astype(np.float32)
    out = np.zeros((keys.shape[0], k), dtype=np.int64)
    for r in range(keys.shape[0]):
        kr = keys[r].copy()
        kr[np.isnan(kr)] = np.inf
        out[r] = np.lexsort((np.arange(kr.shape[0]), -kr))[:k]
    return out, keys


def verify_probe(target_logits16, draft_logits16, tokens, r32, successors, gt, temperature, u24, margins=None):
    """SpecTreeTest.verify (Tree/SpecTree.py:419-481): Sequoia's walk with fp32 r and the fp32 test p >= r q; the accepted
    tokens are gathered before the bonus token is appended (:472-474)."""
    return verify_stochastic(target_logits16, draft_logits16, tokens, r32, successors, gt, temperature, u24, margins,
                             gather_first=True, probe=True)
