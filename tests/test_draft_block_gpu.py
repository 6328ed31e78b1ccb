"""GPU parity of the fused draft attention block (csrc/draft_block.hip: qkv projection + RoPE + KV write + tree attention +
o_proj partials of one layer in ONE launch) against the four-launch sequence it replaces and against the numpy oracle.

Bar: the K / V rows written to the cache are bit-exact on operands whose partial sums are exact in fp32 (the two paths
differ only in fp32 summation order) and within one fp16 ulp on random operands; the o_proj output is within 4e-3 of the
four-launch result relative to its magnitude; a forward of a small draft with the block on equals the forward with it off
within the logit tolerance of the trace tests, and its top-8 sets agree.
"""
import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sequoia_amd.ops import HipOps
    o = HipOps()
    assert o.lib.sq_device_ready() == 1, "no gfx950 device visible"
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def level_tree(rng, sizes):
    """BFS-ordered tree with the given level sizes (level 0 = the root): children contiguous and ascending, like
    tree_search.py's output.  Returns (successors, first node id of every level)."""
    n = int(sum(sizes))
    succ = [[] for _ in range(n)]
    first = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    for lv in range(1, len(sizes)):
        parents = np.sort(rng.randint(first[lv - 1], first[lv], size=sizes[lv]))
        for j, p in enumerate(parents):
            succ[int(p)].append(int(first[lv] + j))
    return succ, first


def _case(rng, H, hidden, M, gt, sizes, level, exact=False):
    D = 64
    succ, first = level_tree(rng, sizes)
    n = len(succ)
    bm = O.bitmask_from_successors(succ)
    a, b = int(first[level]), int(first[level + 1])
    q_len = b - a
    q_slot0 = gt - 1 + a
    depth = np.zeros(n, np.int64)
    for p, ch in enumerate(succ):
        for c in ch:
            depth[c] = depth[p] + 1
    pos = (depth[a:b] + gt - 1).astype(np.int64)
    sid = (q_slot0 + np.arange(q_len)).astype(np.int64)
    if exact:       # every partial sum of the projections is exact in fp32: the two summation orders give the same bits
        x = (rng.randint(-4, 5, (q_len, hidden)) * 0.25).astype(np.float16)
        wqkv = (rng.randint(-2, 3, (3 * H * D, hidden)) * 0.125).astype(np.float16)
    else:
        x = rng.randn(q_len, hidden).astype(np.float16)
        wqkv = (rng.randn(3 * H * D, hidden) * 0.04).astype(np.float16)
    wo = (rng.randn(hidden, H * D) * 0.04).astype(np.float16)
    k = (rng.randn(H, M, D)).astype(np.float16)
    v = (rng.randn(H, M, D)).astype(np.float16)
    k[:, q_slot0:] = 0
    v[:, q_slot0:] = 0
    cos, sin = O.rope_tables(D, 512)
    return dict(H=H, hidden=hidden, M=M, gt=gt, n=n, bm=bm, q_len=q_len, q_slot0=q_slot0, pos=pos, sid=sid, x=x, wqkv=wqkv,
                wo=wo, k=k, v=v, cos=cos, sin=sin)


def _unfused(ops, c):
    """The launch sequence the block replaces, on the same operands."""
    H, hidden, q, D = c["H"], c["hidden"], c["q_len"], 64
    a_f = ops.repack_rows(dev(c["x"]))
    qkv = torch.empty((q, 3 * H * D), dtype=torch.float16, device=DEV)
    ops.linear_ts(a_f, ops.repack_weight(dev(c["wqkv"])), q, 3 * H * D, hidden, out=qkv, tiles=3 * H * D // 16 // 2)
    k, v = dev(c["k"]), dev(c["v"])
    q_rot = torch.empty((H, q, D), dtype=torch.float16, device=DEV)
    ops.rope_kv_write(qkv, q_rot, k, v, dev(c["cos"]), dev(c["sin"]), dev(c["pos"]), dev(c["sid"]), H, H, D)
    attn = torch.empty((q, H * D), dtype=torch.float16, device=DEV)
    ops.tree_attention(q_rot, k, v, attn, c["q_slot0"] + q, D ** -0.5, q_slot0=c["q_slot0"], gt=c["gt"], n_tree=c["n"],
                       bitmask=dev(c["bm"].view(np.int64)))
    o = torch.empty((q, hidden), dtype=torch.float16, device=DEV)
    ops.linear_ts(ops.repack_rows(attn), ops.repack_weight(dev(c["wo"])), q, hidden, H * D, out=o, tiles=hidden // 16)
    return k, v, attn, o


def _fused(ops, c, kv_only=False, ctx=None, host_slot0=None, host_gt=None):
    H, hidden, q, D = c["H"], c["hidden"], c["q_len"], 64
    a_f = ops.repack_rows(dev(c["x"]))
    k, v = dev(c["k"]), dev(c["v"])
    slab = torch.full((H, q, hidden), float("nan"), dtype=torch.float32, device=DEV)
    ops.draft_attn_block(a_f, ops.repack_weight(dev(c["wqkv"])), None if kv_only else ops.repack_weight(dev(c["wo"])),
                         None if kv_only else slab, k, v, dev(c["cos"]), dev(c["sin"]), dev(c["pos"]), dev(c["sid"]), q, H, D,
                         hidden, D ** -0.5, c["q_slot0"] if host_slot0 is None else host_slot0,
                         c["gt"] if host_gt is None else host_gt, c["n"], bitmask=dev(c["bm"].view(np.int64)), ctx=ctx,
                         kv_only=kv_only)
    return k, v, slab


def _rows_close(a, b, ulps, own=False):
    a, b = a.float(), b.float()
    assert torch.isfinite(b).all()
    scale = a.abs() if own else a.abs().amax(dim=-1, keepdim=True)
    tol = ulps * torch.clamp(scale, min=2.0 ** -10) * 2.0 ** -10
    bad = (a - b).abs() > tol
    assert not bad.any(), float(((a - b).abs() / tol).max())


CASES = [
    # H, hidden, M, gt, level sizes, level
    (12, 768, 384, 130, [1, 8, 34, 40, 45], 2),      # config B's draft: a 34-row level (3 row tiles)
    (12, 768, 384, 130, [1, 8, 34, 40, 45], 1),      # 8 rows
    (12, 768, 384, 250, [1, 8, 34, 40, 45], 4),      # 45 rows, kv range 332 slots
    (12, 768, 384, 37, [1, 5, 20], 0),               # the root alone (one row, a tree row with tnode 0 == committed text)
    (16, 1024, 256, 64, [1, 16, 48, 63], 3),         # hidden 1024: 16 heads, 63 rows
    (8, 512, 512, 200, [1, 30, 100, 116, 120], 3),   # 8 bitmask words (367 nodes), 116 rows: the S512 growmap's widest level
]


@pytest.mark.parametrize("H,hidden,M,gt,sizes,level", CASES)
def test_block_matches_unfused_sequence(ops, H, hidden, M, gt, sizes, level):
    rng = np.random.RandomState(H + level * 7 + gt)
    c = _case(rng, H, hidden, M, gt, sizes, level)
    k0, v0, attn0, o0 = _unfused(ops, c)
    k1, v1, slab = _fused(ops, c)
    torch.cuda.synchronize()
    # K / V rows: h(sum) of the same products in another order: equal up to one fp16 ulp; rows outside the level untouched
    q0, q1 = c["q_slot0"], c["q_slot0"] + c["q_len"]
    assert torch.equal(k1[:, :q0], dev(c["k"])[:, :q0]) and torch.equal(k1[:, q1:], dev(c["k"])[:, q1:])
    assert torch.equal(v1[:, :q0], dev(c["v"])[:, :q0]) and torch.equal(v1[:, q1:], dev(c["v"])[:, q1:])
    # (V: one ulp of the value; K: the rotation x1 cos - x2 sin can cancel, so one ulp of its INPUTS = of the row's magnitude)
    _rows_close(v0[:, q0:q1], v1[:, q0:q1], 1.01, own=True)
    _rows_close(k0[:, q0:q1], k1[:, q0:q1], 2.0)
    o1 = slab.sum(0)
    assert torch.isfinite(o1).all()
    ref = o0.float()
    err = float((o1.half().float() - ref).abs().max())
    assert err <= 4e-3 * max(1.0, float(ref.abs().max())), (err, float(ref.abs().max()))


@pytest.mark.parametrize("H,hidden,M,gt,sizes,level", CASES[:3] + CASES[4:5])
def test_block_kv_rows_bit_exact_on_exact_operands(ops, H, hidden, M, gt, sizes, level):
    rng = np.random.RandomState(11 * H + level)
    c = _case(rng, H, hidden, M, gt, sizes, level, exact=True)
    k0, v0, attn0, o0 = _unfused(ops, c)
    k1, v1, slab = _fused(ops, c)
    k2, v2, _ = _fused(ops, c, kv_only=True)
    torch.cuda.synchronize()
    assert torch.equal(k0, k1) and torch.equal(v0, v1)
    assert torch.equal(k0, k2) and torch.equal(v0, v2)


@pytest.mark.parametrize("H,hidden,M,gt,sizes,level", CASES[:1] + CASES[5:])
def test_block_against_oracle_attention(ops, H, hidden, M, gt, sizes, level):
    """Attention + o_proj of the block against the numpy oracle evaluated on the block's own K / V / rotated q rows."""
    rng = np.random.RandomState(5 * H + level)
    c = _case(rng, H, hidden, M, gt, sizes, level)
    k0, v0, attn0, o0 = _unfused(ops, c)
    k1, v1, slab = _fused(ops, c)
    torch.cuda.synchronize()
    q_len, q_slot0, D = c["q_len"], c["q_slot0"], 64
    kv_len = q_slot0 + q_len
    # rotated queries recomputed by the oracle from the fp16 projection rows
    qkv = O.linear_f16(c["x"], c["wqkv"])
    kk, vv = np.zeros((H, M, D), np.float16), np.zeros((H, M, D), np.float16)
    q_rot = O.rope_kv_write(qkv, H, H, D, c["cos"], c["sin"], c["pos"], c["sid"], kk, vv)
    mask = O.tree_mask_dense(q_slot0, q_len, kv_len, c["gt"], c["n"], c["bm"])
    # the level's rows see the cached keys and themselves only
    blk = mask[:, q_slot0:kv_len]
    assert (np.diag(blk) == 0).all() and (blk[~np.eye(q_len, dtype=bool)] < 0).all()
    want = O.tree_attention(q_rot, k1.cpu().numpy(), v1.cpu().numpy(), kv_len, D ** -0.5, mask).astype(np.float32)
    want_o = want @ c["wo"].astype(np.float32).T
    got_o = slab.sum(0).cpu().numpy()
    scale = max(1.0, float(np.abs(want_o).max()))
    assert np.abs(got_o - want_o).max() <= 6e-3 * scale, (np.abs(got_o - want_o).max(), scale)


def test_block_reads_context_from_device(ops):
    """hipGraph-replayable form: {q_slot0, gt} come from the device context block, the host arguments are ignored."""
    rng = np.random.RandomState(3)
    c = _case(rng, 12, 768, 384, 130, [1, 8, 34, 40, 45], 2)
    k1, v1, slab1 = _fused(ops, c)
    ctx = torch.tensor([c["q_slot0"], c["gt"], c["q_slot0"] + c["q_len"]], dtype=torch.int32, device=DEV)
    k2, v2, slab2 = _fused(ops, c, ctx=ctx, host_slot0=0, host_gt=1)
    torch.cuda.synchronize()
    assert torch.equal(k1, k2) and torch.equal(v1, v2) and torch.equal(slab1, slab2)


def test_block_rejects_unsupported_shapes(ops):
    from sequoia_amd.native import SequoiaNativeError
    rng = np.random.RandomState(4)
    c = _case(rng, 12, 768, 384, 130, [1, 8, 34], 1)
    a_f = ops.repack_rows(dev(c["x"]))
    k, v = dev(c["k"]), dev(c["v"])
    slab = torch.zeros((12, c["q_len"], 768), dtype=torch.float32, device=DEV)
    with pytest.raises(SequoiaNativeError, match="unsupported"):
        ops.draft_attn_block(a_f, ops.repack_weight(dev(c["wqkv"])), ops.repack_weight(dev(c["wo"])), slab, k, v,
                             dev(c["cos"]), dev(c["sin"]), dev(c["pos"]), dev(c["sid"]), c["q_len"], 6, 128, 768, 0.1,
                             c["q_slot0"], c["gt"], c["n"], bitmask=dev(c["bm"].view(np.int64)))


def test_draft_forward_block_on_equals_off():
    """A 68m-dims draft forward over one tree level with the block on against the same forward with it off: logits within
    the trace tests' tolerance, identical top-8 sets per row, identical KV cache rows up to one fp16 ulp."""
    from sequoia_amd.Engine import ts_linear
    from sequoia_amd.Engine.Engine import InferenceEngine
    from sequoia_amd.Engine.Llama_modules import TreeContext
    rng = np.random.RandomState(9)
    M, gt = 384, 130
    succ, first = level_tree(rng, [1, 8, 34, 40, 45])
    n = len(succ)
    bm = dev(O.bitmask_from_successors(succ).view(np.int64))
    a, b = int(first[2]), int(first[3])
    q_len, q_slot0 = b - a, gt - 1 + a
    outs = []
    for on in (False, True):
        ts_linear.DRAFT_BLOCK = on
        try:
            eng = InferenceEngine(max_length=M, model_name_or_path="random:JackFram/llama-68m:seed=5", device=DEV)
            g = torch.Generator().manual_seed(1)
            prompt = torch.randint(0, 32000, (1, q_slot0), generator=g).to(DEV)
            sid = torch.arange(M, device=DEV)
            ctx0 = TreeContext(q_slot0=0, gt=q_slot0, n_tree=1, bitmask=bm, kv_len=q_slot0, contiguous_slots=True)
            eng.model_run(input_ids=prompt, storage_ids=sid[:q_slot0], position_ids=sid[:q_slot0].unsqueeze(0),
                          attention_mask=None, tree=ctx0)
            ids = torch.randint(0, 32000, (1, q_len), generator=g).to(DEV)
            pos = torch.full((1, q_len), gt + 1, dtype=torch.long, device=DEV)
            tree = TreeContext(q_slot0=q_slot0, gt=gt, n_tree=n, bitmask=bm, kv_len=q_slot0 + q_len, contiguous_slots=True,
                               independent_rows=True)
            logits = eng.model_run(input_ids=ids, storage_ids=sid[q_slot0:q_slot0 + q_len], position_ids=pos,
                                   attention_mask=None, tree=tree)[0].float()
            kv = eng.kv_cache
            outs.append((logits, kv.k_cache[:, 0, :, q_slot0:q_slot0 + q_len].float().clone(),
                         kv.v_cache[:, 0, :, q_slot0:q_slot0 + q_len].float().clone()))
        finally:
            ts_linear.DRAFT_BLOCK = True
    (l0, k0, v0), (l1, k1, v1) = outs
    assert torch.isfinite(l1).all()
    assert float((l0 - l1).abs().max()) <= 4e-2, float((l0 - l1).abs().max())
    # layer 0 differs by summation order only (one ulp of the projection); layer 1 sees layer 0's attention output
    _rows_close(v0[0], v1[0], 1.01, own=True)
    _rows_close(k0[0], k1[0], 2.0)
    _rows_close(v0[1], v1[1], 16.0)
    _rows_close(k0[1], k1[1], 16.0)
    top0, top1 = l0.topk(8, dim=-1).indices.sort(dim=-1).values, l1.topk(8, dim=-1).indices.sort(dim=-1).values
    assert (top0 == top1).float().mean() > 0.97


# ---- level attention: RoPE + KV write + tree attention in one launch, any head dim --------------------------------------------
LA_CASES = [
    # H, Hkv, D, M, gt, level sizes, level, splits
    (16, 16, 128, 384, 140, [1, 8, 20, 35], 2, 2),       # configuration D's draft (Sheared-LLaMA-1.3B: 16 heads of 128), 20-row level
    (16, 16, 128, 384, 140, [1, 8, 20, 35], 3, 0),       # 35 rows (3 tiles), fp16 rows instead of partials
    (32, 32, 128, 1024, 700, [1, 64, 64], 1, 2),         # configuration E's draft (7B dims): 64-row level, long prefix
    (12, 12, 64, 384, 130, [1, 8, 34, 40, 45], 2, 3),    # 68m dims (what the fused block covers; the generic kernel must agree too)
    (8, 2, 128, 512, 200, [1, 30, 100, 116, 120], 3, 4), # GQA 4:1, 8 bitmask words, 116 rows
    (4, 1, 64, 256, 64, [1, 16], 0, 0),                  # one row (the root), one KV head
]


@pytest.mark.parametrize("H,Hkv,D,M,gt,sizes,level,splits", LA_CASES)
def test_level_attention_matches_two_launches(ops, H, Hkv, D, M, gt, sizes, level, splits):
    rng = np.random.RandomState(H * 3 + D + level)
    succ, first = level_tree(rng, sizes)
    n = len(succ)
    bm = dev(O.bitmask_from_successors(succ).view(np.int64))
    a, b = int(first[level]), int(first[level + 1])
    q_len, q_slot0 = b - a, gt - 1 + a
    stride = (H + 2 * Hkv) * D
    cos, sin = O.rope_tables(D, 1024)
    pos = dev(np.full(q_len, gt + level - 1, np.int64))
    sid = dev((q_slot0 + np.arange(q_len)).astype(np.int64))
    k = rng.randn(Hkv, M, D).astype(np.float16); v = rng.randn(Hkv, M, D).astype(np.float16)
    k[:, q_slot0:] = 0; v[:, q_slot0:] = 0
    if splits:
        slab = dev((rng.randn(splits, q_len, stride) * 0.7).astype(np.float32))
    else:
        rows = dev(rng.randn(q_len, stride).astype(np.float16))
    outs = []
    for fused in (False, True):
        kk, vv = dev(k), dev(v)
        for frag in (False, True):
            out = torch.zeros(ops.frag_shape(q_len, H * D) if frag else (q_len, H * D), dtype=torch.float16, device=DEV)
            if fused:
                ops.level_attention(None if splits else rows, out, kk, vv, dev(cos), dev(sin), pos, sid, H, Hkv, D, D ** -0.5,
                                    q_slot0, gt, n, bitmask=bm, out_frag=frag,
                                    qkv_slab=(slab, splits, q_len, stride) if splits else None)
            else:
                q_rot = torch.empty((H, q_len, D), dtype=torch.float16, device=DEV)
                if splits:
                    ops.rope_kv_write_slabs(slab, splits, stride, q_rot, kk, vv, dev(cos), dev(sin), pos, sid, H, Hkv, D)
                else:
                    ops.rope_kv_write(rows, q_rot, kk, vv, dev(cos), dev(sin), pos, sid, H, Hkv, D)
                ops.tree_attention(q_rot, kk, vv, out, q_slot0 + q_len, D ** -0.5, q_slot0=q_slot0, gt=gt, n_tree=n, bitmask=bm,
                                   out_frag=frag)
            outs.append((fused, frag, out, kk, vv))
    torch.cuda.synchronize()
    ref = {f: (o, kk, vv) for fu, f, o, kk, vv in outs if not fu}
    for fu, f, o, kk, vv in outs:
        if not fu:
            continue
        ro, rk, rv = ref[f]
        assert torch.equal(kk, rk) and torch.equal(vv, rv)              # same rounding points, same split order: bit-exact rows
        assert torch.isfinite(o).all()
        err = float((o.float() - ro.float()).abs().max())
        assert err <= 4e-3, err                                          # P rounded to fp16 against another running maximum
