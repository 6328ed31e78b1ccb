"""Tall-skinny linear layers (sq_linear_ts_f16 and the fragment-major producers) against the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import ops_np as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from sequoia_amd.ops import get_ops
    return get_ops()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _exact_operands(rng, m, n, k):
    """Small-integer operands scaled by powers of two: every partial sum is exact in fp32, so the result does not
    depend on the summation order and the comparison can be bit-exact."""
    a = (rng.integers(-4, 5, (m, k)) * 0.25).astype(np.float16)
    w = (rng.integers(-2, 3, (n, k)) * 0.125).astype(np.float16)
    return a, w


@pytest.mark.parametrize("m,k", [(1, 32), (16, 64), (34, 768), (48, 4096), (128, 256)])
def test_repack_rows_matches_oracle_layout(m, k):
    rng = np.random.default_rng(m * 7 + k)
    x = rng.standard_normal((m, k)).astype(np.float16)
    got = _ops().repack_rows(_t(x)).cpu().numpy()
    assert np.array_equal(got, O.frag_rows(x))
    assert np.array_equal(O.unfrag_rows(got, m, k), x)


@pytest.mark.parametrize("n,k", [(16, 32), (64, 96), (2304, 768), (4096, 11008)])
def test_repack_weight_matches_oracle_layout(n, k):
    rng = np.random.default_rng(n + k)
    w = rng.standard_normal((n, k)).astype(np.float16)
    got = _ops().repack_weight(_t(w)).cpu().numpy()
    assert np.array_equal(got, O.frag_weight(w))


CASES = [  # m, n_out, k, tiles, splits
    (1, 64, 256, 4, 1), (16, 64, 256, 2, 1), (19, 2304, 768, 48, 1), (34, 768, 3072, 48, 1), (48, 4096, 4096, 256, 1),
    (48, 4096, 4096, 64, 3), (64, 768, 768, 12, 2), (65, 4096, 1024, 128, 2), (96, 512, 2048, 32, 4), (128, 4096, 4096, 64, 4),
    (128, 12288, 1024, 256, 1), (128, 1000 * 16, 512, 500, 1),
    # 129 rows (64x2, 16x8, ...: 128 nodes + the root): 8 MFMA row tiles + the extra row on the vector ALU
    # (ts_linear_body<.., TAIL>); 65 rows (the 8x8 tree) stay on the 6-tile build
    (65, 12288, 512, 256, 1), (65, 4096, 4096, 64, 4), (65, 2048, 256, 16, 1), (129, 4096, 4096, 64, 4), (129, 10240, 1024, 128, 2),
    (129, 8192, 512, 128, 1), (129, 8192, 2048, 86, 8), (129, 64, 256, 4, 1),
]


@pytest.mark.parametrize("m,n,k,tiles,splits", CASES)
def test_linear_ts_bit_exact_on_order_independent_operands(m, n, k, tiles, splits):
    ops = _ops()
    rng = np.random.default_rng(m + n + k + tiles)
    a, w = _exact_operands(rng, m, n, k)
    af, wf = ops.repack_rows(_t(a)), ops.repack_weight(_t(w))
    ref32 = O.f(a) @ O.f(w).T
    if splits > 1:
        slab = torch.full((splits * m * n + 16,), float("nan"), dtype=torch.float32, device=DEV)
        ops.linear_ts(af, wf, m, n, k, tiles=tiles, splits=splits, slab=slab)
        parts = slab[:splits * m * n].reshape(splits, m, n).cpu().numpy()
        assert np.array_equal(parts.sum(0, dtype=np.float32), ref32)
        return
    out = torch.full((m, n), float("nan"), dtype=torch.float16, device=DEV)
    ops.linear_ts(af, wf, m, n, k, out=out, tiles=tiles)
    assert np.array_equal(out.cpu().numpy(), O.h(ref32))
    res = (rng.integers(-8, 9, (m, n)) * 0.5).astype(np.float16)
    out2 = torch.empty_like(out)
    ops.linear_ts(af, wf, m, n, k, out=out2, res=_t(res), tiles=tiles)
    assert np.array_equal(out2.cpu().numpy(), O.linear_f16(a, w, res16=res))


@pytest.mark.parametrize("m,inter,k,tiles,out_frag", [(16, 64, 128, 2, False), (34, 3072, 768, 96, True), (48, 11008, 4096, 230, True),
                                                      (64, 1024, 512, 32, False), (128, 2048, 1024, 64, True), (100, 5504, 256, 172, True),
                                                      (128, 11008, 512, 230, True), (96, 3072, 768, 64, False),
                                                      # 4 gate+up units per workgroup (8 MFMA column tiles): the 13B plans
                                                      (64, 13824, 512, 216, True), (128, 13824, 256, 216, True), (17, 1024, 256, 16, False),
                                                      (128, 11008, 512, 172, True), (100, 2048, 128, 32, False),
                                                      # 65 rows (6-tile build) and 129 rows (the extra-row build: 3 and 4 gate+up units per workgroup)
                                                      (65, 11008, 512, 230, True), (65, 1024, 256, 16, False), (129, 3584, 512, 75, True),
                                                      (129, 7168, 256, 112, True), (129, 28672, 128, 598, True), (129, 2048, 256, 43, False)])
def test_linear_ts_swiglu_epilogue(m, inter, k, tiles, out_frag):
    ops = _ops()
    rng = np.random.default_rng(inter + m)
    a, w = _exact_operands(rng, m, 2 * inter, k)
    af, wf = ops.repack_rows(_t(a)), ops.repack_weight(_t(w))
    ref = O.linear_f16(a, w, silu=True)
    out = torch.zeros(ops.frag_shape(m, inter) if out_frag else (m, inter), dtype=torch.float16, device=DEV)
    ops.linear_ts(af, wf, m, inter, k, out=out, silu=True, out_frag=out_frag, tiles=tiles)
    got = out.cpu().numpy()
    if out_frag:
        got = O.unfrag_rows(got, m, inter)
    # silu goes through expf: one fp32 ulp of exp can move the fp16 rounding of silu(g) for ~1e-4 of the elements
    diff = np.abs(got.astype(np.float32) - ref.astype(np.float32))
    assert (diff > 0).mean() < 2e-3 and diff.max() <= np.abs(ref.astype(np.float32)).max() * 2 ** -9


@pytest.mark.parametrize("m,inter,k,tiles,splits,out_frag", [(129, 3584, 8192, 75, 3, True), (128, 1024, 1024, 43, 2, True),
                                                             (34, 3072, 768, 96, 2, False), (144, 7168, 2048, 150, 4, True)])
def test_split_k_gate_up_with_swiglu_from_slabs(m, inter, k, tiles, splits, out_frag):
    """gate|up run as a plain [2 inter] x k projection with K-splits + sq_silu_mul_slabs_f16 == the fused SwiGLU epilogue
    (exact operands: bit-equal up to the expf ulp of silu)."""
    ops = _ops()
    rng = np.random.default_rng(inter + m + splits)
    a, w = _exact_operands(rng, m, 2 * inter, k)
    af, wf = ops.repack_rows(_t(a)), ops.repack_weight(_t(w))
    ref = O.linear_f16(a, w, silu=True)
    slab = torch.empty(splits * m * 2 * inter, dtype=torch.float32, device=DEV)
    ops.linear_ts(af, wf, m, 2 * inter, k, tiles=tiles, splits=splits, slab=slab)
    out = torch.zeros(ops.frag_shape(m, inter) if out_frag else (m, inter), dtype=torch.float16, device=DEV)
    ops.silu_mul_slabs(slab, splits, out, m, inter, out_frag=out_frag)
    got = out.cpu().numpy()
    if out_frag:
        got = O.unfrag_rows(got, m, inter)
    diff = np.abs(got.astype(np.float32) - ref.astype(np.float32))
    assert (diff > 0).mean() < 2e-3 and diff.max() <= np.abs(ref.astype(np.float32)).max() * 2 ** -9
    fused = torch.zeros_like(out)
    ops.linear_ts(af, wf, m, inter, k, out=fused, silu=True, out_frag=out_frag, tiles=min(inter // 16, 256))
    assert torch.equal(fused, out)          # same expf, same roundings: the two forms agree bit for bit on exact operands


def test_linear_ts_random_operands_within_accumulation_tolerance():
    """Gaussian operands at the 7B o_proj shape: fp32 accumulation order is the only freedom (split-K, 4-wave
    partials), so results stay within a few fp32 ulps of a float64 reference before the fp16 rounding."""
    ops = _ops()
    rng = np.random.default_rng(5)
    m, n, k = 48, 4096, 4096
    a = rng.standard_normal((m, k)).astype(np.float16)
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float16)
    af, wf = ops.repack_rows(_t(a)), ops.repack_weight(_t(w))
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    out = torch.empty((m, n), dtype=torch.float16, device=DEV)
    ops.linear_ts(af, wf, m, n, k, out=out, tiles=256)
    got = out.cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() <= 2e-3 + 1.5e-3 * np.abs(ref).max()
    slab = torch.empty(3 * m * n, dtype=torch.float32, device=DEV)
    ops.linear_ts(af, wf, m, n, k, tiles=64, splits=3, slab=slab)
    s = slab.reshape(3, m, n).cpu().numpy().astype(np.float64).sum(0)
    assert np.abs(s - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-4


@pytest.mark.parametrize("rows,hidden,splits,frag", [(1, 768, 2, False), (34, 768, 3, True), (48, 4096, 4, True), (128, 4096, 8, False)])
def test_add_rmsnorm_slabs_matches_oracle(rows, hidden, splits, frag):
    ops = _ops()
    rng = np.random.default_rng(rows + hidden)
    slabs = (rng.standard_normal((splits, rows, hidden)) * 0.7).astype(np.float32)
    res = rng.standard_normal((rows, hidden)).astype(np.float16)
    wt = (1 + 0.1 * rng.standard_normal(hidden)).astype(np.float16)
    exp_sum, exp_norm = O.add_rmsnorm_slabs(slabs, res, wt, 1e-6)
    sum_out = torch.empty((rows, hidden), dtype=torch.float16, device=DEV)
    out = torch.zeros(ops.frag_shape(rows, hidden) if frag else (rows, hidden), dtype=torch.float16, device=DEV)
    ops.add_rmsnorm_slabs(_t(slabs).reshape(-1), splits, _t(res), sum_out, _t(wt), out, 1e-6, out_frag=frag)
    assert np.array_equal(sum_out.cpu().numpy(), exp_sum)
    got = out.cpu().numpy()
    got = O.unfrag_rows(got, rows, hidden) if frag else got
    # the row statistic is an fp32 sum in another order: allow last-bit differences on a few elements
    d = np.abs(got.astype(np.float32) - exp_norm.astype(np.float32))
    assert (d > 0).mean() < 5e-3 and d.max() <= np.abs(exp_norm.astype(np.float32)).max() * 2 ** -9
    # add-only form
    s2 = torch.empty_like(sum_out)
    ops.add_rmsnorm_slabs(_t(slabs).reshape(-1), splits, _t(res), s2, None, None, 1e-6)
    assert np.array_equal(s2.cpu().numpy(), exp_sum)


def test_fragment_major_producers_equal_their_row_major_forms():
    ops = _ops()
    rng = np.random.default_rng(9)
    rows, hidden, inter = 37, 768, 3072
    x = _t(rng.standard_normal((rows, hidden)).astype(np.float16))
    r = _t(rng.standard_normal((rows, hidden)).astype(np.float16))
    wt = _t((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float16))
    rm = torch.empty_like(x); fr = torch.zeros(ops.frag_shape(rows, hidden), dtype=torch.float16, device=DEV)
    ops.rmsnorm(x, wt, rm, 1e-6); ops.rmsnorm_frag(x, wt, fr, 1e-6)
    assert np.array_equal(O.unfrag_rows(fr.cpu().numpy(), rows, hidden), rm.cpu().numpy())
    s1, s2 = torch.empty_like(x), torch.empty_like(x)
    ops.add_rmsnorm(x, r, s1, wt, rm, 1e-6); ops.add_rmsnorm_frag(x, r, s2, wt, fr, 1e-6)
    assert torch.equal(s1, s2) and np.array_equal(O.unfrag_rows(fr.cpu().numpy(), rows, hidden), rm.cpu().numpy())
    gu = _t(rng.standard_normal((rows, 2 * inter)).astype(np.float16))
    a1 = torch.empty((rows, inter), dtype=torch.float16, device=DEV)
    a2 = torch.zeros(ops.frag_shape(rows, inter), dtype=torch.float16, device=DEV)
    ops.silu_mul(gu, a1); ops.silu_mul_frag(gu, a2, rows, inter)
    assert np.array_equal(O.unfrag_rows(a2.cpu().numpy(), rows, inter), a1.cpu().numpy())


@pytest.mark.parametrize("h,hkv,d,q_len", [(12, 12, 64, 34), (32, 32, 128, 128), (8, 1, 128, 65)])
def test_attention_fragment_output_equals_row_major(h, hkv, d, q_len):
    ops = _ops()
    torch.manual_seed(h + q_len)
    m = 384
    gt = 100
    from sequoia_amd.growmap import GrowMap
    g = GrowMap.load("A100-CNN-68m-7b-stochastic" if q_len <= 128 else "64x2-tree")
    bm = g.device_tensors(DEV)["bitmask"]
    q = torch.randn(h, q_len, d, device=DEV).half()
    kc, vc = torch.randn(hkv, m, d, device=DEV).half(), torch.randn(hkv, m, d, device=DEV).half()
    o1 = torch.empty((q_len, h * d), dtype=torch.float16, device=DEV)
    o2 = torch.zeros(ops.frag_shape(q_len, h * d), dtype=torch.float16, device=DEV)
    kw = dict(q_slot0=gt - 1, gt=gt, n_tree=g.size, bitmask=bm)
    ops.tree_attention(q, kc, vc, o1, gt - 1 + q_len, d ** -0.5, **kw)
    ops.tree_attention(q, kc, vc, o2, gt - 1 + q_len, d ** -0.5, out_frag=True, **kw)
    assert np.array_equal(O.unfrag_rows(o2.cpu().numpy(), q_len, h * d), o1.cpu().numpy())


def test_ts_forward_matches_general_forward():
    """A whole decoder forward on the tall-skinny path vs the general (torch GEMM) path of the same model:
    logits within the stochastic-parity logit tolerance (4e-2), greedy argmax identical."""
    from sequoia_amd.Engine import ts_linear
    from sequoia_amd.Engine.Engine import GraphInferenceEngine
    from sequoia_amd.Engine.Llama_modules import TreeContext
    from sequoia_amd.growmap import GrowMap
    cfg = dict(vocab_size=2048, hidden_size=256, intermediate_size=704, num_hidden_layers=3, num_attention_heads=4,
               num_key_value_heads=2, max_position_embeddings=2048)
    from sequoia_amd.Engine.Llama_model import LlamaDims, LlamaWeights
    weights = LlamaWeights.random(LlamaDims.from_any(cfg), torch.float16, DEV, 3, logit_gain=8.0)
    eng = GraphInferenceEngine(max_length=256, model_name_or_path={"weights": weights}, dtype=torch.float16, device=DEV)
    model = eng.engine.model
    assert model.ts is not None
    g = GrowMap.load("A100-CNN-68m-7b-stochastic")
    bm = g.device_tensors(DEV)["bitmask"]
    torch.manual_seed(0)
    n = 50
    ids = torch.randint(3, 2048, (1, n), device=DEV)
    ar = torch.arange(n, device=DEV)
    tree = TreeContext(q_slot0=0, gt=n - g.size + 1 if n > g.size else 1, n_tree=min(g.size, n), bitmask=bm, kv_len=n,
                       contiguous_slots=True)
    out_ts = eng.inference(input_ids=ids, storage_ids=ar, position_ids=ar[None], attn_mask=None, tree=tree)
    plan = model.ts.plan(n)
    eng.clear_kv()
    saved, model.ts = model.ts, None
    try:
        out_ref = eng.inference(input_ids=ids, storage_ids=ar, position_ids=ar[None], attn_mask=None, tree=tree)
    finally:
        model.ts = saved
    assert any(v is not None for v in plan.values()), plan      # the tall-skinny kernel was actually in the path
    d = (out_ts.float() - out_ref.float()).abs().max().item()
    assert d <= 4e-2, d
    assert torch.equal(out_ts.argmax(-1), out_ref.argmax(-1))


def test_autotune_returns_a_usable_plan_and_never_runs_inside_a_capture():
    from sequoia_amd.Engine import ts_linear
    from sequoia_amd.Engine.Llama_model import LlamaDims, LlamaWeights
    cfg = dict(vocab_size=2048, hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, max_position_embeddings=2048)
    W = LlamaWeights.random(LlamaDims.from_any(cfg), torch.float16, DEV, 1)
    ts = ts_linear.TsLinearSet(W, W.dims)
    for name in ts.NAMES:
        rec = ts.autotune(name, 20)
        n_out, k, silu = ts.shapes[name]
        cands = [list(c) for c in ts_linear.candidates(n_out, k, silu, 20, name in ts_linear.SPLITTABLE)]
        assert rec == "torch" or rec in cands
        key = ts_linear.plan_key(n_out, k, silu, 2)
        assert ts.tuned[key]["torch_us"] > 0 and len(ts.tuned[key]["ts_us"]) == len(cands)
    # a plan requested for the first time during capture falls back to torch for unknown shapes and is not cached
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            p = ts.plan(33)
    assert all(v is None for v in p.values()) and 33 not in ts._plans


def test_autotune_never_frees_weight_images_that_existed_before_it(monkeypatch):
    """The fragment-major weight images of a projection are shared by all its launch plans and referenced by every hipGraph
    captured on them.  An autotune run for a NEW row count that ends on "torch" must therefore only drop the images it
    made itself: round 4 hit the use-after-free (a 129-row prefill autotuned after the 2-row verify graph had been
    captured; PyTorch's GEMM won; all images of that projection were freed; the graph replayed on freed memory)."""
    from sequoia_amd.Engine import ts_linear
    from sequoia_amd.Engine.Llama_model import LlamaDims, LlamaWeights
    cfg = dict(vocab_size=2048, hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, max_position_embeddings=2048)
    W = LlamaWeights.random(LlamaDims.from_any(cfg), torch.float16, DEV, 1)
    ts = ts_linear.TsLinearSet(W, W.dims)
    before = {}
    for name in ("qkv", "gate_up"):
        for li in range(2):
            before[(name, li)] = ts.frag(name, li).data_ptr()          # images in use by some plan / captured graph
    monkeypatch.setattr(ts_linear, "candidates", lambda *a, **k: [])    # no kernel candidate: PyTorch's GEMM "wins"
    for name in ts.NAMES:
        assert ts.autotune(name, 129) == "torch"
    for key, ptr in before.items():
        assert key in ts._frag and ts._frag[key].data_ptr() == ptr, f"{key}: image freed or moved by an autotune run"
    assert ("o", 0) not in ts._frag and ("lm_head", 0) not in ts._frag   # nothing new is kept for a "torch" decision


def test_plan_candidates_respect_kernel_limits():
    from sequoia_amd.Engine.ts_linear import candidates
    for n_out, k, silu, m in [(12288, 4096, False, 128), (11008, 4096, True, 48), (11008, 4096, True, 128), (768, 3072, False, 34),
                              (32000, 768, False, 1), (15360, 5120, False, 64), (13824, 5120, True, 64), (28672, 8192, True, 129)]:
        for tiles, splits in candidates(n_out, k, silu, m, allow_split=True):
            # a SwiGLU layer with K-splits runs as a plain [2 n_out] x k projection (+ sq_silu_mul_slabs_f16)
            plain = not silu or splits > 1
            units = (2 * n_out if silu and splits > 1 else n_out) // 16
            per = (units + tiles - 1) // tiles
            wide = 8 if (not silu and m <= 128 and n_out >= 8192) else (6 if m > 64 else 4)
            assert per <= (wide if plain else (4 if m <= 129 else 3))      # (129 rows: the 8-tile + extra-row build)
            assert k // 32 >= splits * 8


def test_tensor_parallel_hooks_run_on_the_tall_skinny_path():
    """The TP engine attaches its all-reduce / vocab-gather hooks after the model is built; a tree forward with hooks stays
    on the tall-skinny projections and applies the reduction between o_proj / down_proj and the residual add -- with an
    identity hook the logits equal the hook-free forward bit for bit."""
    from sequoia_amd.Engine.Engine import GraphInferenceEngine
    from sequoia_amd.Engine.Llama_model import LlamaDims, LlamaWeights
    from sequoia_amd.Engine.Llama_modules import TreeContext
    cfg = dict(vocab_size=2048, hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, max_position_embeddings=2048)
    W = LlamaWeights.random(LlamaDims.from_any(cfg), torch.float16, DEV, 2)
    eng = GraphInferenceEngine(max_length=128, model_name_or_path={"weights": W}, dtype=torch.float16, device=DEV)
    model = eng.engine.model
    assert model.ts is not None
    ids = torch.randint(3, 2048, (1, 20), device=DEV)
    ar = torch.arange(20, device=DEV)
    bm = torch.ones((1, 1), dtype=torch.int64, device=DEV)
    ctx = TreeContext(q_slot0=0, gt=20, n_tree=1, bitmask=bm, kv_len=20, contiguous_slots=True)
    plain = eng.inference(input_ids=ids, storage_ids=ar, position_ids=ar[None], attn_mask=None, tree=ctx).clone()
    calls, gathers = [], []
    model.reduce_fn = lambda t: (calls.append(tuple(t.shape)), t)[1]
    model.gather_logits_fn = lambda t: (gathers.append(1), t)[1]
    eng.clear_kv()
    hooked = eng.inference(input_ids=ids, storage_ids=ar, position_ids=ar[None], attn_mask=None, tree=ctx)
    assert calls == [(20, 256)] * 4 and len(gathers) == 1     # o_proj and down_proj of both layers, then the logits
    assert torch.equal(plain, hooked)


@pytest.mark.parametrize("rows,hidden,vocab,frag", [(1, 768, 1000, True), (34, 768, 32000, False), (128, 4096, 32000, True), (48, 5120, 500, True)])
def test_embed_rmsnorm_equals_gather_then_rmsnorm(rows, hidden, vocab, frag):
    ops = _ops()
    torch.manual_seed(rows + hidden)
    embed = torch.randn(vocab, hidden, device=DEV).half()
    wt = (1 + 0.1 * torch.randn(hidden, device=DEV)).half()
    ids = torch.randint(0, vocab, (rows,), device=DEV)
    x = torch.empty((rows, hidden), dtype=torch.float16, device=DEV)
    out = torch.zeros(ops.frag_shape(rows, hidden) if frag else (rows, hidden), dtype=torch.float16, device=DEV)
    ops.embed_rmsnorm(ids, embed, wt, x, out, 1e-6, out_frag=frag)
    x_ref = embed[ids]
    assert torch.equal(x, x_ref)
    ref = torch.empty_like(x_ref)
    ops.rmsnorm(x_ref.contiguous(), wt, ref, 1e-6)
    got = out.cpu().numpy()
    got = O.unfrag_rows(got, rows, hidden) if frag else got
    d = np.abs(got.astype(np.float32) - ref.cpu().numpy().astype(np.float32))
    # the row statistic is reduced over another thread count: last-bit differences on a few elements at most
    assert (d > 0).mean() < 5e-3 and d.max() <= np.abs(ref.float().cpu().numpy()).max() * 2 ** -9
