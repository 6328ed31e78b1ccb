"""API-level behaviour on the GPU: KV_Cache methods, engine helpers, terminal conditions, hipGraph
capturability of every device entry point of the C ABI."""
import pytest
import torch

from conftest import load_trace
from helpers import build_engines, make_tree

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_kv_cache_api_matches_reference_semantics():
    from sequoia_amd.Engine.Llama_KV import KV_Cache
    from sequoia_amd.Engine.Llama_model import KVConfigView, LlamaDims
    dims = LlamaDims(vocab_size=64, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                     num_key_value_heads=2)
    kv = KV_Cache(KVConfigView(dims), max_length=48, device=DEV)
    L, H, M, D = 3, 2, 48, 64
    g = torch.Generator(device=DEV).manual_seed(0)
    # update_kv_cache: reference layout [1, H_kv, q, D]; kv_offset advances on the last layer only (:87-88)
    sid = torch.tensor([0, 1, 2, 5, 9], device=DEV)
    for l in range(L):
        nk = torch.randn(1, H, 5, D, generator=g, device=DEV).half()
        nv = torch.randn(1, H, 5, D, generator=g, device=DEV).half()
        k_l, v_l = kv.update_kv_cache(nk, nv, l, sid)
        assert torch.equal(k_l[0][:, sid], nk[0]) and torch.equal(v_l[0][:, sid], nv[0])
        assert kv.kv_offset == (5 if l == L - 1 else 0)
        assert kv.get_usable_length(l, 5) == 5
    ref_k, ref_v = kv.k_cache.clone(), kv.v_cache.clone()
    # gather_kv_incremental (Python list of slots, offset) == the reference's tensor expression
    KV_Cache.ZERO_POLICY = "full"
    try:
        kv.gather_kv_incremental([5, 9], 3)
        ref_k[..., 3:5, :] = ref_k[..., [5, 9], :]; ref_k[..., 5:, :] = 0
        ref_v[..., 3:5, :] = ref_v[..., [5, 9], :]; ref_v[..., 5:, :] = 0
        assert torch.equal(kv.k_cache, ref_k) and torch.equal(kv.v_cache, ref_v) and kv.kv_offset == 5
        kv.gather_kv([0, 2, 4])
        ref_k[..., :3, :] = ref_k[..., [0, 2, 4], :].clone(); ref_k[..., 3:, :] = 0
        assert torch.equal(kv.k_cache, ref_k) and kv.kv_offset == 3
    finally:
        KV_Cache.ZERO_POLICY = "none"
    kv.set_kv_len(2); assert kv.kv_offset == 2
    k2 = torch.randn_like(kv.k_cache); v2 = torch.randn_like(kv.v_cache)
    kv.initialize_kv(k2, v2, 7)
    assert torch.equal(kv.k_cache[..., :7, :], k2[..., :7, :]) and kv.kv_offset == 7
    kv.clear()
    assert kv.kv_offset == 0 and not kv.k_cache.any() and not kv.v_cache.any()
    with pytest.raises(ValueError):
        KV_Cache(KVConfigView(dims), batch_size=2, max_length=8, device=DEV)


def test_terminal_conditions_and_max_target_seq():
    z, meta = load_trace("demo4")
    draft, target = build_engines(z, meta, DEV)
    tree = make_tree(z, meta, draft, target, DEV)
    # force EOS: make every proposed token 2 and r = 0 (accept iff p > 0)
    tree.construct_grow_map()
    gt, n = tree.ground_truth_len, tree.tree_size
    tree.tokens[gt:gt + n - 1] = 2
    tree.r.zero_()
    valid, a, _, terminal = tree.verify()
    if a > gt:                                   # a child was accepted -> it is EOS -> terminal, no bonus
        assert terminal and valid.shape[0] == a and int(valid[-1]) == 2
    # max_target_seq: prepare_for_next_iter returns early without touching state (Tree/SpecTree.py:262-263)
    draft.clear_kv(); target.clear_kv()
    tree = make_tree(z, meta, draft, target, DEV)
    tree.max_target_seq = tree.ground_truth_len          # a + 1 > max_target_seq at the first step
    tree.construct_grow_map()
    gt0 = tree.ground_truth_len
    tree.verify()
    assert tree.ground_truth_len == gt0


def test_prefix_plus_tree_must_fit():
    z, meta = load_trace("B_seq128")
    draft, target = build_engines(z, meta, DEV)
    from sequoia_amd.growmap import GrowMap
    from sequoia_amd.Tree.SpecTree import SpecTree
    M = meta["M"]
    g = GrowMap.from_successors(meta["successors"]).to_reference_dict()
    with pytest.raises(ValueError):
        SpecTree(prefix=torch.randint(3, 1000, (M - 10,)), device=DEV, temperature=0.6, top_p=1.0, draft_kv_len=0,
                 target_kv_len=0, draft_model_engine=draft, target_model_engine=target, max_length=M, max_target_seq=M,
                 grow_map=g, attn_mask=None, sequence=None, new_tokens_buffer=None, parents_buffer=None,
                 position_ids=torch.zeros(M, device=DEV).long(), residual_graph=None, sampling_callables=None,
                 sample_gather_indices=None, vocab_size=meta["vocab"])


def test_every_device_entry_point_is_graph_capturable():
    """All launches are stream-ordered, allocation-free and sync-free: one hipGraph captures one call
    of every op; replaying it on fresh inputs gives the eager results bit for bit."""
    from sequoia_amd.growmap import GrowMap
    from sequoia_amd.ops import get_ops
    ops = get_ops()
    g = GrowMap.load("8x8-tree"); gd = g.device_tensors(DEV)
    n, V, M, H, D, L = g.size, 4096, 128, 4, 64, 2
    gen = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=gen, device=DEV).half()
    tl, dl = rnd(n, V) * 3, rnd(n, V) * 3
    rand = (torch.randint(0, 2048, (n, V), generator=gen, device=DEV).float() / 2048).half()
    tokens = torch.randint(3, V, (M,), generator=gen, device=DEV)
    r = (torch.randint(0, 2048, (M,), generator=gen, device=DEV).float() / 2048).half()
    kc, vc = rnd(L, 1, H, M, D), rnd(L, 1, H, M, D)
    qkv = rnd(n, 3 * H * D); q_rot = torch.empty(H, n, D, dtype=torch.float16, device=DEV)
    cos, sin = rnd(256, D), rnd(256, D)
    pos = torch.arange(20, 20 + n, device=DEV); sid = torch.arange(20, 20 + n, device=DEV)
    attn = torch.empty(n, H * D, dtype=torch.float16, device=DEV)
    x = rnd(n, 256); w = torch.ones(256, device=DEV).half(); xo = torch.empty_like(x)
    gu = rnd(n, 512); act = torch.empty(n, 256, dtype=torch.float16, device=DEV)
    mask = torch.empty(n, 64, dtype=torch.float16, device=DEV)
    ws = ops.verify_workspace(n, DEV)
    res_s = torch.zeros(64 + n, dtype=torch.int32, device=DEV); res_g = torch.zeros_like(res_s)
    toks_s, toks_g = tokens.clone(), tokens.clone()
    samp = torch.zeros(8 * 8, dtype=torch.int64, device=DEV); top = torch.zeros(8 * 4, dtype=torch.int64, device=DEV)
    ctx = torch.tensor([19, 20, 20 + n - 1], dtype=torch.int32, device=DEV)
    slots = torch.tensor([21, 25, 30], dtype=torch.int32, device=DEV)
    tl_f = tl.clone()

    def step():
        ops.tree_mask_dense(mask, 10, 20, n, gd["bitmask"])
        ops.rope_kv_write(qkv, q_rot, kc[0, 0], vc[0, 0], cos, sin, pos, sid, H, H, D)
        ops.store_i32(ctx, [19, 20, 20 + n - 1])
        ops.tree_attention(q_rot, kc[0, 0], vc[0, 0], attn, 20 + n - 1, D ** -0.5, q_slot0=19, gt=20, n_tree=n,
                           bitmask=gd["bitmask"], ctx=ctx)
        ops.rmsnorm(x, w, xo, 1e-6); ops.add_rmsnorm(x, xo, xo, w, xo, 1e-6); ops.silu_mul(gu, act)
        ops.sample_wor(dl, rand, gd["levels"][0]["row_ids"], 8, 0.6, samp)
        ops.topk(dl, gd["levels"][1]["row_ids"][:8], 4, top)
        ops.top_p_filter(tl_f, 0.9, 0.6)
        ops.verify_stochastic(tl, dl, toks_s, r, gd["child_off"], gd["child_ids"], n, 20, 0.6, 777, ws, res_s)
        ops.verify_greedy(tl, toks_g, gd["child_off"], gd["child_ids"], n, 20, ws, res_g)
        ops.kv_compact(kc, vc, slots, None, 3, 20, 0)
        ops.kv_scatter(kc[1, 0], vc[1, 0], kc[0, 0][:, :5].contiguous(), vc[0, 0][:, :5].contiguous(), sid[:5])

    outs = (mask, q_rot, attn, xo, act, samp, top, tl_f, toks_s, toks_g, res_s, res_g, kc, vc, dl)
    ins = (kc, vc, dl, tl_f, toks_s, toks_g, xo)
    saved = [t.clone() for t in ins]
    step(); torch.cuda.synchronize()
    want = [t.clone() for t in outs]
    for t, s in zip(ins, saved):
        t.copy_(s)
    gph = torch.cuda.CUDAGraph()
    s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        step()                                   # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s0)
    for t, s in zip(ins, saved):
        t.copy_(s)
    with torch.cuda.graph(gph):
        step()
    for t, s in zip(ins, saved):
        t.copy_(s)
    gph.replay(); torch.cuda.synchronize()
    for got, w_ in zip(outs, want):
        assert torch.equal(got, w_)
