"""Kernel micro-bench at config-B shapes (no model weights): HIP-event timing on the launch
stream, algorithmic GB/s.  python tools/kbench.py [attn|samp|verify|glue|all]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequoia_amd.growmap import GrowMap  # noqa: E402
from sequoia_amd.ops import get_ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
ops = get_ops()
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timeit(fn, reps=200, warm=10, per_graph=32):
    """Average GPU time per call: the calls are captured into a hipGraph (per_graph launches) and
    the graph is replayed, so host launch overhead (ctypes + hipLaunch, ~7 us per eager call) does
    not bound the measurement."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(per_graph):
            fn()
    n_rep = max(1, reps // per_graph)
    gph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n_rep):
        gph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n_rep * per_graph)   # us


g = GrowMap.load("A100-CNN-68m-7b-stochastic")
gd = g.device_tensors(dev)
n, V, M = g.size, 32000, 384
out = {}
if what in ("attn", "all"):
    for (name, H, Hkv, D, L, qn, gt) in [("target7b_verify", 32, 32, 128, 32, n, 160),
                                         ("target7b_prefill", 32, 32, 128, 32, 128 + n - 1, 128),
                                         ("draft68m_level", 12, 12, 64, 2, 34, 160), ("draft68m_1tok", 12, 12, 64, 2, 1, 160),
                                         ("target70b_shard", 8, 1, 128, 80, 129, 160), ("target70b_shard_tp2", 32, 4, 128, 80, 129, 160)]:
        q_slot0 = gt - 1 if qn in (n, 129) else (0 if qn > n else gt + 20)
        kv_len = q_slot0 + qn
        q = torch.randn(H, qn, D, device=dev).half()
        kc = torch.randn(L, Hkv, M, D, device=dev).half()
        vc = torch.randn_like(kc)
        o = torch.empty(qn, H * D, dtype=torch.float16, device=dev)
        li = [0]

        def f():
            l = li[0] % L
            li[0] += 1
            ops.tree_attention(q, kc[l], vc[l], o, kv_len, D ** -0.5, q_slot0=q_slot0, gt=gt, n_tree=n,
                               bitmask=gd["bitmask"])
        t = timeit(f, 320)
        byts = 2 * Hkv * kv_len * D * 2 + 2 * H * qn * D * 2
        out["attn_" + name] = dict(us=round(t, 2), GBps=round(byts / t / 1e3, 1), MB=round(byts / 1e6, 2))
        # the RoPE + KV write launch that precedes it
        qkv = torch.randn(qn, (H + 2 * Hkv) * D, device=dev).half()
        cos = torch.randn(2048, D, device=dev).half()
        pos = torch.arange(q_slot0, q_slot0 + qn, device=dev)
        qr = torch.empty(H, qn, D, device=dev).half()

        def f_rope():
            l = li[0] % L
            li[0] += 1
            ops.rope_kv_write(qkv, qr, kc[l], vc[l], cos, cos, pos, pos, H, Hkv, D)

        out["rope_" + name] = dict(us=round(timeit(f_rope, 320), 2))
if what in ("samp", "all"):
    dl = (torch.randn(n, V, device=dev) * 3).half()
    rand = torch.rand(n, V, device=dev).half()
    tok = torch.zeros(M, dtype=torch.long, device=dev)
    for i, lv in enumerate(gd["levels"]):
        def f(lv=lv):
            ops.sample_wor(dl, rand, lv["row_ids"], lv["k"], 0.6, tok, branch=lv["branch"], out_off=lv["out_off"])
        t = timeit(f, 100)
        byts = lv["n_rows"] * V * 4
        out[f"samp_level{i}_rows{lv['n_rows']}_k{lv['k']}"] = dict(us=round(t, 2), GBps=round(byts / t / 1e3, 1))

    def f():
        ops.topk(dl, gd["levels"][1]["row_ids"], 8, tok)
    out["topk_rows19_k8"] = dict(us=round(timeit(f, 100), 2))
if what in ("verify", "all"):
    tl = (torch.randn(n, V, device=dev) * 3).half()
    r = torch.rand(M, device=dev).half()
    rand = torch.rand(n, V, device=dev).half()
    ws = ops.verify_workspace(n, dev)
    rr = torch.zeros(64 + n, dtype=torch.int32, device=dev)
    n_int = sum(1 for s in g.successors if s)
    gt0 = 160
    # draft quality scenarios; the tree's tokens are SAMPLED from the draft rows (level by level, like the loop), so the
    # accept / reject pattern is the algorithm's, not that of random token ids
    for name, mix, noise in (("independent_draft", 0.0, 3.0), ("poor_draft", 1.0, 2.0), ("good_draft", 1.0, 0.5), ("identical", 1.0, 0.0)):
        dl0 = (tl.float() * mix + torch.randn(n, V, device=dev) * noise).half()
        toks = torch.randint(3, V, (M,), device=dev)
        for lv in gd["levels"]:
            ops.sample_wor(dl0, rand, lv["row_ids"], lv["k"], 0.6, toks[gt0 + lv["first_child"] - 1:], branch=lv["branch"],
                           out_off=lv["out_off"])
        dl = dl0.clone()

        def f(dl=dl, dl0=dl0, toks=toks):
            dl.copy_(dl0)                      # the verifier writes -65504 into rejected entries
            ops.verify_stochastic(tl, dl, toks, r, gd["child_off"], gd["child_ids"], n, gt0, 0.6, 12345, ws, rr)

        def f_copy(dl=dl, dl0=dl0):
            dl.copy_(dl0)
        t = timeit(f, 50) - timeit(f_copy, 50)
        torch.cuda.synchronize()
        res = rr.cpu()
        out["verify_stochastic_" + name] = dict(us=round(t, 2), GBps=round((n + n_int) * V * 2 / t / 1e3, 1), accepted=int(res[1]))

    def f2():
        ops.verify_greedy(tl, toks, gd["child_off"], gd["child_ids"], n, 160, ws, rr)
    t2 = timeit(f2, 50)
    out["verify_greedy"] = dict(us=round(t2, 2), GBps=round(n * V * 2 / t2 / 1e3, 1))
if what in ("glue", "all"):
    x = torch.randn(128, 4096, device=dev).half(); w = torch.ones(4096, device=dev).half(); o = torch.empty_like(x)
    out["rmsnorm_128x4096"] = dict(us=round(timeit(lambda: ops.rmsnorm(x, w, o, 1e-6)), 2))
    out["add_rmsnorm_128x4096"] = dict(us=round(timeit(lambda: ops.add_rmsnorm(x, x, x, w, o, 1e-6)), 2))
    gu = torch.randn(128, 22016, device=dev).half(); ao = torch.empty(128, 11008, device=dev).half()
    out["silu_mul_128x11008"] = dict(us=round(timeit(lambda: ops.silu_mul(gu, ao)), 2))
    qkv = torch.randn(128, 12288, device=dev).half()
    kc = torch.zeros(32, M, 128, device=dev).half(); vc = torch.zeros_like(kc)
    qo = torch.empty(32, 128, 128, device=dev).half()
    cos = torch.randn(2048, 128, device=dev).half(); pos = torch.arange(128, device=dev)
    out["rope_kv_write_7b_q128"] = dict(us=round(timeit(lambda: ops.rope_kv_write(qkv, qo, kc, vc, cos, cos, pos, pos, 32, 32, 128)), 2))
    kc7 = torch.zeros(32, 1, 32, M, 128, device=dev).half(); vc7 = torch.zeros_like(kc7)
    sl = torch.tensor([161, 170, 200, 250], dtype=torch.int32, device=dev)
    out["kv_compact_7b_4rows"] = dict(us=round(timeit(lambda: ops.kv_compact(kc7, vc7, sl, None, 4, 160, 0)), 2))
    e = torch.empty(4, dtype=torch.int32, device=dev)
    out["store_i32(launch floor)"] = dict(us=round(timeit(lambda: ops.store_i32(e, [1, 2, 3])), 2))
for k, v in out.items():
    print(f"{k:34s} {json.dumps(v)}")
