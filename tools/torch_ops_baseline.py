"""Per-kernel "before": the reference's own PyTorch op sequences (re-typed from the cited lines, run
on PyTorch-ROCm on the same MI355X, inside a hipGraph wherever the reference graph-captures them)
next to the HIP kernels that replace them, at config-B shapes (68m -> 7B, 128-node tree, V=32000,
M=384).  Measurement tooling only.

    python tools/torch_ops_baseline.py
"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequoia_amd.growmap import GrowMap  # noqa: E402
from sequoia_amd.ops import get_ops  # noqa: E402

dev = "cuda:0"
ops = get_ops()
g = GrowMap.load("A100-CNN-68m-7b-stochastic")
gd = g.device_tensors(dev)
n, V, M, T = g.size, 32000, 384, 0.6
gt = 160


def graph_time(fn, reps=96, per_graph=16):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(per_graph):
            fn()
    gph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nrep = max(1, reps // per_graph)
    torch.cuda.synchronize(); e0.record()
    for _ in range(nrep):
        gph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (nrep * per_graph)


def wall_time(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e6 / reps


rows = []
torch.manual_seed(0)
# ---- K1 sampler: utils.sampling_without_replacement (utils.py:10-18), graph-captured per level (:138-177)
dl = (torch.randn(n, V, device=dev) * 3).half()
rand = torch.rand(n, V, device=dev).half()
tok = torch.zeros(M, dtype=torch.long, device=dev)
t_ref = t_hip = 0.0
for lv in gd["levels"]:
    idx = lv["row_ids"].long()
    lg, rd, k = dl[idx].clone(), rand[idx].clone(), lv["k"]
    def ref():
        q = torch.softmax(lg / T, dim=-1)
        return (rd.log() / q).topk(k=k).indices.flatten()
    t_ref += graph_time(ref)
    t_hip += graph_time(lambda lv=lv: ops.sample_wor(dl, rand, lv["row_ids"], lv["k"], T, tok, branch=lv["branch"], out_off=lv["out_off"]))
rows.append(("K1 sampler, 5 levels (utils.py:10-18)", t_ref, t_hip))

# ---- K4 target tree attention, one 7B layer (Engine/Llama_modules.py:220-248): slice, matmul, +mask, fp32 softmax, matmul
H, D = 32, 128
kv_len = gt - 1 + n
q = torch.randn(1, H, n, D, device=dev).half()
kc = torch.randn(1, H, M, D, device=dev).half(); vc = torch.randn_like(kc)
mask = torch.empty(n, kv_len, dtype=torch.float16, device=dev)
ops.tree_mask_dense(mask, gt - 1, gt, n, gd["bitmask"])
mask4 = mask[None, None]
def ref_attn():
    ks, vs = kc[..., :kv_len, :], vc[..., :kv_len, :]
    w = torch.matmul(q, ks.transpose(2, 3)) / math.sqrt(D)
    w = w + mask4
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, vs)
    return o.transpose(1, 2).contiguous().reshape(1, n, H * D)
o_hip = torch.empty(n, H * D, dtype=torch.float16, device=dev)
q3 = q[0].contiguous()
t_ref = graph_time(ref_attn)
t_hip = graph_time(lambda: ops.tree_attention(q3, kc[0], vc[0], o_hip, kv_len, D ** -0.5, q_slot0=gt - 1, gt=gt, n_tree=n, bitmask=gd["bitmask"]))
err = (ref_attn()[0].float() - o_hip.float()).abs().max().item()
rows.append((f"K4 target attention, 1 layer 7B q=128 kv=287 (Llama_modules.py:220-248) [max |diff| {err:.1e}]", t_ref, t_hip))
# draft flavour: SDPA over all M slots (Llama_modules.py:127-134)
Hd, Dd, qd = 12, 64, 34
qq = torch.randn(1, Hd, qd, Dd, device=dev).half(); kd = torch.randn(1, Hd, M, Dd, device=dev).half(); vd = torch.randn_like(kd)
md = torch.empty(qd, M, dtype=torch.float16, device=dev); ops.tree_mask_dense(md, gt + 20, gt, n, gd["bitmask"])
od = torch.empty(qd, Hd * Dd, dtype=torch.float16, device=dev)
t_ref = graph_time(lambda: F.scaled_dot_product_attention(qq, kd, vd, attn_mask=md[None, None], dropout_p=0.0, is_causal=False).transpose(1, 2).contiguous())
t_hip = graph_time(lambda: ops.tree_attention(qq[0].contiguous(), kd[0], vd[0], od, gt + 20 + qd, Dd ** -0.5, q_slot0=gt + 20, gt=gt, n_tree=n, bitmask=gd["bitmask"]))
rows.append(("K4 draft attention, 1 layer 68m q=34 (SDPA over M=384, Llama_modules.py:127-134)", t_ref, t_hip))

# ---- RoPE + K3 scatter (offload_engine.py:42-67 + Llama_KV.py:84-85), 7B q=128
cos = torch.randn(2048, D, device=dev).half(); sin = torch.randn(2048, D, device=dev).half()
pos = torch.arange(gt - 1, gt - 1 + n, device=dev)
qs = torch.randn(1, H, n, D, device=dev).half(); ks_ = torch.randn(1, H, n, D, device=dev).half(); vs_ = torch.randn(1, H, n, D, device=dev).half()
def rot(x):
    return torch.cat((-x[..., D // 2:], x[..., :D // 2]), dim=-1)
def ref_rope():
    c, s_ = cos[pos[None]].unsqueeze(1), sin[pos[None]].unsqueeze(1)
    qe, ke = qs * c + rot(qs) * s_, ks_ * c + rot(ks_) * s_
    kc.index_copy_(-2, pos, ke); vc.index_copy_(-2, pos, vs_)
    return qe
qkv = torch.randn(n, 3 * H * D, device=dev).half(); qo = torch.empty(H, n, D, dtype=torch.float16, device=dev)
rows.append(("RoPE + K3 KV scatter, 7B q=128 (offload_engine.py:63-66, Llama_KV.py:84-85)", graph_time(ref_rope),
             graph_time(lambda: ops.rope_kv_write(qkv, qo, kc[0], vc[0], cos, sin, pos, pos, H, H, D))))

# ---- K8 compaction on the 7B cache, 4 accepted nodes (Llama_KV.py:60-68, zeroes the whole tail)
K7 = torch.randn(32, 1, 32, M, 128, device=dev).half(); V7 = torch.randn_like(K7)
acc = [gt + 1, gt + 20, gt + 50, gt + 90]
def ref_compact():
    K7[..., gt:gt + 4, :] = K7[..., acc, :]; V7[..., gt:gt + 4, :] = V7[..., acc, :]
    K7[..., gt + 4:, :] = 0.0; V7[..., gt + 4:, :] = 0.0
sl = torch.tensor(acc, dtype=torch.int32, device=dev)
rows.append(("K8 KV compaction, 7B cache, 4 accepted (Llama_KV.py:60-68)", wall_time(ref_compact),
             graph_time(lambda: ops.kv_compact(K7, V7, sl, None, 4, gt, 0))))

# ---- K5-K7 verification: softmax + host walk with a sync per child (SpecTree.py:136-157,196-222)
tl = (torch.randn(n, V, device=dev) * 3).half()
dlv = (tl.float() + torch.randn(n, V, device=dev) * 2).half()
toks = torch.randint(3, V, (M,), device=dev)
for p_, ch in enumerate(g.successors):      # children drawn from the draft like the real loop
    if ch:
        q_ = torch.softmax(dlv[p_].float() / T, -1)
        toks[torch.tensor(ch, device=dev) + gt - 1] = torch.multinomial(q_, len(ch), replacement=False)
r = torch.rand(M, device=dev).half()
def ref_verify():
    d2 = dlv.clone()
    tp = torch.softmax(tl / T, dim=-1)
    node, path = 0, []
    while True:
        p = tp[node]; drow = d2[node]; ch = g.successors[node]; nxt = -1
        for c in ch:
            token = toks[c + gt - 1]
            qd_ = torch.softmax(drow / T, dim=-1)
            if p[token] > r[c + gt - 1] * qd_[token]:          # device -> host sync per child
                nxt = c; break
            res = (p - qd_).relu_(); p = res / res.sum(dim=-1).unsqueeze(-1)
            drow[token] = torch.finfo(torch.float16).min
        if nxt < 0:
            break
        path.append(nxt); node = nxt
        if toks[nxt + gt - 1] == 0 or toks[nxt + gt - 1] == 2:
            return path
    if not torch.isnan(p).any():
        p.float().multinomial(num_samples=1, replacement=True)
    return path
ws = ops.verify_workspace(n, dev); rr = torch.zeros(64 + n, dtype=torch.int32, device=dev)
d3 = dlv.clone(); tk3 = toks.clone()
def hip_verify():
    ops.verify_stochastic(tl, d3, tk3, r, gd["child_off"], gd["child_ids"], n, gt, T, 4242, ws, rr)
    return rr.cpu()                                            # the one host read of the step
rows.append(("K5-K7 stochastic verification incl. host reads (SpecTree.py:136-157,196-222)", wall_time(ref_verify), wall_time(hip_verify)))

# ---- RMSNorm (Llama_modules.py:282-288) and SiLU*up (:271), 7B q=128
x = torch.randn(n, 4096, device=dev).half(); w = torch.ones(4096, device=dev).half(); xo = torch.empty_like(x)
def ref_norm():
    hf = x.to(torch.float32); var = hf.pow(2).mean(-1, keepdim=True)
    return w * (hf * torch.rsqrt(var + 1e-6)).to(torch.float16)
rows.append(("RMSNorm 128x4096 (Llama_modules.py:282-288)", graph_time(ref_norm), graph_time(lambda: ops.rmsnorm(x, w, xo, 1e-6))))
gu = torch.randn(n, 22016, device=dev).half(); ao = torch.empty(n, 11008, dtype=torch.float16, device=dev)
rows.append(("SiLU(gate)*up 128x11008 (Llama_modules.py:271)", graph_time(lambda: F.silu(gu[:, :11008]) * gu[:, 11008:]),
             graph_time(lambda: ops.silu_mul(gu, ao))))

print("| op sequence (reference lines) | PyTorch-ROCm ops, us | HIP kernel, us | x |")
print("|---|---:|---:|---:|")
for name, a, b in rows:
    print(f"| {name} | {a:.1f} | {b:.1f} | {a / b:.1f} |")
