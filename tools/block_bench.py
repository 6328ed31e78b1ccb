"""Microbenchmark of the fused draft attention block (csrc/draft_block.hip) against the four launches it replaces, per row
count, timed as hipGraph replays of `reps` back-to-back calls over the layers of a 68m-dims model (python tools/block_bench.py)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequoia_amd.ops import get_ops

DEV = "cuda:0"
H, D, hidden, M, L = 12, 64, 768, 384, 8
ops = get_ops()
g = torch.Generator(device="cpu").manual_seed(0)
wqkv = [ops.repack_weight((torch.randn(3 * H * D, hidden, generator=g) * 0.04).half().to(DEV)) for _ in range(L)]
wo = [ops.repack_weight((torch.randn(hidden, H * D, generator=g) * 0.04).half().to(DEV)) for _ in range(L)]
k = torch.randn(L, H, M, D, generator=g).half().to(DEV)
v = torch.randn(L, H, M, D, generator=g).half().to(DEV)
cos = torch.randn(512, D, generator=g).half().to(DEV); sin = torch.randn(512, D, generator=g).half().to(DEV)
bm = torch.zeros((128, 2), dtype=torch.int64, device=DEV)
slab = torch.empty(144 * 8 * 2304, dtype=torch.float32, device=DEV)
x_res = torch.randn(144, hidden, generator=g).half().to(DEV)
ln = torch.ones(hidden, dtype=torch.float16, device=DEV)


def timeit(fn, reps=32):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        for i in range(2):
            fn(i % L)
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for i in range(reps):
                fn(i % L)
        gr.replay(); s.synchronize()
        best = 1e9
        for _ in range(5):
            e0.record(s); gr.replay(); e1.record(s); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


out = {}
for q in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 20, 34, 48]:
    q_slot0, gt = 200, 180
    x = torch.randn(q, hidden, generator=g).half().to(DEV)
    a_f = ops.repack_rows(x)
    pos = torch.full((q,), 190, dtype=torch.long, device=DEV)
    sid = (q_slot0 + torch.arange(q, device=DEV)).long()
    qkv_rows = torch.empty((q, 3 * H * D), dtype=torch.float16, device=DEV)
    q_rot = torch.empty((H, q, D), dtype=torch.float16, device=DEV)
    attn = torch.empty(ops.frag_shape(q, H * D), dtype=torch.float16, device=DEV)
    h_out = torch.empty(ops.frag_shape(q, hidden), dtype=torch.float16, device=DEV)
    xr = x_res[:q].clone()
    mtp = (q + 15) // 16
    from sequoia_amd.Engine.ts_linear import shipped_plans, plan_key
    pq = shipped_plans().get(plan_key(3 * H * D, hidden, False, mtp)) or [72, 2]
    po = shipped_plans().get(plan_key(hidden, H * D, False, mtp)) or [24, 3]

    def unfused(li):
        ops.linear_ts(a_f, wqkv[li], q, 3 * H * D, hidden, tiles=pq[0], splits=pq[1], slab=slab)
        ops.rope_kv_write_slabs(slab, pq[1], 3 * H * D, q_rot, k[li], v[li], cos, sin, pos, sid, H, H, D)
        ops.tree_attention(q_rot, k[li], v[li], attn, q_slot0 + q, D ** -0.5, q_slot0=q_slot0, gt=gt, n_tree=128, bitmask=bm,
                           out_frag=True)
        ops.linear_ts(attn, wo[li], q, hidden, H * D, tiles=po[0], splits=po[1], slab=slab)
        ops.add_rmsnorm_slabs(slab, po[1], xr, xr, ln, h_out, 1e-5, out_frag=True)

    def fused(li):
        ops.draft_attn_block(a_f, wqkv[li], wo[li], slab, k[li], v[li], cos, sin, pos, sid, q, H, D, hidden, D ** -0.5, q_slot0, gt,
                             128, bitmask=bm)
        ops.add_rmsnorm_slabs(slab, H, xr, xr, ln, h_out, 1e-5, out_frag=True)

    def fused_only(li):
        ops.draft_attn_block(a_f, wqkv[li], wo[li], slab, k[li], v[li], cos, sin, pos, sid, q, H, D, hidden, D ** -0.5, q_slot0, gt,
                             128, bitmask=bm)

    def kv_only(li):
        ops.draft_attn_block(a_f, wqkv[li], None, None, k[li], v[li], cos, sin, pos, sid, q, H, D, hidden, D ** -0.5, q_slot0, gt,
                             128, bitmask=bm, kv_only=True)

    def norm12(li):
        ops.add_rmsnorm_slabs(slab, H, xr, xr, ln, h_out, 1e-5, out_frag=True)

    def norm3(li):
        ops.add_rmsnorm_slabs(slab, po[1], xr, xr, ln, h_out, 1e-5, out_frag=True)

    out[q] = dict(unfused_5_launches=round(timeit(unfused), 2), block_plus_norm=round(timeit(fused), 2),
                  block=round(timeit(fused_only), 2), block_kv_only=round(timeit(kv_only), 2),
                  norm_12_slabs=round(timeit(norm12), 2), norm_plan_slabs=round(timeit(norm3), 2))
    print(q, out[q], flush=True)
print(json.dumps(out))
