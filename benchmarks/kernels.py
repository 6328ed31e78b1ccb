"""Per-kernel timings of a bench run: HIP-event durations on the launch stream at the workload's shapes, algorithmic bytes
(SURVEY.md §8d), and the PMC records (HBM traffic, MFMA utilisation) committed under profiles/ that belong to them."""
from __future__ import annotations

import json
import os
import sys

import torch

from sequoia_amd.harness import Loop

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def capture_step_inputs(cfg, loop, device):
    """One real speculation step of the loop's model pair, host-driven, on a fresh prompt (the second step of the prompt:
    the target cache is prefilled, the step is a steady one).  Returns what its samplers and its verifier saw -- the draft
    rows BEFORE the verifier masks rejected tokens, the target rows, tokens, acceptance uniforms, sampler noise, gt -- so
    that `kernels` times those launches on the loop's own data, not on synthetic logits (VERDICT r03 #3a: the synthetic
    pair of round 3 rejected less than the loop does and the line flattered the verifier)."""
    probe = Loop(cfg, loop.draft, loop.target, loop.gm_obj, device, loop.prompts, use_graphs=True, pipelined=False)
    loop.draft.clear_kv(); loop.target.clear_kv()
    probe.run_steps(1)                                  # the prefill-bearing first step
    tree = probe.tree
    if tree is None:                                    # (a prompt that ended in one step: take the next one)
        probe.run_steps(1)
        tree = probe.tree
    tree.construct_grow_map()
    snap = dict(gt=int(tree.ground_truth_len), draft_logits=tree.draft_logits[:tree.tree_size].clone(), tokens=tree.tokens.clone(),
                r=tree.r.clone() if getattr(tree, "r", None) is not None else None,
                rand=tree.rand if getattr(tree, "rand", None) is not None else None)
    tree.verify()
    snap["target_logits"] = tree.target_logits.clone()
    snap["accepted"] = int(tree.last_result[1])
    loop.draft.clear_kv(); loop.target.clear_kv()
    return snap


def kernel_rooflines(cfg, loop, device):
    """Per-kernel average duration at the workload's shapes, HIP events on the launch stream
    (torch's current stream is the one the C ABI launches on), and algorithmic bytes
    (SURVEY.md §8d formulas).  Sampler and verifier run on the inputs of a captured loop step (capture_step_inputs)
    and as the launch sequence the device-driven loop issues (Tree/step_graph.py::body)."""
    from sequoia_amd.ops import get_ops
    ops = get_ops()
    tgt = loop.target.engine
    from sequoia_amd.Tree.Tree import growmap_on_device
    g, gdev = growmap_on_device(loop.grow_map, device)
    n, V, M = g.size, 32000, cfg["M"]
    dims = tgt.model.dims
    H, Hkv, D, L = dims.local_heads, dims.local_kv_heads, dims.head_dim, dims.num_hidden_layers
    import torch.distributed as dist
    if cfg.get("tp") and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # tensor-parallel job: a forward is a collective and only rank 0 is here -- synthetic rows instead of a captured step
        tl = (torch.randn(n, V, device=device) * 3).half()
        snap = dict(gt=160, target_logits=tl, draft_logits=(tl.float() + torch.randn(n, V, device=device) * 2).half(),
                    tokens=torch.randint(3, V, (M,), device=device), r=torch.rand(M, device=device).half(),
                    rand=torch.rand(n, V, device=device).half(), accepted=-1)
    else:
        snap = capture_step_inputs(cfg, loop, device)
    # the prompts run from 128 committed tokens to 256: the attention launch is timed at the middle of that range
    gt = 192 if M >= 384 else snap["gt"]
    kv_len = gt - 1 + n
    res = {}

    def timeit(fn, reps=192, per_graph=32):
        """Average GPU time per call with HIP events on the launch stream.  The calls are captured
        into a hipGraph and the graph is replayed (like the real loop, whose forwards are graph
        replays), so the ~7 us host cost of an eager ctypes launch does not bound the number."""
        s0 = torch.cuda.Stream()
        s0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(s0)
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        # the engines captured their graphs under inference_mode; the generator state tensors that
        # capture_begin updates are therefore inference tensors -> capture under the same mode
        with torch.inference_mode():
            with torch.cuda.graph(gph):
                for _ in range(per_graph):
                    fn()
        n_rep = max(1, reps // per_graph)
        gph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n_rep):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / (n_rep * per_graph)

    # target tree attention, one layer (launched L times per verify)
    q = torch.randn(H, n, D, device=device).half()
    out = torch.empty(n, H * D, dtype=torch.float16, device=device)
    kc, vc = tgt.kv_cache.k_cache, tgt.kv_cache.v_cache
    kc.normal_(); vc.normal_()
    layer = [0]

    def attn():
        l = layer[0] % L
        layer[0] += 1
        ops.tree_attention(q, kc[l, 0], vc[l, 0], out, kv_len, D ** -0.5, q_slot0=gt - 1, gt=gt, n_tree=n,
                           bitmask=gdev["bitmask"])
    t = timeit(attn, 320)
    byts = 2 * Hkv * kv_len * D * 2 + 2 * H * n * D * 2
    res["tree_attention_target"] = dict(seconds=t, bytes=byts, launches_per_step=L,
                                        flops=4 * H * n * kv_len * D, kv_len=kv_len)
    # verifier (nodes + walk) on the captured step
    n_internal = sum(1 for s in g.successors if s)
    sgt = snap["gt"]
    if cfg["mode"] == "stochastic":
        tl, dl, toks0, r = snap["target_logits"], snap["draft_logits"], snap["tokens"], snap["r"]
        toks = toks0.clone()
        ws = ops.verify_workspace(n, device)
        rr = torch.zeros(64 + n, dtype=torch.int32, device=device)
        dl2 = dl.clone()

        # The verifier masks the rejected tokens in the draft rows (-65504 writes, Tree/SpecTree.py:156) and compacts
        # `tokens`: every timed launch starts from a fresh copy of both; the copies are timed alone and subtracted.
        def restore():
            dl2.copy_(dl)
            toks.copy_(toks0)

        def ver():
            restore()
            ops.verify_stochastic(tl, dl2, toks, r, gdev["child_off"], gdev["child_ids"], n, sgt, 0.6, 12345, ws, rr)
        t = timeit(ver, 64, 16) - timeit(restore, 64, 16)
        res["verify_stochastic"] = dict(seconds=t, bytes=(n + n_internal) * V * 2, launches_per_step=1,
                                        inputs=(f"captured loop step (gt {sgt}, {snap['accepted']} tree tokens accepted)"
                                                if snap["accepted"] >= 0 else "synthetic rows (tensor-parallel job)"))
        # samplers of one step as the device-driven loop issues them (Tree/step_graph.py::body): per level the two sampler
        # launches on statistics the preceding forward's row adoption left (sq_logits_stats_f16 with the row copy -- the
        # reference's `draft_logits[...] = logits` slice copy rides on that launch), plus the adoption of the next root row
        rand = snap["rand"]
        tokbuf = torch.zeros(M, dtype=torch.long, device=device)
        stats = torch.zeros(ops.stats_shape(n, V), dtype=torch.float32, device=device)
        dl3 = dl.clone()

        ops.logits_stats(dl, 0.6, stats)                  # valid statistics for every row before the first timed pass

        def samp():
            for lv in gdev["levels"]:
                first, total = lv["first_child"], lv["total"]
                ops.sample_wor(dl3, rand, lv["row_ids"], lv["k"], 0.6, tokbuf, branch=lv["branch"], out_off=lv["out_off"], stats=stats)
                ops.logits_stats(dl[first:first + total], 0.6, stats[first:first + total], copy_dst=dl3[first:first + total])
            ops.logits_stats(dl[0:1], 0.6, stats[0:1], copy_dst=dl3[0:1])
        t = timeit(samp, 64, 16)
        rows = sum(lv["n_rows"] for lv in gdev["levels"])
        res["sample_wor_all_levels"] = dict(seconds=t, bytes=rows * V * 4 + sum(lv["total"] for lv in gdev["levels"]) * 8,
                                            launches_per_step=3 * len(gdev["levels"]) + 1,
                                            inputs="captured loop step; statistics + row-adoption launches included")
    else:
        tl, toks0 = snap["target_logits"], snap["tokens"]
        toks = toks0.clone()
        ws = ops.verify_workspace(n, device)
        rr = torch.zeros(64 + n, dtype=torch.int32, device=device)

        def verg():
            toks.copy_(toks0)
            ops.verify_greedy(tl, toks, gdev["child_off"], gdev["child_ids"], n, sgt, ws, rr)
        t = timeit(verg, 64, 16) - timeit(lambda: toks.copy_(toks0), 64, 16)
        res["verify_greedy"] = dict(seconds=t, bytes=n * V * 2, launches_per_step=1, inputs=f"captured loop step (gt {sgt})")
        dl = snap["draft_logits"]
        tokbuf = torch.zeros(M, dtype=torch.long, device=device)

        def topk():
            for lv in gdev["levels"]:
                ops.topk(dl, lv["row_ids"], lv["k"], tokbuf, branch=lv["branch"], out_off=lv["out_off"])
        t = timeit(topk, 64, 16)
        rows = sum(lv["n_rows"] for lv in gdev["levels"])
        res["topk_all_levels"] = dict(seconds=t, bytes=rows * V * 2, launches_per_step=2 * len(gdev["levels"]))
    # KV compaction of 4 accepted nodes on the target cache
    slots = torch.tensor([gt + 1, gt + 20, gt + 50, gt + 90], dtype=torch.int32, device=device)

    def comp():
        ops.kv_compact(kc, vc, slots, None, 4, gt, 0)
    t = timeit(comp, 192)
    res["kv_compact_target"] = dict(seconds=t, bytes=4 * 2 * L * Hkv * D * 2 * 2, launches_per_step=1)
    kc.zero_(); vc.zero_()
    # tall-skinny projections of the verify forward (q = tree size rows), rotating over the layers' weights so
    # that every launch streams its weights from HBM (32 x 33-180 MB >> the 256 MiB Infinity Cache)
    ts = getattr(tgt.model, "ts", None)
    from sequoia_amd.Engine.ts_linear import MAX_ROWS as TS_MAX_ROWS
    if ts is not None and n <= TS_MAX_ROWS:
        plan = ts.plan(n)
        for name in ("qkv", "o", "gate_up", "down"):
            if plan.get(name) is None:
                continue
            tiles, splits = plan[name]
            n_out, k, silu = ts.shapes[name]
            xf = ops.repack_rows((torch.randn(n, k, device=device) * 0.5).half())
            out = torch.empty(ops.frag_shape(n, n_out) if silu else (n, n_out), dtype=torch.float16, device=device)
            li = [0]

            def proj(name=name, tiles=tiles, splits=splits, n_out=n_out, k=k, xf=xf, out=out, silu=silu):
                w = ts.frag(name, li[0] % L)
                li[0] += 1
                if silu and splits > 1:
                    # split-K SwiGLU plan (tensor-parallel shards, Engine/ts_linear.py::forward_ts): the layer runs as a
                    # plain [2 inter] x k projection into fp32 partials, the activation is a pass over them
                    ops.linear_ts(xf, w, n, 2 * n_out, k, tiles=tiles, splits=splits, slab=ts._slab)
                    ops.silu_mul_slabs(ts._slab, splits, out, n, n_out, out_frag=True)
                    return
                ops.linear_ts(xf, w, n, n_out, k, out=out, silu=silu, out_frag=silu, tiles=tiles, splits=splits,
                              slab=ts._slab if splits > 1 else None)
            t = timeit(proj, 128, 32)
            w_rows = 2 * n_out if silu else n_out          # SwiGLU: gate rows + up rows
            out_bytes = splits * n * (w_rows if silu else n_out) * 4 if splits > 1 else n * n_out * 2
            res[f"linear_ts_{name}"] = dict(seconds=t, bytes=w_rows * k * 2 + n * k * 2 + out_bytes, launches_per_step=L,
                                            flops=2 * n * w_rows * k, plan=[tiles, splits], pmc_key=f"{name}@{(n + 15) // 16}")
    elif n > TS_MAX_ROWS and all(getattr(tgt.model.weights.layers[0], a, None) is not None for a in ("wqkv", "wo", "w_gate_up", "w_down")):
        # more than 144 rows (the reference's 193- / 256- / 512-node growmaps): the verify forward's projections are hipBLASLt
        # GEMMs on the row-major weights -- timed the same way, rotating over the layers' weights
        import torch.nn.functional as F
        W = tgt.model.weights
        for name, attr in (("qkv", "wqkv"), ("o", "wo"), ("gate_up", "w_gate_up"), ("down", "w_down")):
            w0 = getattr(W.layers[0], attr)
            n_rows, k = w0.shape
            x = (torch.randn(n, k, device=device) * 0.5).half()
            li = [0]

            def gemm(attr=attr, x=x):
                y = F.linear(x, getattr(W.layers[li[0] % L], attr))
                li[0] += 1
                return y
            t = timeit(gemm, 64, 32)
            res[f"gemm_{name}"] = dict(seconds=t, bytes=n_rows * k * 2 + n * k * 2 + n * n_rows * 2, launches_per_step=L,
                                       flops=2 * n * n_rows * k, plan="hipBLASLt")
    return res


def in_loop_duration(dom, plan=None, rows=128):
    """Average duration of the dominant projection kernel INSIDE the speculation loop, from the newest committed rocprofv3
    kernel-trace summary of the loop alone (profiles/r*_bench_kernel_stats_loop_only.md, `tools/gpu_r0N.sh loop`): the
    `roofline` object times the kernel on an isolated launch loop with HIP events; this is the same kernel between its real
    neighbours.  Returns (avg_us, file) or (None, None).  Only the SwiGLU projection is unambiguous in
    the summary (the one `ts_linear_kernel<MT, NT, D, true>` instantiation with the row count's MT)."""
    import glob
    import re
    if dom != "linear_ts_gate_up":
        return None, None
    mt = (rows + 15) // 16
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench_kernel_stats_loop_only.md")), reverse=True):
        best = None
        for line in open(path):
            m = re.match(r"\| `void ts_linear_kernel<(\d+), (\d+), (\d+), true>\(TsParams\)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
            if m and int(m.group(1)) == mt and (best is None or float(m.group(5)) > best[0]):
                best = (float(m.group(5)), float(m.group(6)))
        if best is not None:
            return best[1], os.path.basename(path)
    return None, None


def source_sha(*names):
    """sha256[:16] over kernel sources: a PMC record is only valid for the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(REPO, "sequoia_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_lookup(pmc_key, dom):
    """HBM bytes per launch and MFMA utilisation of the dominant kernel from the newest profiles/r*_pmc.json (rocprofv3 PMC
    passes, tools/pmc_r04.sh: FETCH_SIZE / WRITE_SIZE / SQ group in separate passes, gfx950 correction 2 FETCH + WRITE).
    The record must carry the sha of the kernel source it was measured on and that sha must match the tree bench runs
    from: a stale record gives traffic = null and says so.  -> (traffic, mfma_util, file, note)"""
    import glob
    src = "tree_attention.hip" if dom == "tree_attention_target" else "ts_linear.hip"
    want = source_sha(src, "common.h")
    notes = []
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                pm = json.load(f)
            rec = pm["kernels"][pmc_key]
        except (OSError, KeyError, ValueError) as e:
            notes.append(f"{os.path.basename(path)}: no record for {pmc_key} ({type(e).__name__})")
            continue
        have = (pm.get("source_sha") or {}).get(src)
        if have != want:
            notes.append(f"{os.path.basename(path)}: measured on {src} {have}, this tree has {want}")
            continue
        return rec["hbm_bytes_per_launch"], rec["mfma_util"], os.path.relpath(path, REPO), None
    return None, None, None, "no valid PMC record for " + pmc_key + ": " + "; ".join(notes) + " -- re-run tools/pmc_r04.sh"


def pmc_northstar(growmap_levels):
    """HBM-side bytes per step of the kernels BASELINE.json's north_star names -- the samplers (statistics + parts + merge
    launches of every tree level) and the verifier (nodes + walk) -- from the newest profiles/r*_pmc_northstar.json
    (tools/gpu_r05.sh pmc_ns: rocprofv3 PMC passes over tools/kbench.py, FETCH_SIZE / WRITE_SIZE separately, 2 FETCH + WRITE).
    Records are keyed by (kernel, grid): a level of R parent rows launches R x 8 parts of 256 threads.  The record must have
    been measured on this tree's sampler.hip / verify.hip.  -> {kernels-key: dict(traffic, pmc_file) or dict(traffic=None, note)}"""
    import glob
    out = {}
    want = {"sampler.hip": source_sha("sampler.hip", "common.h"), "verify.hip": source_sha("verify.hip", "common.h")}
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_northstar.json")), reverse=True):
        try:
            with open(path) as f:
                pm = json.load(f)
        except (OSError, ValueError):
            continue
        have = pm.get("source_sha") or {}
        K = pm.get("kernels", {})

        def one(sub, grid=None, flavour=None):
            for k, v in K.items():
                if sub in k and (grid is None or v.get("grid") == grid) and (flavour is None or flavour in k) and v.get("hbm_bytes_per_launch") is not None:
                    return v["hbm_bytes_per_launch"]
            return None
        rel = os.path.relpath(path, REPO)
        if have.get("verify.hip") == want["verify.hip"] and "verify_stochastic" not in out:
            a, b = one("verify_nodes_kernel"), one("verify_walk_kernel")
            if a is not None and b is not None:
                out["verify_stochastic"] = dict(traffic=a + b, pmc_file=rel)
        if have.get("sampler.hip") == want["sampler.hip"] and "sample_wor_all_levels" not in out:
            tot, ok = 0, True
            for rows in growmap_levels:
                parts = one("sample_parts_kernel", rows * 8 * 256, "ILi1E")
                stats = one("logits_stats_kernel", rows * 8 * 256)
                merge = one("sample_merge_rank_kernel", rows * 256)
                if None in (parts, stats, merge):
                    ok = False
                    break
                tot += parts + stats + merge
            if ok:
                out["sample_wor_all_levels"] = dict(traffic=tot, pmc_file=rel)
    for k, src in (("verify_stochastic", "verify.hip"), ("sample_wor_all_levels", "sampler.hip")):
        out.setdefault(k, dict(traffic=None, traffic_note=f"no profiles/r*_pmc_northstar.json record measured on this tree's {src}"))
    return out
