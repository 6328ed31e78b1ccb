#!/usr/bin/env python
"""bench.py — accepted tokens/sec of the Sequoia speculation loop on MI355X.

Workload (BASELINE.json configs[1]): JackFram/llama-68m draft -> Llama-2-7b target
architectures (random-init weights, no checkpoints offline), growmap
A100-CNN-68m-7b-stochastic (128-node tree), T = 0.6, top-p = 1.0, M = 384, prompts = first 128
tokens of the reference's c4_small rows, generation to 256 tokens — the loop of
tests/testbed.py:45-95 (`simulation_fast`).  A "step" is one speculation step:
construct_grow_map() + verify().

    python bench.py --gpus N --steps K --warmup W

N > 1 = independent replicas (one process per GPU, prompts sharded rank::world, no data-path
collective: the loop is a batch-1 latency loop); `--config E` shards the 70B
target tensor-parallel over the N ranks instead (RCCL all-reduce over xGMI).  When N > 1 and the
process was not started by torchrun (no WORLD_SIZE in the environment) bench.py launches the N
ranks itself through torch.distributed.run on 127.0.0.1.  Prints one JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from sequoia_amd.harness import MODELS, Loop, build, load_prompts  # noqa: E402


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
    return res


def step_weight_bytes(loop, gm):
    """Weight bytes one speculation step streams from HBM: every projection of the target once (the verify forward) and of
    the draft once per tree level plus once for the next-root forward (the end-to-end step is HBM-bound on
    these bytes).  Tensor-parallel shards count their own rank's bytes."""
    def model_bytes(engine):
        m = engine.engine.model
        W = m.weights
        per_layer = sum(w.numel() * 2 for lw in W.layers[:1] for w in (lw.wqkv, lw.wo, lw.w_gate_up, lw.w_down) if w is not None)
        if per_layer == 0 and getattr(m, "ts", None) is not None:       # exclusive mode: only the fragment-major images exist
            per_layer = m.ts.layer_weight_bytes() // len(W.layers)
        return per_layer * len(W.layers) + W.lm_head.numel() * 2
    n_draft_forwards = (len(gm.levels) if hasattr(gm, "levels") else 0) + 1
    t, d = model_bytes(loop.target), model_bytes(loop.draft)
    return dict(target=t, draft=d, draft_forwards=n_draft_forwards, total=t + d * n_draft_forwards)


def tp_bytes_per_rank(wb):
    """Weight bytes one speculation step streams PER RANK at TP = 1 / 2 / 4 / 8 for the two draft placements (the choice
    harness.build leaves to SEQUOIA_TP_DRAFT: VERDICT r04 weak #10 -- decide it from these bytes and the first real all-reduce
    latencies, not from ranks time-slicing one GPU).  Target: column- / row-parallel shards + the vocabulary-parallel lm_head
    = target / W.  Draft replicated: the whole draft x (tree levels + the next-root forward) on every rank; sharded: / W, at
    the price of 2 all-reduces per draft layer and forward."""
    out = {}
    for w in (1, 2, 4, 8):
        t = wb["target"] / w
        rep, shd = wb["draft"] * wb["draft_forwards"], wb["draft"] * wb["draft_forwards"] / w
        out[f"tp{w}"] = dict(target_GB=round(t / 1e9, 2), draft_replicated_GB=round(rep / 1e9, 2), draft_sharded_GB=round(shd / 1e9, 2),
                             step_GB_replicated_draft=round((t + rep) / 1e9, 2), step_GB_sharded_draft=round((t + shd) / 1e9, 2),
                             ms_at_6p3TBps_replicated=round((t + rep) / 6.3e12 * 1e3, 2), ms_at_6p3TBps_sharded=round((t + shd) / 6.3e12 * 1e3, 2))
    return out


def run_other_config(name, args, device, prompts, engines=None, steps=20, warmup=5):
    """Configs C / D / E after the headline (VERDICT r03 #3c, r04 #5): the same device-driven loop, `steps` timed steps
    beginning with a fresh prompt like the headline window, their own roofline object (dominant kernel by time per step,
    HIP-event timing on the launch stream).  E = the 70B target on ONE GPU (TP = 1: 138 GB of fragment-major weights, the
    only hardware anchor the tensor-parallel configuration has while no multi-GPU node is available to the driver)."""
    cfg = dict(MODELS[name])
    t0 = time.perf_counter()
    if engines is None:
        draft, target, gm = build(cfg, device, args.pair)
        torch.cuda.synchronize()
    else:
        from sequoia_amd.growmap import GrowMap
        draft, target = engines
        gm = GrowMap.load(cfg["growmap"])
        draft.clear_kv(); target.clear_kv()
    torch.manual_seed(17)
    loop = Loop(cfg, draft, target, gm, device, prompts, use_graphs=not args.no_graphs,
                pipelined=not args.sync_loop and not args.no_graphs)
    weight_build_s = time.perf_counter() - t0 if engines is None else None
    loop.run_steps(warmup)
    if not args.steady_window:
        loop.start_fresh_prompt()
    torch.cuda.synchronize()
    p0, ps0, pt0 = loop.prefill_steps, loop.prefill_seconds, loop.prefill_tokens
    secs, new_tok, steps_done = loop.run_steps(steps)
    torch.cuda.synchronize()
    pf_n, pf_s, pf_t = loop.prefill_steps - p0, loop.prefill_seconds - ps0, loop.prefill_tokens - pt0
    kr = kernel_rooflines(cfg, loop, device)
    per_step = {k: v["seconds"] * v["launches_per_step"] if (k == "tree_attention_target" or k.startswith("linear_ts_")) else v["seconds"]
                for k, v in kr.items()}
    dom = max(per_step, key=per_step.get)
    d = kr[dom]
    wb = step_weight_bytes(loop, gm)
    out = dict(workload=f"config {name}: {cfg['draft']} -> {cfg['target']} architectures, growmap {cfg['growmap']} ({gm.size}-node tree)",
               value=new_tok / secs, unit="tokens/s", ms_per_step=secs / steps_done * 1e3, steps=steps_done, warmup=warmup,
               mean_accepted_len=new_tok / steps_done, prefill_steps_in_timed_region=pf_n,
               prefill_step_ms_in_timed_region=(pf_s / pf_n * 1e3) if pf_n else None,
               value_steady=(new_tok - pf_t) / max(secs - pf_s, 1e-9), steady_ms_per_step=(secs - pf_s) / max(steps_done - pf_n, 1) * 1e3,
               weight_build_s=None if weight_build_s is None else round(weight_build_s, 1),
               roofline=dict(bound="hbm", kernel=dom, achieved=d["bytes"] / d["seconds"] / 1e9, peak=8000.0, unit="GB/s",
                             frac=d["bytes"] / d["seconds"] / 1e9 / 8000.0, avg_launch_us=d["seconds"] * 1e6,
                             algorithmic_bytes_per_launch=d["bytes"], time_per_step_us=per_step[dom] * 1e6, plan=d.get("plan"),
                             traffic=None),
               step_roofline=dict(weight_bytes=wb["total"], frac=wb["total"] / ((secs - pf_s) / max(steps_done - pf_n, 1)) / 8e12,
                                  note="weight bytes of one step over steady_ms_per_step"),
               **(dict(parallelism="tp1 (the 70B target on one GPU, fragment-major weights only)", tp_bytes_per_rank=tp_bytes_per_rank(wb))
                  if cfg.get("tp") else {}),
               kernels={k: dict(avg_us=round(v["seconds"] * 1e6, 2), per_step_us=round(per_step[k] * 1e6, 1),
                                frac=round(v["bytes"] / v["seconds"] / 8e12, 4), plan=v.get("plan")) for k, v in kr.items()},
               seconds_total=round(time.perf_counter() - t0, 1))
    tuned_names = {"D": "MI355X-synthetic-1.3b-13b-stochastic"}
    if name in tuned_names:
        # the growmap sequoia_amd.growmap_tuning searched for this GPU and this (synthetic) model pair, like `mi355x_growmap` of
        # the headline: the config's `value` stays on the growmap BASELINE.json names
        try:
            from sequoia_amd.growmap import GrowMap
            gm2 = GrowMap.load(tuned_names[name])
            draft.clear_kv(); target.clear_kv()
            torch.manual_seed(17)
            loop2 = Loop(cfg, draft, target, gm2, device, prompts, use_graphs=not args.no_graphs,
                         pipelined=not args.sync_loop and not args.no_graphs)
            loop2.run_steps(warmup)
            torch.cuda.synchronize()
            s2, t2, k2 = loop2.run_steps(steps)
            torch.cuda.synchronize()
            out["mi355x_growmap"] = dict(growmap=tuned_names[name], nodes=gm2.size, levels=[lv.total for lv in gm2.levels],
                                         value=t2 / s2, unit="tokens/s", ms_per_step=s2 / k2 * 1e3, mean_accepted_len=t2 / k2, steps=k2)
            del loop2
        except Exception as e:
            out["mi355x_growmap"] = dict(error=f"{type(e).__name__}: {e}")
    if name == "D":
        pk = None
        if dom.startswith("linear_ts_") and d.get("plan"):
            pk = f"D:{dom[len('linear_ts_'):]}@{(gm.size + 15) // 16}:{d['plan'][0]}x{d['plan'][1]}"
            traffic, mfma_util, pmc_file, note = pmc_lookup(pk, dom)
            out["roofline"].update(traffic=traffic, mfma_util=mfma_util, pmc_key=pk, pmc_file=pmc_file)
            if note:
                out["roofline"]["traffic_note"] = note
    del loop
    return out, (draft, target)


T_START = time.perf_counter()


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


def cpu_baseline(cfg, n_steps=3, pair="calibrated", engines=None, numpy_ops=False):
    """The CPU path timed on this box's host cores: the same host loop with the reference's PyTorch op sequences
    restated for CPU tensors (oracle/ops_torch_cpu.py; verification on the numpy oracle) and PyTorch CPU GEMMs, fp16
    like the reference, on a bounded sample: n_steps
    speculation steps of the first prompt.  The first step carries the 255-token target prefill (the reference's
    timer includes it, tests/testbed.py:78-89): it is reported separately, `value` / `steps_per_s` are the steady
    steps after it.  profiles/r02_cpu_reference_vs_port.json holds a run of the IMPORTED reference
    (oracle/ref_cpu_baseline.py) beside this port on the same weights, prompt and noise."""
    from oracle.ops_adapter import OracleOps
    from oracle.ops_torch_cpu import TorchCpuOps
    from sequoia_amd import ops as ops_mod
    prev = ops_mod._OPS
    ops_mod.set_ops_for_testing(OracleOps() if numpy_ops else TorchCpuOps())     # numpy_ops: the checking oracle (slow)
    # fp16 CPU GEMMs of <= 255 rows do not scale past a few dozen threads (128 threads: 13 s / step, 8 threads: 3.3 s on one
    # box, the other way round on another): after the prefill step ONE steady step is timed at each of 8 / 16 / 32
    # threads and 64 (SEQUOIA_CPU_THREADS=a,b,c overrides) and the fastest is the baseline -- the honest best of this host
    prev_threads = torch.get_num_threads()
    avail = os.cpu_count() or prev_threads
    sweep = [int(x) for x in os.environ.get("SEQUOIA_CPU_THREADS", "16,32,64").split(",") if x.strip()]
    sweep = sorted({max(1, min(t, avail)) for t in sweep}) or [prev_threads]
    torch.set_num_threads(sweep[len(sweep) // 2])
    try:
        t0 = time.perf_counter()
        draft, target, gm = engines if engines is not None else build(cfg, "cpu", pair)
        build_s = time.perf_counter() - t0
        from sequoia_amd.Tree.GreedyTree import GreedyTree
        from sequoia_amd.Tree.SpecTree import SpecTree
        from sequoia_amd.Tree._native_tree import COMMIT_ORDER
        M = cfg["M"]
        cls = SpecTree if cfg["mode"] == "stochastic" else GreedyTree
        p = torch.tensor(load_prompts()[0][:128], dtype=torch.long)
        torch.manual_seed(17)
        tree = cls(prefix=p, device="cpu", temperature=0.6, top_p=1.0, draft_kv_len=0, target_kv_len=0,
                   draft_model_engine=draft, target_model_engine=target, max_length=M, max_target_seq=M,
                   grow_map=gm.to_reference_dict(), attn_mask=None, sequence=None, new_tokens_buffer=None,
                   parents_buffer=None, position_ids=torch.zeros(M, dtype=torch.long), residual_graph=None,
                   sampling_callables=None, sample_gather_indices=None, commit_order=COMMIT_ORDER)
        cur, step_s, step_tok, step_thr = len(p), [], [], []
        # step 0: prefill-bearing; then one steady step per thread count of the sweep; then 2 more at the fastest count, so
        # that the baseline is the MEDIAN of 3 steady steps at the best thread count (single samples of 5-8 s steps scatter
        # by more than the reference-vs-port difference they are quoted next to: VERDICT r03 weak #8)
        n_total = max(2, n_steps, 1 + len(sweep) + 2)
        for i in range(n_total):
            if i == 0:
                thr = sweep[len(sweep) // 2]
            elif i <= len(sweep):
                thr = sweep[i - 1]
            else:
                seen = {}
                for sec_, thr_ in zip(step_s[1:], step_thr[1:]):
                    seen.setdefault(thr_, []).append(sec_)
                thr = min(seen, key=lambda t_: min(seen[t_]))
            torch.set_num_threads(thr)
            t1 = time.perf_counter()
            tree.construct_grow_map()
            valid, _, _, term = tree.verify()
            step_s.append(time.perf_counter() - t1)
            step_tok.append(valid.shape[0] - cur)
            step_thr.append(thr)
            cur = valid.shape[0]
            if term:
                break
        n_steady = len(step_s) - 1
        by_thr = {}
        for sec, thr in zip(step_s[1:], step_thr[1:]):
            by_thr.setdefault(thr, []).append(sec)
        import statistics
        mean_by_thr = {t: statistics.median(v) for t, v in by_thr.items()}          # (median: 3 samples at the best count)
        best_thr = max(by_thr, key=lambda t: (len(by_thr[t]), -mean_by_thr[t])) if by_thr else step_thr[0]
        best_s = mean_by_thr.get(best_thr)
        tok_per_step = (sum(step_tok[1:]) / n_steady) if n_steady else None
        # the imported reference beside this port on the same weights / prompt / noise (oracle/ref_cpu_baseline.py, run in
        # the build container: the reference checkout does not travel): seconds-per-step ratio, to scale `value`
        ref_over_port = ref_record = None
        try:
            import glob
            newest = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_cpu_reference_vs_port.json")))[-1]
            with open(newest) as f:
                rp = json.load(f)
            ref_over_port = rp["reference"]["steps_per_s"] / rp["port"]["steps_per_s"]
            ref_record = os.path.basename(newest)          # which round's container run the ratio comes from
        except (OSError, KeyError, ValueError, ZeroDivisionError, IndexError):
            pass
        return dict(value=(tok_per_step / best_s) if n_steady else None, unit="tokens/s", cores=best_thr,
                    kind="port", commit_order=COMMIT_ORDER,
                    sample=f"{len(step_s)} speculation steps of prompt 0, config {cfg['draft']} -> {cfg['target']}, the "
                           f"reference's torch op sequences on CPU fp16 tensors; step 0 (with the 255-token target prefill) "
                           f"{step_s[0]:.1f} s, then {n_steady} steady steps: one per thread count of {sweep}, two more at the "
                           f"fastest; median seconds / step by thread count { {t: round(v, 2) for t, v in mean_by_thr.items()} }; "
                           f"value = mean tokens/step of the steady steps / the MEDIAN of the {len(by_thr.get(best_thr, []))} steps at "
                           f"{best_thr} threads (+{build_s:.0f} s weight init)",
                    samples_at_best=[round(x, 3) for x in by_thr.get(best_thr, [])],
                    steps_per_s=(1.0 / best_s) if n_steady else None, prefill_step_s=step_s[0],
                    step_seconds=[round(x, 3) for x in step_s], step_threads=step_thr, step_tokens=step_tok,
                    seconds_per_step_by_threads={str(t): round(v, 3) for t, v in mean_by_thr.items()},
                    reference_over_port=ref_over_port, reference_over_port_record=ref_record, host_cores=avail, tokens=valid[:cur].tolist())
    finally:
        ops_mod.set_ops_for_testing(prev)
        torch.set_num_threads(prev_threads)


def allreduce_timing(target, device, rows, reps=40):
    """Tensor-parallel runs (collective: every rank calls it): one all-reduce of the verify forward's [rows, hidden] fp16
    message, HIP events on the launch stream, for the engine's xGMI kernel (if it is active) and for RCCL."""
    import torch.distributed as dist
    inner = target.engine
    hidden = inner.model.dims.hidden_size
    x = torch.zeros((rows, hidden), dtype=torch.float16, device=device)
    out = dict(kind=getattr(inner, "allreduce_kind", "rccl"), message_bytes=x.numel() * 2, per_verify=2 * inner.model.dims.num_hidden_layers)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) * 1e3 / reps], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)
    from sequoia_amd.Engine import xgmi_allreduce as XA
    if getattr(inner, "xgmi", None) is not None:
        out["xgmi_us"] = timeit(lambda: inner.xgmi(x))
        out["xgmi_status"] = inner.xgmi.status()
        out["xgmi_fault_word"] = int(inner.xgmi.fault[0]) if inner.xgmi.fault is not None else None
        out["xgmi_self_check"] = "passed (all-reduce, all-gather, all-reduce + RMSNorm against torch.distributed at set-up)"
        out["workspace"] = XA.WS_MODE
    else:
        out["xgmi_self_check"] = "not running on the xGMI kernels: " + (XA.LAST_REFUSAL or "SEQUOIA_TP_ALLREDUCE=rccl")
    out[("rccl" if dist.get_backend() == "nccl" else dist.get_backend()) + "_us"] = timeit(lambda: dist.all_reduce(x))
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks (one process per GPU) and relay their output;
    rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def tp_extra(n: int, args) -> dict:
    """Configuration E beside the replica headline when several GPUs are available: the 70B target tensor-parallel over
    the same N GPUs (whole-step graphs, collectives on the xGMI kernels, RCCL as their fallback), as a CHILD job with a
    timeout -- a stuck collective cannot take the headline line with it.  A failed or timed-out first attempt is retried
    once on RCCL collectives only (SEQUOIA_TP_ALLREDUCE=rccl).  Returns the child's JSON line (trimmed) or an error record."""
    first = _tp_child(n, args, {})
    if "error" not in first:
        return first
    if os.environ.get("SEQUOIA_TP_REQUIRE_XGMI", "0") == "1":
        first["note"] = "SEQUOIA_TP_REQUIRE_XGMI=1: no retry on RCCL"      # fail loudly, not silently on the fallback
        return first
    second = _tp_child(n, args, {"SEQUOIA_TP_ALLREDUCE": "rccl"}, timeout_s=int(os.environ.get("SEQUOIA_TP_RETRY_TIMEOUT", "150")))
    second["first_attempt"] = dict(collectives="xgmi", **{k: first[k] for k in ("error", "stderr") if k in first})
    return second


def _tp_child(n: int, args, extra_env: dict, timeout_s: int = 0) -> dict:
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--config", "E", "--steps", str(min(args.steps, 12)),
           "--warmup", "2", "--no-cpu-baseline", "--no-autoregressive", "--no-tuned-growmap", "--no-tp-extra",
           "--backend", args.backend]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                        "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                        "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")}
    env["SEQUOIA_TS_EXCLUSIVE"] = "1"          # one copy of the 70B shard per rank
    env.update(extra_env)
    import signal
    from types import SimpleNamespace
    # own session: on a timeout the whole tree (launcher + ranks) is killed by process group, nothing keeps a GPU
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        so, se = proc.communicate(timeout=timeout_s or int(os.environ.get("SEQUOIA_TP_EXTRA_TIMEOUT", "240")))
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.communicate()
        return dict(error="timeout")
    out = SimpleNamespace(stdout=so, stderr=se, returncode=proc.returncode)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        return dict(error=f"rc {out.returncode}", stderr=out.stderr[-400:])
    d = json.loads(lines[-1])
    keep = ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "mean_accepted_len", "rccl_ranks", "config",
            "roofline", "step_roofline", "allreduce", "prefill_steps_in_timed_region", "value_steady", "steady_ms_per_step",
            "tp_bytes_per_rank")
    res = {k: d[k] for k in keep if k in d}
    if "config" in d:
        res["step_loop"] = d["config"].get("step_loop")
    ar = d.get("allreduce") or {}
    # what the collectives actually ran on, at the top level: a first multi-GPU run that fell back to RCCL must be readable
    # as such from the line alone (VERDICT r03 #4b)
    res["allreduce_kind"] = ar.get("kind")
    res["xgmi_status"] = ar.get("xgmi_status")
    res["xgmi_self_check"] = ar.get("xgmi_self_check")
    res["collectives_env"] = extra_env.get("SEQUOIA_TP_ALLREDUCE", os.environ.get("SEQUOIA_TP_ALLREDUCE", "xgmi"))
    return res


def selftest(args, world, rank):
    """Launcher / rendezvous / aggregation check without a model: K timed no-op steps per rank, the same barrier +
    max-over-ranks timing and the same JSON assembly as the real run (used by the CPU test of the N > 1 path)."""
    import torch.distributed as dist
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    secs = time.perf_counter() - t0
    ranks = 1
    if world > 1:
        dist.barrier()
        t = torch.tensor([secs]); dist.all_reduce(t, op=dist.ReduceOp.MAX); secs = float(t)
        c = torch.tensor([float(args.steps)]); dist.all_reduce(c); steps_all = float(c)
        ranks = dist.get_world_size()
    else:
        steps_all = float(args.steps)
    if rank == 0:
        print(json.dumps(dict(metric="accepted tokens/sec", value=None, unit="tokens/s", n_gpus=world, steps=args.steps,
                              warmup=args.warmup, ms_per_step=secs / args.steps * 1e3, higher_is_better=True,
                              scaling="weak", vs_baseline=None, dtype="f16", data="synthetic", selftest=True,
                              rccl_ranks=ranks, steps_per_s=steps_all / secs,
                              config=dict(workload="launcher selftest (no model)", parallelism="replicas" if world > 1 else "single"))))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="B", choices=sorted(MODELS))
    ap.add_argument("--pair", default="calibrated", choices=["calibrated", "random"])
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-tuning", action="store_true", help="leave PyTorch's GEMM algorithm choice at its default")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--growmap", default=None, help="override the config's growmap: bundled name or path (.json / reference .pt)")
    ap.add_argument("--no-autoregressive", action="store_true", help="skip the target-only baseline (simulation_baseline)")
    ap.add_argument("--no-tuned-growmap", action="store_true",
                    help="skip the second timed loop on the growmap searched for this GPU (config B only)")
    ap.add_argument("--sync-loop", action="store_true",
                    help="drive every step from the host (reference API: construct_grow_map + verify with one result read "
                         "per step) instead of the device-driven whole-step graphs")
    ap.add_argument("--no-kernel-rooflines", action="store_true",
                    help="profiling aid: skip the per-kernel micro-timings (and with them `roofline` / `kernels`), so that a "
                         "rocprofv3 trace of this command contains the loop's own dispatches only")
    ap.add_argument("--commit-order", default="reference", choices=["reference", "lossless"],
                    help="the metric is the reference harness's (tests/testbed.py:45-95), so its loop runs the reference's commit "
                         "order by default -- the token stream, and with it tokens/step of the timed window, is the reference's; "
                         "the package's own default is `lossless` (Tree/_native_tree.py); steps/s is the same in both")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="single-GPU config B: skip the configs C and D runs that follow the headline (`other_configs`)")
    ap.add_argument("--other-steps", type=int, default=20, help="timed steps of each `other_configs` run")
    ap.add_argument("--no-reference-metric", action="store_true",
                    help="skip the whole-prompt run behind `value_reference_metric` / `prefill_step_ms`")
    ap.add_argument("--steady-window", action="store_true",
                    help="time K steps wherever the warm-up left the loop (steady steps only for K <= ~30) instead of starting "
                         "the timed window at a fresh prompt")
    ap.add_argument("--no-config-e", action="store_true", help="skip configuration E (70B target at TP = 1) in `other_configs`")
    ap.add_argument("--no-tp-extra", action="store_true",
                    help="N > 1: skip the secondary run of configuration E (70B target tensor-parallel over the N GPUs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--selftest", action="store_true", help="launcher / aggregation check without a model (CPU-runnable)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SEQUOIA_BENCH_ONE_DEVICE", "0") == "1":
        local = 0              # test rig: every rank on GPU 0 (with --backend gloo) to run the N > 1 code paths on a 1-GPU box
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group("gloo")
    if args.selftest:
        return selftest(args, world, rank)
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    torch.manual_seed(17 + rank)

    from sequoia_amd.Tree import _native_tree as _NT
    _NT.COMMIT_ORDER = args.commit_order      # every tree built from here on (the loops, the CPU baseline) commits in this order
    cfg = dict(MODELS[args.config])
    if args.growmap:
        cfg["growmap"] = args.growmap
    gemm_tuned = False
    if not args.no_gemm_tuning and not (cfg.get("tp") and world > 1):      # TP: per-rank TunableOp picks would make the
        #                                                                     replicated draft differ between ranks
        from sequoia_amd import gemm_tuning
        gemm_tuned = gemm_tuning.enable()
    draft, target, gm = build(cfg, device, args.pair)
    tp_mode = bool(cfg.get("tp"))
    prompts = load_prompts()[rank::world] if (world > 1 and not tp_mode) else load_prompts()
    if tp_mode:
        torch.manual_seed(17)          # identical noise on every rank: replicated decisions, no broadcast
    loop = Loop(cfg, draft, target, gm, device, prompts, use_graphs=not args.no_graphs,
                pipelined=not args.sync_loop and not args.no_graphs)

    from sequoia_amd.Tree._native_tree import QUIRK_STEPS
    commit_order = args.commit_order
    loop.run_steps(args.warmup)
    # The timed window is the reference's metric (tests/testbed.py:78-95): K consecutive speculation steps of the harness
    # loop BEGINNING WITH A FRESH PROMPT, so the prefill-bearing first verify of a prompt (255 rows through the target) is
    # inside it, as it is inside the reference's timer; `value_steady` / `steady_ms_per_step` are the same window without
    # its prefill-bearing steps (rounds 1-4 quoted that as `value`).  --steady-window keeps the old window.
    if not args.steady_window:
        loop.start_fresh_prompt()
    QUIRK_STEPS[0] = 0           # counted over the timed steps only (config.commit_order_quirk_steps)
    if tp_mode and world > 1:
        from sequoia_amd.Engine.ts_linear import assert_same_plans_across_ranks
        assert_same_plans_across_ranks(draft.engine.model, target.engine.model)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    p0, ps0, pt0 = loop.prefill_steps, loop.prefill_seconds, loop.prefill_tokens
    secs, new_tok, steps = loop.run_steps(args.steps)
    torch.cuda.synchronize()
    prefill_steps = loop.prefill_steps - p0      # steps of the timed region that carried a prompt's target prefill
    prefill_secs, prefill_tok = loop.prefill_seconds - ps0, loop.prefill_tokens - pt0
    steady_steps = max(steps - prefill_steps, 1)
    steady_ms = (secs - prefill_secs) / steady_steps * 1e3
    value_steady = (new_tok - prefill_tok) / max(secs - prefill_secs, 1e-9)
    rccl_ranks = 1
    allreduce = None
    if world > 1:
        dist.barrier()
        if tp_mode:
            allreduce = allreduce_timing(target, device, gm.size)
        t = torch.tensor([secs], device=device); dist.all_reduce(t, op=dist.ReduceOp.MAX); secs = float(t)
        rccl_ranks = dist.get_world_size()
        if tp_mode:                    # all ranks produced the SAME tokens: count them once
            steps_all = steps
        else:
            c = torch.tensor([float(new_tok), float(steps)], device=device); dist.all_reduce(c)
            new_tok, steps_all = float(c[0]), float(c[1])
    else:
        steps_all = steps

    if rank == 0 and args.no_kernel_rooflines:
        print(json.dumps(dict(metric="accepted tokens/sec", value=new_tok / secs, unit="tokens/s", n_gpus=world, steps=args.steps,
                              warmup=args.warmup, ms_per_step=secs / args.steps * 1e3, higher_is_better=True, dtype="f16",
                              data="synthetic", mean_accepted_len=new_tok / steps_all, roofline=None, cpu_baseline=None,
                              value_steady=value_steady, steady_ms_per_step=steady_ms, prefill_steps_in_timed_region=prefill_steps,
                              note="--no-kernel-rooflines: profiling aid, not a benchmark line",
                              config=dict(workload=f"config {args.config}", commit_order=commit_order))))
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return
    if rank == 0:
        kr = kernel_rooflines(cfg, loop, device)
        per_step = {k: v["seconds"] * (v["launches_per_step"] if (k == "tree_attention_target" or k.startswith("linear_ts_")) else 1)
                    for k, v in kr.items()}
        dom = max(per_step, key=per_step.get)
        d = kr[dom]
        peak_hbm = 8000.0
        # HBM bytes per launch and MFMA utilisation of the dominant kernel: rocprofv3 PMC passes committed under profiles/
        # (tools/pmc_r04.sh; FETCH_SIZE / WRITE_SIZE / SQ group in separate passes, gfx950 correction 2 FETCH + WRITE), keyed
        # by the launch plan THIS run used -- a plan the passes did not cover gives traffic = null and says so
        traffic = mfma_util = None
        traffic_note = None
        pmc_key = None
        if args.config == "B":
            if dom == "tree_attention_target":
                pmc_key = "tree_attention_target7b"
            elif dom.startswith("linear_ts_"):
                pmc_key = f"{dom[len('linear_ts_'):]}@{(gm.size + 15) // 16}:{d['plan'][0]}x{d['plan'][1]}"
        pmc_file = None
        if pmc_key is not None:
            traffic, mfma_util, pmc_file, traffic_note = pmc_lookup(pmc_key, dom)
            if traffic_note:
                print("bench.py: " + traffic_note, file=sys.stderr)
        roof = dict(bound="hbm", kernel=dom, achieved=d["bytes"] / d["seconds"] / 1e9, peak=peak_hbm, unit="GB/s",
                    frac=d["bytes"] / d["seconds"] / 1e9 / peak_hbm, traffic=traffic, mfma_util=mfma_util, pmc_key=pmc_key,
                    pmc_file=pmc_file,
                    avg_launch_us=d["seconds"] * 1e6, algorithmic_bytes_per_launch=d["bytes"],
                    time_per_step_us=per_step[dom] * 1e6)
        if traffic_note:
            roof["traffic_note"] = traffic_note
        kernels = {k: dict(avg_us=v["seconds"] * 1e6, gbps=v["bytes"] / v["seconds"] / 1e9,
                           frac=v["bytes"] / v["seconds"] / 1e9 / peak_hbm, algorithmic_bytes=v["bytes"],
                           launches_per_step=v["launches_per_step"], per_step_us=per_step[k] * 1e6,
                           **{kk: v[kk] for kk in ("inputs", "plan", "kv_len") if kk in v}) for k, v in kr.items()}
        if args.config == "B" and cfg["mode"] == "stochastic":
            # PMC traffic of the north-star kernels (per step), next to their algorithmic bytes
            for k, rec in pmc_northstar([len(lv.row_ids) for lv in gm.levels]).items():
                if k in kernels:
                    kernels[k].update(rec)
                    if rec.get("traffic"):
                        kernels[k]["traffic_over_algorithmic"] = round(rec["traffic"] / kernels[k]["algorithmic_bytes"], 3)
        tuned = None
        tuned_name = "MI355X-synthetic-68m-7b-stochastic"
        if not args.no_tuned_growmap and world == 1 and args.config == "B" and not args.growmap and args.pair == "calibrated":
            # the same loop on the growmap sequoia_amd.growmap_tuning searched for this GPU and this (synthetic)
            # model pair -- the headline `value` stays on the growmap BASELINE.json names
            from sequoia_amd.growmap import GrowMap
            gm2 = GrowMap.load(tuned_name)
            draft.clear_kv(); target.clear_kv()
            loop2 = Loop(cfg, draft, target, gm2, device, prompts, use_graphs=not args.no_graphs,
                         pipelined=not args.sync_loop and not args.no_graphs)
            loop2.run_steps(args.warmup)
            s2, t2, k2 = loop2.run_steps(args.steps)
            tuned = dict(growmap=tuned_name, nodes=gm2.size, value=t2 / s2, unit="tokens/s", ms_per_step=s2 / k2 * 1e3,
                         mean_accepted_len=t2 / k2, steps=k2)
        host_loop = None
        if loop.pipelined and world == 1:
            # the same loop driven from the host (reference API: construct_grow_map + verify, one result read per
            # step): what the device-driven step graphs buy, measured on the same box and the same prompts
            draft.clear_kv(); target.clear_kv()
            torch.manual_seed(17 + rank)
            loop3 = Loop(cfg, draft, target, gm, device, prompts, use_graphs=not args.no_graphs, pipelined=False)
            loop3.run_steps(args.warmup)
            s3, t3, k3 = loop3.run_steps(args.steps)
            host_loop = dict(ms_per_step=s3 / k3 * 1e3, value=t3 / s3, unit="tokens/s", steps=k3)
        autoreg = None
        if not args.no_autoregressive and world == 1:
            # the reference's own comparison point (tests/testbed.py:99-143): the target alone, 1 token / forward
            from sequoia_amd.harness import AutoregressiveLoop
            draft.clear_kv(); target.clear_kv()
            autoreg = AutoregressiveLoop(cfg, target, device, prompts).run(3)
            autoreg["speedup"] = (new_tok / secs) / autoreg["tokens_per_s"]
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = cpu_baseline(cfg, args.cpu_steps, args.pair)
                cpu.pop("tokens", None)
            except Exception as e:  # the baseline is a report, never the measured path
                cpu = dict(value=None, unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                           sample=f"failed: {type(e).__name__}: {e}")
        # the reference's own metric (tests/testbed.py:78-95): total_time / tokens over WHOLE prompts, each prompt's first
        # verify carrying the target prefill of the 128-token prompt -- `value` above is steady steps (the driver's 20 timed
        # steps never reach a second prompt); both are printed, with the time of a prefill-bearing step
        ref_metric = None
        if not args.no_reference_metric and world == 1:
            draft.clear_kv(); target.clear_kv()
            torch.manual_seed(17 + rank)
            loop4 = Loop(cfg, draft, target, gm, device, prompts, use_graphs=not args.no_graphs,
                         pipelined=not args.sync_loop and not args.no_graphs)
            loop4.run_prompts(1)                                   # warm-up prompt (graphs, plans)
            pf_s0, pf_n0 = loop4.prefill_seconds, loop4.prefill_steps
            s4, t4, k4 = loop4.run_prompts(3)
            pf_n = loop4.prefill_steps - pf_n0
            ref_metric = dict(value=t4 / s4, unit="tokens/s", prompts=3, steps=k4, tokens=t4, seconds=s4,
                              prefill_steps=pf_n, prefill_step_ms=(loop4.prefill_seconds - pf_s0) / max(pf_n, 1) * 1e3,
                              steady_ms_per_step=(s4 - (loop4.prefill_seconds - pf_s0)) / max(k4 - pf_n, 1) * 1e3,
                              note="whole prompts, 128 prompt tokens -> 256 tokens, per-prompt setup (tree constructor, draft "
                                   "prefill) outside the timer like tests/testbed.py:67-79; the first verify of every prompt "
                                   "carries the 255-row target prefill")
            del loop4
        # the package's DEFAULT commit order (`lossless`) on the same loop: what a user who does not ask for the reference's
        # token stream gets -- same kernels and steps/s, its own token stream and therefore tokens / step
        lossless = None
        if commit_order != "lossless" and world == 1 and not args.no_reference_metric and cfg["mode"] == "stochastic":
            _NT.COMMIT_ORDER = "lossless"
            try:
                draft.clear_kv(); target.clear_kv()
                torch.manual_seed(17 + rank)
                loop5 = Loop(cfg, draft, target, gm, device, prompts, use_graphs=not args.no_graphs,
                             pipelined=not args.sync_loop and not args.no_graphs)
                loop5.run_steps(args.warmup)
                if not args.steady_window:
                    loop5.start_fresh_prompt()
                s5, t5, k5 = loop5.run_steps(args.steps)
                lossless = dict(commit_order="lossless (package default)", value=t5 / s5, unit="tokens/s", ms_per_step=s5 / k5 * 1e3,
                                mean_accepted_len=t5 / k5, steps=k5)
                del loop5
            finally:
                _NT.COMMIT_ORDER = commit_order
        wb = step_weight_bytes(loop, gm)
        step_roof = dict(weight_bytes=wb["total"], target_bytes=wb["target"], draft_bytes_per_forward=wb["draft"],
                         draft_forwards=wb["draft_forwards"], achieved=wb["total"] / (steady_ms * 1e-3) / 1e9, peak=8000.0,
                         unit="GB/s", frac=wb["total"] / (steady_ms * 1e-3) / 8e12,
                         note="projection + lm_head weight bytes one speculation step streams (rank-local), over steady_ms_per_step")
        other = None
        if (not args.no_other_configs and world == 1 and args.config == "B" and not args.growmap and args.pair == "calibrated"
                and not args.sync_loop and not args.no_graphs):
            other = {}
            try:
                other["C"], _ = run_other_config("C", args, device, prompts, engines=(draft, target), steps=args.other_steps)
            except Exception as e:
                other["C"] = dict(error=f"{type(e).__name__}: {e}")
            import gc
            for oc, osteps in (("D", args.other_steps), ("E", min(args.other_steps, 12))):
                if oc == "E" and args.no_config_e:
                    continue
                try:
                    other[oc], eng_o = run_other_config(oc, args, device, prompts, steps=osteps, warmup=5 if oc == "D" else 3)
                    from sequoia_amd.Tree.step_graph import StepState
                    StepState.release(eng_o[1])
                    del eng_o
                except Exception as e:
                    other[oc] = dict(error=f"{type(e).__name__}: {e}")
                gc.collect(); torch.cuda.empty_cache()
        line = dict(metric="accepted tokens/sec", value=new_tok / secs, unit="tokens/s", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=secs / args.steps * 1e3,
                    higher_is_better=True, scaling="strong" if tp_mode else "weak", vs_baseline=None, dtype="f16", data="synthetic",
                    config=dict(workload=f"config {args.config}: {cfg['draft']} -> {cfg['target']} architectures "
                                         f"({args.pair} random-init weights), growmap {os.path.basename(str(cfg['growmap']))} "
                                         f"({gm.size}-node tree), T=0.6, top_p=1.0, M={cfg['M']}, 128-token c4_small "
                                         f"prompts, generate to 256",
                                parallelism=(f"tp{world}" if tp_mode else ("replicas" if world > 1 else "single")), graphs=not args.no_graphs,
                                commit_order=commit_order, commit_order_quirk_steps=QUIRK_STEPS[0],
                                commit_order_note="the reference harness's own order (its token stream, bonus-id quirk included, "
                                                  "Tree/SpecTree.py:222-224); the package default is `lossless` -- same kernels, same "
                                                  "steps/s, one flag bit in the walker",
                                step_loop="device-driven (one hipGraph per speculation step, results read one step late)"
                                if loop.pipelined else "host-driven (one result read per step)",
                                gemm="tree forwards (<= 144 rows): sq_linear_ts_f16 (fragment-major weight stream, plans "
                                     "ts_plans_gfx950.json); prompt prefill and lm_head at 128 rows: PyTorch GEMM ("
                                     + ("TunableOp-selected hipBLASLt / rocBLAS solutions" if gemm_tuned else "default algorithm") + ")"),
                    mean_accepted_len=new_tok / steps_all, steps_per_s=steps_all / secs, rccl_ranks=rccl_ranks,
                    prefill_steps_in_timed_region=prefill_steps,
                    prefill_step_ms_in_timed_region=(prefill_secs / prefill_steps * 1e3) if prefill_steps else None,
                    value_steady=value_steady, steady_ms_per_step=steady_ms,
                    value_note="`value` = tokens / seconds over the K timed steps of the harness loop starting at a fresh prompt: "
                               "the reference's metric (tests/testbed.py:78-95, the prefill-bearing first verify inside the timer); "
                               "`value_steady` = the same window without its prefill-bearing steps (what rounds 1-4 printed as `value`); "
                               "`value_reference_metric` = the same metric over 3 whole prompts",
                    allreduce=allreduce,
                    roofline=roof, step_roofline=step_roof, kernels=kernels,
                    tp_bytes_per_rank=(tp_bytes_per_rank(dict(wb, target=wb["target"] * world,
                                                              draft=wb["draft"] * (world if os.environ.get("SEQUOIA_TP_DRAFT", "0") == "1" else 1)))
                                       if tp_mode else None),
                    value_reference_metric=ref_metric["value"] if ref_metric else None,
                    prefill_step_ms=ref_metric["prefill_step_ms"] if ref_metric else None, reference_metric=ref_metric,
                    lossless_commit_order=lossless, other_configs=other, host_driven_loop=host_loop, mi355x_growmap=tuned,
                    autoregressive_baseline=autoreg,
                    cpu_baseline=cpu)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not tp_mode and not args.no_tp_extra:
            # the other ranks are exiting: their GPUs are free for the tensor-parallel child job
            del loop, draft, target
            import gc
            gc.collect()                     # step states (graphs, static buffers) hang off the target engine
            torch.cuda.empty_cache()
            time.sleep(3.0)
            spent = time.perf_counter() - T_START
            if spent > float(os.environ.get("SEQUOIA_TP_EXTRA_AFTER", "360")):
                line["tp_70b"] = dict(error=f"skipped: {spent:.0f} s already spent on the replica run")
            else:
                line["tp_70b"] = tp_extra(world, args)
        print(json.dumps(line))


if __name__ == "__main__":
    main()
