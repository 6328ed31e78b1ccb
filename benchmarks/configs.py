"""The configurations that ride in the headline's line (`other_configs`: C, D and E at TP = 1) and the byte accounting of a
speculation step (`step_roofline`, `tp_bytes_per_rank`)."""
from __future__ import annotations

import time

import torch

from sequoia_amd.harness import MODELS, Loop, build

from .kernels import kernel_rooflines, pmc_lookup


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
    # The draft forward over the LAST tree level stops after its last layer's KV write (Tree/step_graph.py::KV_ONLY_LAST_LEVEL:
    # leaves have no children to sample): it streams neither the last layer's o / gate_up / down nor lm_head.
    skipped = 0
    try:
        from sequoia_amd.Engine.ts_linear import MAX_ROWS
        from sequoia_amd.Tree.step_graph import KV_ONLY_LAST_LEVEL
        m = loop.draft.engine.model
        levels = list(gm.levels) if hasattr(gm, "levels") else []
        last_rows = int(getattr(levels[-1], "total", 0)) if levels else 0
        if KV_ONLY_LAST_LEVEL and getattr(m, "ts", None) is not None and levels and 0 < last_rows <= MAX_ROWS \
                and getattr(loop, "pipelined", True):
            sh = m.ts.shapes
            skipped = m.weights.lm_head.numel() * 2 + sum((2 if sh[n][2] else 1) * sh[n][0] * sh[n][1] * 2 for n in ("o", "gate_up", "down"))
    except Exception:
        skipped = 0
    return dict(target=t, draft=d, draft_forwards=n_draft_forwards, draft_skipped_last_level=skipped,
                total=t + d * n_draft_forwards - skipped)


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
    per_step = {k: v["seconds"] * v["launches_per_step"] if (k == "tree_attention_target" or k.startswith(("linear_ts_", "gemm_"))) else v["seconds"]
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
    if name in ("D", "E"):
        pk = None
        if dom.startswith("linear_ts_") and d.get("plan"):
            pk = f"{name}:{dom[len('linear_ts_'):]}@{(gm.size + 15) // 16}:{d['plan'][0]}x{d['plan'][1]}"
            traffic, mfma_util, pmc_file, note = pmc_lookup(pk, dom)
            out["roofline"].update(traffic=traffic, mfma_util=mfma_util, pmc_key=pk, pmc_file=pmc_file)
            if note:
                out["roofline"]["traffic_note"] = note
    del loop
    return out, (draft, target)
