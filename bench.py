#!/usr/bin/env python
"""bench.py — accepted tokens/sec of the Sequoia speculation loop on MI355X.

This file: argument parsing, the timed window and the JSON line.  benchmarks/kernels.py: per-kernel timings + PMC records;
benchmarks/configs.py: the configurations that ride in the line (C, D, E at TP = 1) and the byte accounting; benchmarks/cpu.py:
`cpu_baseline`; benchmarks/launch.py: rank spawning, the tensor-parallel child job, the launcher self-test.

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
from benchmarks.configs import run_other_config, step_weight_bytes, tp_bytes_per_rank  # noqa: E402
from benchmarks.cpu import cpu_baseline  # noqa: E402
from benchmarks.kernels import in_loop_duration, kernel_rooflines, pmc_lookup, pmc_northstar  # noqa: E402
from benchmarks.launch import allreduce_timing, selftest, spawn_ranks, tp_extra  # noqa: E402


T_START = time.perf_counter()


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
    secs_local = secs
    if world > 1:
        dist.barrier()
        if tp_mode:
            allreduce = allreduce_timing(target, device, gm.size)
        t = torch.tensor([secs], device=device); dist.all_reduce(t, op=dist.ReduceOp.MAX); secs = float(t)
        rccl_ranks = dist.get_world_size()
        if tp_mode:                    # all ranks produced the SAME tokens: count them once
            steps_all = steps
        else:
            c = torch.tensor([float(new_tok), float(steps), float(new_tok - prefill_tok)], device=device); dist.all_reduce(c)
            new_tok, steps_all = float(c[0]), float(c[1])
            # `value_steady` is whole-job like `value`: steady tokens of all replicas over the slowest replica's steady time
            st = torch.tensor([secs_local - prefill_secs], device=device); dist.all_reduce(st, op=dist.ReduceOp.MAX)
            value_steady = float(c[2]) / max(float(st), 1e-9)
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
        loop_us, loop_file = in_loop_duration(dom, d.get("plan"), gm.size) if args.config == "B" else (None, None)
        if loop_us is not None:
            # the same kernel between its real neighbours (rocprofv3 kernel trace of the loop alone, committed under profiles/)
            roof.update(avg_launch_us_in_loop=loop_us, frac_in_loop=d["bytes"] / (loop_us * 1e-6) / 1e9 / peak_hbm, in_loop_record=loop_file)
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
        cpu, cpu_keep = None, {}
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu_keep = {}
                cpu = cpu_baseline(cfg, args.cpu_steps, args.pair, keep_engines=cpu_keep)
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
            for oc in ("A", "C"):                 # the same 68m / 7B engines on BASELINE.json configs[0] (2-chain) and configs[2] (8x8 greedy)
                try:
                    other[oc], _ = run_other_config(oc, args, device, prompts, engines=(draft, target), steps=args.other_steps)
                except Exception as e:
                    other[oc] = dict(error=f"{type(e).__name__}: {e}")
            if not args.no_cpu_baseline and "error" not in other["A"]:
                # configs[0] is the case the reference runs on a CPU: the port's time for it on this host beside the GPU's
                try:
                    from sequoia_amd.growmap import GrowMap as _GM
                    from sequoia_amd.harness import MODELS as _M
                    eng = cpu_keep.get("engines")
                    if eng is not None:
                        eng[0].clear_kv(); eng[1].clear_kv()
                    cb = cpu_baseline(dict(_M["A"]), n_steps=3, pair=args.pair,
                                      engines=None if eng is None else (eng[0], eng[1], _GM.load(_M["A"]["growmap"])))
                    other["A"]["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "steps_per_s",
                                                                          "prefill_step_s", "steady_step_s")}
                    other["A"]["gpu_over_cpu_port"] = (other["A"]["value_steady"] / cb["value"]) if cb.get("value") else None
                except Exception as e:
                    other["A"]["cpu_baseline"] = dict(error=f"{type(e).__name__}: {e}")
            cpu_keep.clear()
            import gc
            for oc, osteps in (("D", args.other_steps), ("L", args.other_steps), ("E", min(args.other_steps, 12))):
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
