"""Growmaps for THIS GPU: measure what the growmap search needs, then run it.

The reference ships growmaps searched for A100 / L40 timings; the three inputs of the search are
  1. the acceptance-rate vector  P[k-th child drawn without replacement is the accepted one]
     — reference: tests/test_accept.py:36-140 on SpecTreeTest / GreedyTreeTest (Tree/SpecTree.py:283-489),
     a one-level tree of `max_width` children, counting which child the verifier accepts;
  2. the time of one draft inference, and
  3. the verification time per tree budget — reference: Engine/offloading_profile.py and hand-edited
     config files (demo-config.json).
Here all three are measured on the production path itself (the HIP sampler / verifier / attention kernels
and the hipGraph-replayed forwards): (1) runs the ordinary SpecTree / GreedyTree on a star growmap and reads
which child the verifier's result record names, (2) and (3) time construct_grow_map() and verify() of real
speculation steps on a representative growmap of each budget.  The result feeds sequoia_amd.tree_search.

    python -m sequoia_amd.growmap_tuning --config B --out gpurun_out/MI355X-68m-7b-stochastic.json

Configuration E (70B target tensor-parallel over N GPUs; the reference's analogue is Engine/offloading_profile.py:1-48 + its
missing `L40-*-7b-70b-*` growmaps) runs under the launcher, one rank per GPU; every rank measures (the forwards are
collective), rank 0 writes the growmap in the reference's format (tree_search.py:121-128):

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m sequoia_amd.growmap_tuning \
        --config E --out growmaps/MI355X-TP8-7b-70b-stochastic.json
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np
import torch

from . import tree_search
from .growmap import GrowMap
from .harness import MODELS, AutoregressiveLoop, Loop, build, load_prompts
from .native import SQ_RES_LAST_NODE, SQ_RES_N_TREE


def star_growmap(width: int) -> GrowMap:
    """Root with `width` children (SpecTreeTest's tree, Tree/SpecTree.py:307-308)."""
    return GrowMap.from_successors([list(range(1, width + 1))] + [[] for _ in range(width)])


def measure_acceptance_vector(cfg, draft, target, device, prompts, width: int = 32, steps: int = 400,
                              T: float = 0.6, top_p: float = 1.0, use_graphs: bool = True, vocab: int = 32000) -> np.ndarray:
    """[0, p_1 .. p_width, p_none] — the layout tests/test_accept.py:86-89 stores and tree_search.py:14
    reads.  Greedy mode counts non-terminal steps only, like simulation_greedy (:118-121)."""
    loop = Loop(cfg, draft, target, star_growmap(width), device, prompts, use_graphs=use_graphs, T=T, top_p=top_p,
                vocab=vocab)
    counts = np.zeros(width + 1, dtype=np.float64)
    greedy = cfg["mode"] != "stochastic"

    def on_step(tree, terminate):
        if greedy and terminate:
            return
        res = tree.last_result
        accepted_child = int(res[SQ_RES_LAST_NODE]) - 1 if int(res[SQ_RES_N_TREE]) > 0 else width
        counts[accepted_child] += 1

    loop.run_steps(steps, on_step)
    vec = np.zeros(width + 2, dtype=np.float32)
    vec[1:] = counts / max(counts.sum(), 1.0)
    return vec


def _sync():
    torch.cuda.synchronize()


def _progress(msg: str):
    """Stage markers on stderr (a tuning run takes minutes and ends on one JSON line)."""
    import sys
    print(f"growmap_tuning: {msg}", file=sys.stderr, flush=True)


def measure_autoregressive_time(cfg, target, device, prompts, n_prompts: int = 2) -> float:
    """Seconds per token of the target-only baseline (harness.AutoregressiveLoop = simulation_baseline,
    tests/testbed.py:99-143)."""
    return AutoregressiveLoop(cfg, target, device, prompts).run(n_prompts)["ms_per_token"] * 1e-3


def measure_step_times(cfg, draft, target, device, prompts, budgets, p_vec, max_depth: int = 10, steps: int = 16,
                       warmup: int = 4, T: float = 0.6, vocab: int = 32000):
    """Per budget b: the search's best b-node tree under flat times, then `steps` real speculation steps with
    a device sync between the two phases.  Returns (draft_time, {b: target_time}, {b: detail}), seconds:
    draft_time = construct_grow_map() / (levels - 1), target_time = verify() - draft_time (verify() ends with
    the 1-token draft forward of prepare_for_next_iter, the search's cost model counts it as a draft step)."""
    tab = tree_search.search_tables(np.asarray(p_vec, dtype=np.float32)[:-1], max(budgets), max_depth)
    per_level, target_time, detail = [], {}, {}
    for b in budgets:
        depth = int(np.argmax(tab.best[b] / (np.arange(max_depth + 1) * 0.02 + 1.0)))      # mild depth penalty
        gm = GrowMap.from_successors(tree_search.build_growmap(tab, b, depth)["Successors"])
        _progress(f"step times: budget {b} (depth {depth}, level sizes {[lv.total for lv in gm.levels]})")
        loop = Loop(cfg, draft, target, gm, device, prompts, use_graphs=True, T=T, vocab=vocab)
        t_grow = t_verify = 0.0
        done = 0
        loop.run_steps(warmup)
        while done < steps:
            if loop.tree is None:
                loop._new_prompt()
            _sync(); t0 = time.perf_counter()
            loop.tree.construct_grow_map()
            _sync(); t1 = time.perf_counter()
            valid, _, _, terminate = loop.tree.verify()
            _sync(); t2 = time.perf_counter()
            t_grow += t1 - t0; t_verify += t2 - t1; done += 1
            loop.cur_len = valid.shape[0]
            if terminate or loop.cur_len >= loop.max_new or int(valid[-1]) in (0, 2):
                loop.tree = None
        levels = gm.draft_step - 1
        d_time = t_grow / done / max(levels, 1)
        per_level.append(d_time)
        target_time[b] = t_verify / done - d_time
        detail[b] = dict(depth=depth, levels=levels, grow_ms=t_grow / done * 1e3, verify_ms=t_verify / done * 1e3)
        draft.clear_kv(); target.clear_kv()
    return float(np.median(per_level)), target_time, detail


def tune(config_name: str = "B", pair: str = "calibrated", device: str = "cuda:0", width: int = 32,
         accept_steps: int = 400, budgets=(2, 4, 8, 16, 32, 48, 64, 96, 128), max_depth: int = 10,
         time_steps: int = 16, static_rows: int = 0):
    """static_rows > 0: the acceptance vector comes from the teacher-forced estimator (acceptance_static.py, the
    reference's tests/fast_test.py) over that many 256-token rows instead of star-tree speculation steps."""
    import os
    cfg = dict(MODELS[config_name])
    tp = bool(cfg.get("tp")) and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if not tp:
        from . import gemm_tuning
        gemm_tuning.enable(tune_missing=False)      # shipped winners for the 128-row shapes, defaults elsewhere
        # (tensor-parallel ranks must run identical arithmetic: no per-rank GEMM tuning there)
    if tp:
        torch.manual_seed(17)                       # identical noise on every rank: replicated decisions
    draft, target, _ = build(cfg, device, pair)
    prompts = load_prompts()
    with torch.inference_mode():
        if static_rows > 0:
            from .acceptance_static import static_acceptance_vector
            # (the bundled c4_small rows hold 128 tokens: positions 64..127 are evaluated; the reference evaluates
            # positions 128..255 of 256-token rows, tests/fast_test.py:63)
            rows = [p[:256] for p in prompts[:static_rows]]
            a = static_acceptance_vector(draft, target, rows, k=width, T=0.6, top_p=1.0, draft_top_p=1.1,
                                         start=min(128, min(len(r) for r in rows) // 2), device=device)
            p_vec = np.zeros(width + 2, dtype=np.float32)
            p_vec[:width + 1] = a.numpy()
            p_vec[width + 1] = max(0.0, 1.0 - float(a.sum()))
        else:
            _progress(f"acceptance vector: {accept_steps} steps on a {width}-child star tree")
            p_vec = measure_acceptance_vector(cfg, draft, target, device, prompts, width, accept_steps)
        draft.clear_kv(); target.clear_kv()
        _progress("autoregressive baseline")
        t_ar = measure_autoregressive_time(cfg, target, device, prompts)
        d_time, t_time, detail = measure_step_times(cfg, draft, target, device, prompts, list(budgets), p_vec,
                                                    max_depth, time_steps)
    search_cfg = dict(acceptance_rate_vector=p_vec.tolist(), max_depth=max_depth, max_budget=max(budgets),
                      draft_time=d_time, valid_budget=[1] + list(budgets),
                      target_time=[t_ar] + [t_time[b] for b in budgets])
    g, report = tree_search.search(search_cfg)
    report.update(config=config_name, pair=pair, autoregressive_ms=t_ar * 1e3, draft_ms=d_time * 1e3,
                  target_ms={str(b): t_time[b] * 1e3 for b in budgets}, detail={str(b): v for b, v in detail.items()},
                  acceptance_vector=p_vec.tolist(), search_config=search_cfg,
                  predicted_tokens_per_s=1.0 / report["time_per_token"])
    return g, report


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B", choices=sorted(MODELS))
    ap.add_argument("--pair", default="calibrated", choices=["calibrated", "random"])
    ap.add_argument("--width", type=int, default=32, help="children of the star tree (test_accept.py --W)")
    ap.add_argument("--accept-steps", type=int, default=400)
    ap.add_argument("--time-steps", type=int, default=16)
    ap.add_argument("--max-depth", type=int, default=10)
    ap.add_argument("--budgets", type=int, nargs="+", default=[2, 4, 8, 16, 32, 48, 64, 96, 128])
    ap.add_argument("--out", required=True, help="growmap destination (.json successors fixture or reference-format .pt)")
    ap.add_argument("--static-rows", type=int, default=0,
                    help="> 0: acceptance vector from the teacher-forced estimator (tests/fast_test.py) over this many rows")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args(argv)
    import os
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SEQUOIA_BENCH_ONE_DEVICE", "0") == "1":
        local = 0                                   # test rig: every rank on GPU 0 (with --backend gloo)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local)
    g, report = tune(args.config, args.pair, f"cuda:{local}", args.width, args.accept_steps, tuple(args.budgets), args.max_depth,
                     args.time_steps, args.static_rows)
    report["tp_world"] = world if MODELS[args.config].get("tp") else 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
    tree_search.save_growmap(g, args.out)
    with open(args.out.rsplit(".", 1)[0] + ".report.json", "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: report[k] for k in ("budget", "depth", "expected_accepted", "predicted_tokens_per_s",
                                             "speedup_vs_autoregressive", "autoregressive_ms", "draft_ms", "target_ms")}))


if __name__ == "__main__":
    main()
