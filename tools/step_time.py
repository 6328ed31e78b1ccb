"""GPU time of one speculation step, config B: back-to-back replays of the whole-step hipGraph (device-driven loop, no
host in between) against the host-driven loop over the same steps.  python tools/step_time.py [config]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequoia_amd import gemm_tuning  # noqa: E402
from sequoia_amd.harness import MODELS, Loop, build, load_prompts  # noqa: E402

dev = "cuda:0"
cfg = dict(MODELS[sys.argv[1] if len(sys.argv) > 1 else "B"])
gemm_tuning.enable()
draft, target, gm = build(cfg, dev, "calibrated")
prompts = load_prompts()
N = 14

for mode in ("sync", "piped"):
    torch.manual_seed(17)
    draft.clear_kv(); target.clear_kv()
    loop = Loop(cfg, draft, target, gm, dev, prompts, pipelined=(mode == "piped"))
    loop.run_steps(1)                      # prompt 0, step 0 (prefill-bearing); piped: pipeline begun
    tree = loop.tree
    torch.cuda.synchronize()
    if mode == "sync":
        t0 = time.perf_counter()
        for _ in range(N):
            tree.construct_grow_map(); tree.verify()
        torch.cuda.synchronize()
        print(f"host-driven loop      : {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step (wall, {N} steps)")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(N):
            tree.state.launch()
        t_host = time.perf_counter() - t0
        e1.record(); torch.cuda.synchronize()
        print(f"whole-step graph      : {e0.elapsed_time(e1) / N:.3f} ms/step (GPU events, {N} replays back to back); "
              f"host time per replay {t_host / N * 1e3:.3f} ms")
        # the pipelined loop as the harness runs it
        torch.manual_seed(17)
        draft.clear_kv(); target.clear_kv()
        loop = Loop(cfg, draft, target, gm, dev, prompts, pipelined=True)
        loop.run_steps(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.run_steps(N)
        torch.cuda.synchronize()
        print(f"device-driven harness : {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step (wall, {N} steps)")
        # host timeline of the pipelined loop: enqueue / collect durations per step (two steps in flight)
        torch.manual_seed(17)
        draft.clear_kv(); target.clear_kv()
        loop = Loop(cfg, draft, target, gm, dev, prompts, pipelined=True)
        loop.run_steps(1)
        tree = loop.tree
        tree.begin_pipeline()
        torch.cuda.synchronize()
        enq, col = [], []
        t_start = time.perf_counter()
        tree.enqueue_step()
        for i in range(N):
            t0 = time.perf_counter()
            tree.enqueue_step()
            t1 = time.perf_counter()
            tree.collect_step()
            t2 = time.perf_counter()
            enq.append((t1 - t0) * 1e3); col.append((t2 - t1) * 1e3)
        tree.collect_step()
        torch.cuda.synchronize()
        total = (time.perf_counter() - t_start) / (N + 1) * 1e3
        print(f"timeline              : {total:.3f} ms/step; enqueue ms {[round(x, 2) for x in enq[:6]]}, collect ms {[round(x, 2) for x in col[:6]]}")
        # the same graph, one launch + device synchronisation per step (no events, no copy stream)
        torch.manual_seed(17)
        draft.clear_kv(); target.clear_kv()
        loop = Loop(cfg, draft, target, gm, dev, prompts, pipelined=True)
        loop.run_steps(1)
        tree = loop.tree
        tree.begin_pipeline()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            tree.state.launch()
            torch.cuda.synchronize()
        print(f"launch + synchronize  : {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step")
        t0 = time.perf_counter()
        for i in range(N // 2):
            tree.state.launch(); tree.state.launch()
            torch.cuda.synchronize()
        print(f"2 launches + sync     : {(time.perf_counter() - t0) / (N // 2 * 2) * 1e3:.3f} ms/step")
