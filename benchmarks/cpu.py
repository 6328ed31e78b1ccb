"""`cpu_baseline` of the bench line: the CPU path timed on the GPU box's host cores.  The ONLY place outside tests/ and
__graft_entry__.smoke() that touches oracle/ -- as the thing that is timed beside the product, never as the product."""
from __future__ import annotations

import json
import os
import time

import torch

from sequoia_amd.harness import build, load_prompts

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_baseline(cfg, n_steps=3, pair="calibrated", engines=None, numpy_ops=False, keep_engines=None):
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
        if isinstance(keep_engines, dict):            # the caller times another configuration on the same CPU engines (config A)
            keep_engines["engines"] = (draft, target)
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
                    sample=f"{len(step_s)} speculation steps of prompt 0, config {cfg['draft']} -> {cfg['target']}, growmap "
                           f"{cfg.get('growmap')} ({gm.size} nodes), the "
                           f"reference's torch op sequences on CPU fp16 tensors; step 0 (with the {127 + gm.size}-token target prefill) "
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
