"""Growmap tuner and autoregressive baseline on the GPU (HIP kernels + hipGraph replays), tiny models."""
import numpy as np
import pytest
import torch

from conftest import load_trace
from helpers import build_engines

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engines(name="B_seq128"):
    z, meta = load_trace(name)
    draft, target = build_engines(z, meta, DEV)
    return draft, target, meta, [[int(t) for t in z["prompt"]]]


def test_acceptance_vector_on_hip_path_matches_oracle_path():
    """Same seeds -> the star-tree acceptance counts of the HIP path and of the CPU oracle path agree
    (decisions may differ only where a margin sits inside the fp16 logit tolerance: allow 2 of 16 steps)."""
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import growmap_tuning as gt
    from sequoia_amd import ops
    draft, target, meta, prompts = _engines()
    cfg = dict(mode="stochastic", M=meta["M"])
    torch.manual_seed(3)
    hip = gt.measure_acceptance_vector(cfg, draft, target, DEV, prompts, width=6, steps=16, T=meta["T"],
                                       vocab=meta["vocab"])
    z, _ = load_trace("B_seq128")
    cdraft, ctarget = build_engines(z, meta, "cpu")
    ops.set_ops_for_testing(OracleOps())
    try:
        torch.manual_seed(3)
        ref = gt.measure_acceptance_vector(cfg, cdraft, ctarget, "cpu", prompts, width=6, steps=16, T=meta["T"],
                                           use_graphs=False, vocab=meta["vocab"])
    finally:
        ops.set_ops_for_testing(None)
    assert abs(hip[1:].sum() - 1) < 1e-6
    assert np.abs(hip - ref).sum() <= 2 * 2 / 16 + 1e-6


def test_step_times_and_search_produce_a_loadable_growmap():
    from sequoia_amd import growmap_tuning as gt
    from sequoia_amd import tree_search
    from sequoia_amd.growmap import GrowMap
    draft, target, meta, prompts = _engines()
    cfg = dict(mode="stochastic", M=meta["M"])
    p = np.array([0, 0.6, 0.15, 0.08, 0.04, 0.13], dtype=np.float32)
    with torch.inference_mode():
        d_time, t_time, detail = gt.measure_step_times(cfg, draft, target, DEV, prompts, [4, 8, 16], p, max_depth=5,
                                                       steps=4, warmup=1, T=meta["T"], vocab=meta["vocab"])
    assert d_time > 0 and all(t > 0 for t in t_time.values())
    g, rep = tree_search.search(dict(acceptance_rate_vector=p.tolist(), max_depth=5, max_budget=16, draft_time=d_time,
                                     valid_budget=[4, 8, 16], target_time=[t_time[b] for b in (4, 8, 16)]))
    assert GrowMap.from_successors(g["Successors"]).size == rep["budget"]


def test_autoregressive_loop_graph_replay_equals_eager():
    from sequoia_amd import harness
    draft, target, meta, prompts = _engines()
    cfg = dict(mode="stochastic", M=meta["M"])
    out = []
    for graphs in (True, False):
        torch.manual_seed(11)
        loop = harness.AutoregressiveLoop(cfg, target, DEV, prompts, T=meta["T"], max_steps=6, use_graphs=graphs,
                                          vocab=meta["vocab"])
        toks = []
        orig = loop.ops.sample_wor

        def spy(logits, rand, row_ids, k, T, o, _orig=orig, _t=toks, **kw):
            r = _orig(logits, rand, row_ids, k, T, o, **kw)
            _t.append(int(o[0]))
            return r
        loop.ops.sample_wor = spy
        try:
            loop.run_prompt()
        finally:
            loop.ops.sample_wor = orig
        out.append(toks)
    assert out[0] == out[1] and len(out[0]) >= 1


def test_static_acceptance_vector_on_native_forwards_matches_cpu_path():
    """The teacher-forced estimator (tests/fast_test.py) on the native forwards: the logits of the HIP forwards agree with
    the CPU engines' within fp16 tolerance, so with the draws pinned (CPU generator, logits moved to the host) the two
    vectors are close; layout [0, a_1 .. a_k] and the search accepts it."""
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    from sequoia_amd.acceptance_static import acceptance_from_logits, static_acceptance_vector
    from sequoia_amd.Engine.Llama_modules import TreeContext
    draft, target, meta, _ = _engines()
    torch.manual_seed(11)
    rows = [torch.randint(3, meta["vocab"], (96,)).tolist() for _ in range(2)]
    vec = static_acceptance_vector(draft, target, rows, k=5, T=meta["T"], top_p=1.0, draft_top_p=1.1, start=64, device=DEV)
    assert vec.shape == (6,) and float(vec[0]) == 0.0 and 0.0 < float(vec[1]) <= 1.0 and float(vec.sum()) <= 1.0 + 1e-5
    # same rows through the CPU engines (oracle ops), same CPU-generator draws on host copies of the logits
    z, _ = load_trace("B_seq128")
    cdraft, ctarget = build_engines(z, meta, "cpu")
    one_d, one_c = torch.ones((1, 1), dtype=torch.int64, device=DEV), torch.ones((1, 1), dtype=torch.int64)

    def logits(eng, ids, dev, one):
        t = torch.tensor(ids, dtype=torch.long, device=dev)
        pos = torch.arange(len(ids), device=dev)
        eng.clear_kv()
        return eng.inference(input_ids=t[None], storage_ids=pos, position_ids=pos[None], attn_mask=None,
                             tree=TreeContext(0, len(ids), 1, one, len(ids))).float().cpu()
    got, want = [], []
    for hip_side in (True, False):
        if not hip_side:
            ops.set_ops_for_testing(OracleOps())
        try:
            torch.manual_seed(5)
            tot, n = torch.zeros(5), 0
            for r in rows:
                tl = logits(target if hip_side else ctarget, r, DEV if hip_side else "cpu", one_d if hip_side else one_c)
                dl = logits(draft if hip_side else cdraft, r, DEV if hip_side else "cpu", one_d if hip_side else one_c)
                (got if hip_side else want).append((tl.clone(), dl.clone()))
                tot, c = acceptance_from_logits(tl, dl, None, 5, meta["T"], 1.0, 1.1, start=64, acc=tot)
                n += c
            (got if hip_side else want).append(tot / n)
        finally:
            ops.set_ops_for_testing(None)
    for (a, b), (c, d) in zip(got[:2], want[:2]):
        assert (a - c).abs().max() < 4e-2 and (b - d).abs().max() < 4e-2
    assert (got[2] - want[2]).abs().max() < 0.08          # 64 positions: a handful of draws may flip inside the logit tolerance
