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
