"""The oracle against the LIVE reference, where its checkout is present (this build container; the GPU box has none, and this
file is CPU-only): the reference's own utils.py -- sampling_without_replacement, sampling_argmax, get_residual,
get_sampling_logits (utils.py:5-18,29-32,65-77) -- is loaded by path and run on many seeded random inputs beside
oracle/ops_np.py, with the comparison rules of the committed golden rows (tests/test_oracle_golden.py): identical
outputs except inside exact fp16 ties, whose order torch leaves unspecified.  The committed fixtures pin the oracle on a
handful of rows; this pins it on a few hundred, every time the CPU suite runs next to the reference."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import ops_np as O

REF_UTILS = "/root/reference/utils.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF_UTILS), reason="reference checkout not present")


@pytest.fixture(scope="module")
def RU():
    spec = importlib.util.spec_from_file_location("_sequoia_reference_utils", REF_UTILS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)            # imports torch / dataclasses only; nothing is installed into sys.modules
    return mod


CASES = [(1024, 3.0, 8, 0.6), (1024, 8.0, 19, 0.6), (4096, 1.0, 13, 1.0), (32000, 2.0, 19, 0.6), (32000, 6.0, 64, 0.6),
         (32000, 10.0, 4, 0.3)]


@pytest.mark.parametrize("V,gain,k,T", CASES)
def test_sampling_without_replacement_and_argmax(RU, V, gain, k, T):
    torch.manual_seed(V + k)
    n_rows = 24 if V <= 4096 else 6
    logits = (torch.randn(n_rows, V) * gain).half()
    rand = torch.empty(n_rows, V, dtype=torch.float16).uniform_()
    want = RU.sampling_without_replacement(logits, rand, k, T).reshape(n_rows, k).numpy()
    got = O.sample_wor(logits.numpy(), rand.numpy(), k, T)
    keys = O.sample_keys(logits.numpy(), rand.numpy(), T)
    diff = 0
    for r in range(n_rows):
        for s in range(k):
            if got[r, s] != want[r, s]:
                assert keys[r, got[r, s]] == keys[r, want[r, s]], (r, s)          # same key: an exact tie
                diff += int(np.isfinite(np.float32(keys[r, got[r, s]])))          # (peaked rows: q = 0 -> key = -inf for most tokens)
    assert diff <= max(2, n_rows * k // 20)                                       # ties between finite fp16 keys are rare
    want_a = RU.sampling_argmax(logits, k).reshape(n_rows, k).numpy()
    got_a = O.topk_ids(logits.numpy(), k)
    ln = logits.numpy()
    for r in range(n_rows):
        for s in range(k):
            if got_a[r, s] != want_a[r, s]:
                assert ln[r, got_a[r, s]] == ln[r, want_a[r, s]], (r, s)


@pytest.mark.parametrize("V,gain,T", [(1024, 4.0, 0.6), (32000, 2.0, 0.6), (32000, 8.0, 0.6), (32000, 3.0, 1.0)])
def test_get_residual(RU, V, gain, T):
    torch.manual_seed(7 * V + int(gain))
    for _ in range(8):
        lp = (torch.randn(V) * gain).half()
        lq = (lp.float() + torch.randn(V) * gain * 0.5).half()
        p = torch.softmax(lp / T, dim=-1)
        q = torch.softmax(lq / T, dim=-1)
        want = RU.get_residual(p.clone(), q.clone()).numpy()
        po, qo = O.scaled_softmax_f16(lp.numpy(), T), O.scaled_softmax_f16(lq.numpy(), T)
        # the softmax itself: torch's fp16 softmax accumulates in fp32; within 1 ulp, almost everywhere equal
        for mine, ref in ((po, p.numpy()), (qo, q.numpy())):
            a, b = mine.view(np.int16).astype(np.int32), ref.view(np.int16).astype(np.int32)
            assert np.abs(a - b).max() <= 1 and (a != b).mean() < 0.01
        res, _ = O.residual_f16(p.numpy(), q.numpy())                  # on the reference's own p, q: isolates get_residual
        a, b = res.view(np.int16).astype(np.int32), want.view(np.int16).astype(np.int32)
        assert np.abs(a - b).max() <= 2 and (a != b).mean() < 0.01


@pytest.mark.parametrize("V,gain,top_p,T", [(1024, 3.0, 0.9, 0.6), (32000, 2.0, 0.9, 0.6), (32000, 5.0, 0.5, 0.6),
                                            (32000, 9.0, 0.95, 0.6), (32000, 1.0, 0.3, 1.0)])
def test_get_sampling_logits(RU, V, gain, top_p, T):
    torch.manual_seed(int(V * top_p) + int(gain))
    logits = (torch.randn(4, V) * gain).half()
    want = RU.get_sampling_logits(logits.clone(), top_p, T).numpy()
    got = O.top_p_filter(logits.numpy(), top_p, T)
    from helpers import assert_top_p_equal_up_to_ties
    # identical up to the identity of equal-logit tokens at the cut (torch's unstable CPU sort of fp16); a probability that
    # differs from torch's by the last fp32 ulp of exp() could move the cut as well -- not observed on these rows
    assert_top_p_equal_up_to_ties(logits.numpy(), got, want, f"V={V} P={top_p}")
    p = O.scaled_softmax_f16(logits.numpy(), T).astype(np.float64)
    assert (np.where(np.isinf(got), 0, p).sum(1) >= min(top_p, p.max(1).min()) - 2e-3).all()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tree_search_against_the_live_script(seed):
    """sequoia_amd.tree_search against the reference's tree_search.py executed here (oracle/gen_tree_search_golden.py::
    run_reference: runpy on the read-only script) for random acceptance vectors and time tables beyond the four committed
    cases: same growmap, same (budget, depth) choice, bit-equal value table."""
    if not os.path.exists("/root/reference/tree_search.py") or not os.path.exists("/root/reference/acceptance-rate-vector.pt"):
        pytest.skip("reference checkout not present")
    from oracle.gen_tree_search_golden import run_reference
    from sequoia_amd import tree_search as ts
    rng = np.random.default_rng(100 + seed)
    width = int(rng.integers(4, 10))
    w = np.sort(rng.dirichlet(np.ones(width + 1) * rng.uniform(0.4, 1.2))).astype(np.float32)[::-1]
    p = [0.0] + w[:width].tolist() + [float(w[width])]
    budgets = sorted({int(b) for b in rng.integers(2, 36, size=4)} | {36})
    t0 = float(rng.uniform(2.0, 8.0))
    case = dict(p=p, max_depth=int(rng.integers(3, 8)), max_budget=36, draft_time=float(rng.uniform(0.05, 0.6)),
                valid_budget=budgets, target_time=[t0 * (1.0 + 0.01 * b) for b in budgets])
    exp = run_reference(case)
    cfg = {k: v for k, v in case.items() if k != "p"}
    cfg["acceptance_rate_vector"] = case["p"]
    g, report = ts.search(cfg)
    assert [report["budget"], report["depth"]] == exp["pair"]
    assert report["time_per_token"] == exp["dec_time"]
    assert g["Successors"] == exp["Successors"] and g["roots"] == exp["roots"] and g["branches"] == exp["branches"]
    tab = ts.search_tables(np.asarray(p, dtype=np.float32)[:-1], cfg["max_budget"], cfg["max_depth"])
    want = np.array([[-np.inf if x is None else x for x in row] for row in exp["results"]], dtype=np.float32)
    assert np.array_equal(tab.best, want)


# ---- fresh traces of the reference's SpecTree / GreedyTree (seeds outside the committed fixtures) ---------------------------
LIVE_SPECS = ["live:stochastic:301", "live:stochastic:302", "live:sequoia128:303", "live:greedy:304",
              "live:topp:310"]     # (+ SpecTree under the harness's default nucleus filter top_p = 0.9)
BASELINE_SPECS = ["live:specinfer:305", "live:greedys:306"]          # the paper's comparison baselines
PROBE_SPECS = ["live:spectest:307", "live:greedytest:308"]           # the acceptance-rate probes
# the reference's large growmaps (256 / 512 / 193 nodes) on fresh seeds: checked op by op on the oracle (the reference's own inputs,
# so no decision margin is involved); the host-loop replay of such trees is pinned on the committed, margin-screened L_* traces
LARGE_SPECS = ["live:s256:311", "live:s512:312", "live:l8x24:313"]
VOCAB_SPECS = ["live:v32k:309"]            # the real vocabulary: 68m-dims -> 160m-dims, config B's growmap, seeded weights, compact logits


@pytest.fixture(scope="module")
def live_traces(tmp_path_factory):
    """oracle/gen_golden.py run in a SUBPROCESS (the reference's top-level Engine / Tree / utils modules never enter this
    process) for a few seeds the committed fixtures do not contain: tiny dims, weights stored in the trace."""
    import json
    import subprocess
    import sys
    out = tmp_path_factory.mktemp("live_traces")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEQUOIA_GOLDEN_OUT=str(out))
    r = subprocess.run([sys.executable, os.path.join(repo, "oracle", "gen_golden.py")] + LIVE_SPECS + BASELINE_SPECS + PROBE_SPECS + VOCAB_SPECS + LARGE_SPECS, env=env, cwd=repo,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    traces = {}
    for spec in LIVE_SPECS + BASELINE_SPECS + PROBE_SPECS + VOCAB_SPECS + LARGE_SPECS:
        _, mode, seed = spec.split(":")
        z = np.load(os.path.join(str(out), f"trace_live_{mode}_{seed}.npz"))
        traces[spec] = (z, json.loads(bytes(z["meta_json"]).decode()))
    return traces


@pytest.mark.parametrize("spec", LIVE_SPECS + LARGE_SPECS)
def test_live_trace_sampler_and_verifier_on_the_oracle(live_traces, spec):
    """Every sampler call and every verification of a fresh reference run, op by op on the oracle (the checks of
    tests/test_oracle_golden.py::test_sampler_matches_reference / test_verify_*_matches_reference)."""
    z, meta = live_traces[spec]
    succ, T = meta["successors"], meta["T"]
    rejections = 0
    for s in range(int(z["n_steps"])):
        lvl = 0
        while f"step{s}/samp{lvl}/logits" in z:
            logits, want = z[f"step{s}/samp{lvl}/logits"], z[f"step{s}/samp{lvl}/out"]
            k = want.shape[0] // logits.shape[0]
            if meta["mode"] == "stochastic":
                rnd = z[f"step{s}/samp{lvl}/rand"]
                got, keys = O.sample_wor(logits, rnd, k, T), O.sample_keys(logits, rnd, T)
            else:
                got, keys = O.topk_ids(logits, k), logits
            want = want.reshape(got.shape)
            bad = got != want
            assert (keys[np.nonzero(bad)[0], got[bad]] == keys[np.nonzero(bad)[0], want[bad]]).all(), (spec, s, lvl)
            lvl += 1
        assert lvl > 0
        gt = int(z[f"step{s}/gt"])
        tokens = z[f"step{s}/tokens_pre"].copy()
        if meta["mode"] == "stochastic":
            draft = z[f"step{s}/draft_logits_pre"].copy()
            res = O.verify_stochastic(z[f"step{s}/target_logits"], draft, tokens, z["r"], succ, gt, T, int(z["bonus_u24"][s]))
            assert res["terminal"] == int(z[f"step{s}/terminal"])
            post = z[f"step{s}/draft_logits_post"]
            for t in [sl - (gt - 1) for sl in res["slots"]]:
                if len(succ[t]):
                    assert np.array_equal(draft[t], post[t]), (spec, s, t)        # the -65504 writes (Tree/SpecTree.py:156)
            rejections += int((draft != z[f"step{s}/draft_logits_pre"]).sum())     # rejected children: masked draft logits
            if f"step{s}/residual" in z:
                a16 = res["final_p"].view(np.int16).astype(np.int32)
                b16 = z[f"step{s}/residual"].view(np.int16).astype(np.int32)
                assert np.abs(a16 - b16).max() <= 2
        else:
            res = O.verify_greedy(z[f"step{s}/target_logits"], tokens, succ, gt)
        assert res["accept_len"] == int(z[f"step{s}/accept_len"]), (spec, s)
        valid = z[f"step{s}/valid_tokens"]
        assert np.array_equal(tokens[:valid.shape[0]], valid), (spec, s)
    if meta["mode"] == "stochastic":
        assert rejections > 0                                                     # residual updates did occur


@pytest.mark.parametrize("spec", LIVE_SPECS)
def test_live_trace_host_loop_replay(live_traces, spec):
    """The package's engines + trees on CPU with the oracle's ops reproduce the fresh reference run step by step (the
    check of tests/test_host_logic_cpu.py on the committed traces)."""
    from helpers import check_replay, replay_trace
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    z, meta = live_traces[spec]
    ops.set_ops_for_testing(OracleOps())
    try:
        steps, tree, draft, target, z, meta = replay_trace(None, "cpu", trace=(z, meta))
    finally:
        ops.set_ops_for_testing(None)
    assert len(steps) == int(z["n_steps"])
    matched, diverged = check_replay(steps, z, meta)
    assert diverged is None and matched == len(steps), f"{spec}: diverged at step {diverged}"
    last = len(steps) - 1
    assert draft.engine.kv_cache.kv_offset == int(z[f"step{last}/kv_draft"][2])
    assert target.engine.kv_cache.kv_offset == int(z[f"step{last}/kv_target"][2])


@pytest.mark.parametrize("spec", BASELINE_SPECS)
def test_live_trace_comparison_baselines(live_traces, spec):
    """Fresh runs of the reference's SpecInferTree / GreedySTree: the draws with replacement, the `p >= r q` walk, the sampled
    target tokens and the token-equality walk on the oracle (the checks of tests/test_baselines_cpu.py)."""
    from test_baselines_cpu import check_greedys_trace, check_specinfer_trace
    z, meta = live_traces[spec]
    (check_specinfer_trace if meta["mode"] == "specinfer" else check_greedys_trace)(z, meta)


@pytest.mark.parametrize("spec", PROBE_SPECS)
def test_live_trace_acceptance_probes(live_traces, spec):
    """Fresh runs of the reference's SpecTreeTest / GreedyTreeTest (fp32 noise, p >= r q in fp32, the 5-tuple) on the oracle."""
    from test_probe_cpu import check_greedytest_trace, check_spectest_trace
    z, meta = live_traces[spec]
    if meta["mode"] == "spectest":
        accepted, rejected_all = check_spectest_trace(z, meta)
        assert accepted + rejected_all == int(z["n_steps"])
    else:
        check_greedytest_trace(z, meta)


@pytest.mark.parametrize("spec", VOCAB_SPECS)
def test_live_trace_real_vocabulary(live_traces, spec):
    """A fresh SpecTree run at V = 32000 (68m-dims draft -> 160m-dims target on the A100-CNN-68m-7b growmap): the oracle's
    sampler on the root rows and its verifier on the rows of the walked path (the checks of the committed V32k_seq128 /
    B_7b traces, tests/test_oracle_golden.py)."""
    from test_oracle_golden import check_compact_sampler, check_compact_verify
    z, meta = live_traces[spec]
    assert check_compact_sampler(z, meta, spec) == int(z["n_steps"])
    margins = check_compact_verify(z, meta, spec)
    assert len(margins) > 0
