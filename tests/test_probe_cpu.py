"""The acceptance-rate probes against traces of the reference's own SpecTreeTest / GreedyTreeTest
(Tree/SpecTree.py:283-481, Tree/GreedyTree.py:267-456; loop of tests/test_accept.py:36-140): the oracle's probe arithmetic
(fp32 noise, fp32 keys, p >= r q in fp32) on the reference's inputs, then the native classes' host logic on the oracle ops."""
import numpy as np
import pytest

from conftest import load_trace
from oracle import ops_np as O
from probe_helpers import replay_probe


@pytest.fixture()
def oracle_ops():
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    ops.set_ops_for_testing(OracleOps())
    yield
    ops.set_ops_for_testing(None)


def _star(w):
    return [list(range(1, w + 1))] + [[] for _ in range(w)]


def test_oracle_probe_sampler_and_verifier_match_reference():
    accepted, rejected_all = check_spectest_trace(*load_trace("P_spectest"))
    assert accepted >= 2 and rejected_all >= 2


def check_spectest_trace(z, meta):
    """(also applied to fresh traces of the live reference: tests/test_oracle_live_reference_cpu.py)"""
    w, T = meta["width"], meta["T"]
    accepted = rejected_all = 0
    for s in range(int(z["n_steps"])):
        gt = len(z[f"step{s}/prefix"])
        dl, tl = z[f"step{s}/draft_logits"], z[f"step{s}/target_logits"]
        got, keys = O.sample_wor_f32noise(dl[0:1], z[f"step{s}/rand32"][0:1], w, T)
        want = z[f"step{s}/tokens_pre"][gt:gt + w]
        for c in range(w):      # identical unless two fp32 keys tie exactly
            assert got[0, c] == want[c] or keys[0, got[0, c]] == keys[0, want[c]], (s, c)
        tokens = z[f"step{s}/tokens_pre"].copy()
        res = O.verify_probe(tl, dl.copy(), tokens, z[f"step{s}/r32"], _star(w), gt, T, int(z["bonus_u24"][s]))
        a, b, term = (int(x) for x in z[f"step{s}/a_b_terminal"])
        assert res["accept_len"] == a and res["terminal"] == term
        assert (res["last_node"] - 1 if res["n_tree"] else -1) == b
        valid = z[f"step{s}/valid_tokens"]
        assert np.array_equal(tokens[:len(valid)], valid), s
        accepted += b >= 0
        rejected_all += b < 0
    return accepted, rejected_all


def test_oracle_greedy_probe_matches_reference():
    check_greedytest_trace(*load_trace("Q_greedytest"))


def check_greedytest_trace(z, meta):
    w = meta["width"]
    for s in range(int(z["n_steps"])):
        gt = len(z[f"step{s}/prefix"])
        dl, tl = z[f"step{s}/draft_logits"], z[f"step{s}/target_logits"]
        got, want = O.topk_ids(dl[0:1], w)[0], z[f"step{s}/tokens_pre"][gt:gt + w]
        for c in range(w):          # torch.topk orders equal fp16 logits arbitrarily; the oracle by token id
            assert got[c] == want[c] or dl[0, got[c]] == dl[0, want[c]], (s, c)
        tokens = z[f"step{s}/tokens_pre"].copy()
        res = O.verify_greedy(tl, tokens, _star(w), gt)
        a, b, term = (int(x) for x in z[f"step{s}/a_b_terminal"])
        assert res["accept_len"] == a and (res["last_node"] - 1 if res["n_tree"] else -1) == b
        valid = z[f"step{s}/valid_tokens"]
        assert np.array_equal(tokens[:len(valid)], valid)


@pytest.mark.parametrize("name", ["P_spectest", "Q_greedytest"])
def test_native_probe_classes_reproduce_reference_5_tuples(oracle_ops, name):
    steps, z, meta, draft, target = replay_probe(name, "cpu")
    assert len(steps) == int(z["n_steps"])
    for s, rec in enumerate(steps):
        key = z[f"step{s}/draft_logits"][0]
        for c, (g_, w_) in enumerate(zip(rec["children"], rec["ref_children"])):
            assert g_ == w_ or (meta["mode"] == "greedytest" and key[g_] == key[w_]), f"{name} step {s}: child {c}"
        assert rec["abt"] == rec["ref_abt"] and rec["a2"] == rec["abt"][0], f"{name} step {s}: {rec['abt']} vs {rec['ref_abt']}"
        assert np.array_equal(rec["valid"], rec["ref_valid"]), f"{name} step {s}"
    # the KV protocol of the probes: both caches rolled back to the accepted path, no next-root forward
    a = steps[-1]["abt"][0]
    assert draft.engine.kv_cache.kv_offset == a and target.engine.kv_cache.kv_offset == a
