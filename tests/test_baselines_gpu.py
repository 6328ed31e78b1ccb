"""SpecInferTree / GreedySTree kernels (sq_sample_iid_f16, sq_verify_specinfer_f16, sq_verify_tokens_f16) against the
oracle and the reference's traces, and the two trees end to end on the GPU."""
import numpy as np
import pytest
import torch

from conftest import load_trace
from helpers import assert_replay_complete, check_replay, note_escape, replay_trace, split_margin
from oracle import ops_np as O
from test_hip_kernels import csr, dev, random_tree

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sequoia_amd.ops import HipOps
    return HipOps()


@pytest.mark.parametrize("V,n_rows,k,gain,seed", [(1024, 9, 8, 3.0, 0), (32000, 34, 13, 2.0, 1), (32000, 1, 64, 6.0, 2)])
def test_sample_iid_matches_oracle(ops, V, n_rows, k, gain, seed):
    rng = np.random.RandomState(seed)
    logits = (rng.randn(n_rows + 3, V) * gain).astype(np.float16)
    rows = rng.permutation(n_rows + 3)[:n_rows].astype(np.int32)
    u = rng.randint(0, 1 << 24, (n_rows, k)).astype(np.int32)
    branch = rng.randint(1, k + 1, n_rows).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(branch)[:-1]]).astype(np.int32)
    out = torch.full((int(branch.sum()) + 4,), -7, dtype=torch.int64, device=DEV)
    ops.sample_iid(dev(logits), dev(u), dev(rows), k, 0.6, out, branch=dev(branch), out_off=dev(off))
    got = out.cpu().numpy()
    want = O.sample_iid(logits[rows], u, k, 0.6)
    assert (got[int(branch.sum()):] == -7).all()
    total = diff = 0
    for i in range(n_rows):
        g = got[off[i]:off[i] + branch[i]]
        w = want[i, :branch[i]]
        total += len(g)
        for a, b in zip(g, w):
            if a != b:
                # the kernel's softmax may differ from the oracle's in the last fp16 ulp of a few elements: a draw whose
                # uniform lands within that sliver of a CDF boundary moves to the neighbouring token with mass
                q = O.scaled_softmax_f16(logits[rows[i]][None], 0.6)[0]
                lo, hi = sorted((int(a), int(b)))
                assert (q[lo + 1:hi] == 0).all(), (i, a, b)
                diff += 1
    assert diff <= max(1, total // 50)


def _run(ops, fn_name, target, draft, tokens, r16, succ, gt, T, u24):
    n = len(succ)
    off, ids = csr(succ)
    d_tokens, d_draft = dev(tokens), dev(draft)
    ws = ops.verify_workspace(n, DEV)
    res = torch.zeros(64 + n, dtype=torch.int32, device=DEV)
    getattr(ops, fn_name)(dev(target), d_draft, d_tokens, dev(r16), dev(off), dev(ids) if len(ids) else None, n, gt, T, u24, ws, res)
    return res.cpu().numpy(), d_tokens.cpu().numpy(), d_draft.cpu().numpy()


def test_verify_specinfer_on_reference_trace(ops):
    z, meta = load_trace("F_specinfer")
    succ = meta["successors"]
    for s in range(int(z["n_steps"])):
        gt = int(z[f"step{s}/gt"])
        tokens, draft, target = z[f"step{s}/tokens_pre"].copy(), z[f"step{s}/draft_logits_pre"].copy(), z[f"step{s}/target_logits"]
        u24 = int(z["bonus_u24"][s])
        res, tok_after, draft_after = _run(ops, "verify_specinfer", target, draft, tokens, z["r"], succ, gt, meta["T"], u24)
        margins = []
        o_tokens = tokens.copy()
        want = O.verify_specinfer(target, draft.copy(), o_tokens, z["r"], succ, gt, meta["T"], u24, margins=margins)
        assert np.array_equal(draft_after, draft)                    # the draft logits are never modified
        assert res[0] == want["accept_len"], (s, res[:8], want["accept_len"])      # committed fixture: fail-closed
        a = want["accept_len"]
        assert list(res[8:8 + res[1]]) == want["slots"] and res[3] == want["terminal"]
        assert np.array_equal(tok_after[:a + 1], o_tokens[:a + 1])
        assert np.array_equal(tok_after[:a + 1], z[f"step{s}/valid_tokens"][:a + 1])      # == the reference itself


@pytest.mark.parametrize("V,n,seed", [(1024, 40, 0), (32000, 65, 1)])
def test_verify_specinfer_random(ops, V, n, seed):
    rng = np.random.RandomState(seed)
    succ = random_tree(rng, n, max_children=6)
    gt, M, T = 30, 30 + n + 8, 0.6
    agree = total = 0
    for trial in range(6):
        target = (rng.randn(n, V) * 2).astype(np.float16)
        draft = (target.astype(np.float32) + rng.randn(n, V) * 1.0).astype(np.float16)
        tokens = rng.randint(3, V, M).astype(np.int64)
        # children: draws with replacement from the parent's draft distribution (duplicates happen)
        for p, ch in enumerate(succ):
            if ch:
                d = O.sample_iid(draft[p][None], rng.randint(0, 1 << 24, (1, len(ch))), len(ch), T)[0]
                tokens[np.asarray(ch) + gt - 1] = d
        r16 = rng.rand(M).astype(np.float16)
        u24 = int(rng.randint(0, 1 << 24))
        res, tok_after, _ = _run(ops, "verify_specinfer", target, draft, tokens, r16, succ, gt, T, u24)
        margins = []
        o_tokens = tokens.copy()
        want = O.verify_specinfer(target, draft.copy(), o_tokens, r16, succ, gt, T, u24, margins=margins)
        total += 1
        if res[0] == want["accept_len"] and list(res[8:8 + res[1]]) == want["slots"]:
            agree += 1
            a = want["accept_len"]
            assert np.array_equal(tok_after[:a], o_tokens[:a])
        else:       # fresh random inputs: the paths may part at ONE decision whose own margin is inside one fp16 ulp of p
            m = split_margin(succ, gt, want["slots"], [int(x) for x in res[8:8 + res[1]]], margins)
            assert m is not None and abs(m) < 1e-3, f"trial {trial}: paths split at a decision with margin {m}"
            note_escape(f"test_verify_specinfer_random V={V} n={n} seed={seed} trial {trial}", m)
    assert agree >= total - 1


@pytest.mark.parametrize("n,seed", [(1, 0), (65, 1), (128, 2)])
def test_verify_tokens_matches_oracle(ops, n, seed):
    rng = np.random.RandomState(seed)
    succ = random_tree(rng, n, max_children=5) if n > 1 else [[]]
    gt, M = 25, 25 + n + 8
    off, ids = csr(succ)
    for trial in range(5):
        tokens = rng.randint(3, 50, M).astype(np.int64)
        tgt = rng.randint(3, 50, n).astype(np.int64)
        # make a path exist with decent probability
        node = 0
        while succ[node] and rng.rand() < 0.8:
            c = succ[node][rng.randint(len(succ[node]))]
            tokens[c + gt - 1] = tgt[node]
            node = c
        want_tokens = tokens.copy()
        want = O.verify_tokens(tgt, want_tokens, succ, gt)
        d_tokens = dev(tokens)
        ws = ops.verify_workspace(n, DEV)
        res = torch.zeros(64 + n, dtype=torch.int32, device=DEV)
        ops.verify_tokens(dev(tgt), d_tokens, dev(off), dev(ids) if len(ids) else None, n, gt, ws, res)
        r = res.cpu().numpy()
        assert r[0] == want["accept_len"] and r[1] == want["n_tree"] and r[2] == want["bonus"] and r[3] == want["terminal"]
        assert list(r[64:64 + r[1]]) == want["slots"]
        assert np.array_equal(d_tokens.cpu().numpy(), want_tokens)


def test_greedys_target_draw_on_reference_trace(ops):
    z, meta = load_trace("G_greedys")
    n = len(meta["successors"])
    for s in range(int(z["n_steps"])):
        out = torch.zeros(n, dtype=torch.int64, device=DEV)
        ops.sample_iid(dev(z[f"step{s}/target_logits"][:n]), dev(z["target_u24"][s].astype(np.int32).reshape(n, 1)), None, 1,
                       meta["T"], out)
        got, want = out.cpu().numpy(), z[f"step{s}/target_token"]
        assert (got != want).sum() <= 1


@pytest.mark.parametrize("name", ["F_specinfer", "G_greedys"])
def test_gpu_loop_follows_reference_trace(name):
    steps, tree, draft, target, z, meta = replay_trace(name, DEV)
    matched, diverged = check_replay(steps, z, meta)
    assert_replay_complete(name, steps, tree, z, meta, matched, diverged)
    print(f"{name}: {matched}/{int(z['n_steps'])} steps token-identical to the reference")
