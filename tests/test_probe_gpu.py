"""The acceptance-rate probes on the GPU: sq_sample_wor_f32noise_f16 / sq_verify_probe_f16 against the oracle, and the native
SpecTreeTest / GreedyTreeTest replaying traces of the reference's own probe classes (5-tuples, sampled children)."""
import numpy as np
import pytest
import torch

from oracle import ops_np as O
from probe_helpers import replay_probe
from test_hip_kernels import csr, dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from sequoia_amd.ops import HipOps
    return HipOps()


@pytest.mark.parametrize("V,n_rows,k,gain", [(1024, 3, 8, 3.0), (32000, 1, 16, 4.0), (32000, 5, 32, 2.0)])
def test_sample_wor_f32noise_matches_oracle(ops, V, n_rows, k, gain):
    rng = np.random.RandomState(V + k)
    logits = (rng.randn(n_rows, V) * gain).astype(np.float16)
    rand = rng.random_sample((n_rows, V)).astype(np.float32)
    want, keys = O.sample_wor_f32noise(logits, rand, k, 0.6)
    out = torch.zeros(n_rows * k, dtype=torch.int64, device=DEV)
    ops.sample_wor_f32noise(dev(logits), dev(rand), None, k, 0.6, out)
    got = out.cpu().numpy().reshape(n_rows, k)
    bad = np.argwhere(got != want)
    for r, c in bad:      # exp / log last-ulp differences: a differing pick must be a near-tie of the fp32 keys
        a, b = keys[r, got[r, c]], keys[r, want[r, c]]
        assert abs(a - b) <= 4e-6 * abs(b), (r, c, a, b)
    assert len(bad) <= max(1, got.size // 50)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_verify_probe_matches_oracle_on_star_trees(ops, seed):
    rng = np.random.RandomState(seed)
    V, w, gt, T, M = 1024, 8, 9, 0.6, 40
    succ = [list(range(1, w + 1))] + [[] for _ in range(w)]
    off, ids = csr(succ)
    tl = (rng.randn(w + 1, V) * 3).astype(np.float16)
    dl = (tl.astype(np.float32) * (0.2 if seed % 2 else 0.9) + rng.randn(w + 1, V) * 1.5).astype(np.float16)
    r32 = rng.random_sample(M).astype(np.float32)
    rand = rng.random_sample((1, V)).astype(np.float32)
    tokens = np.zeros(M, np.int64)
    tokens[:gt] = rng.randint(3, V, gt)
    tokens[gt:gt + w] = O.sample_wor_f32noise(dl[0:1], rand, w, T)[0][0]
    margins = []
    o_tok, o_dl = tokens.copy(), dl.copy()
    want = O.verify_probe(tl, o_dl, o_tok, r32, succ, gt, T, 424242, margins=margins)
    d_tok, d_dl = dev(tokens), dev(dl)
    ws = ops.verify_workspace(w + 1, DEV)
    res = torch.zeros(64 + w + 1, dtype=torch.int32, device=DEV)
    from sequoia_amd.native import SQ_VERIFY_GATHER_FIRST
    ops.verify_probe(dev(tl), d_dl, d_tok, dev(r32), dev(off), dev(ids), w + 1, gt, T, 424242 | SQ_VERIFY_GATHER_FIRST, ws, res)
    r = res.cpu().numpy()
    if r[0] != want["accept_len"]:
        # fresh random inputs: the walk of a star tree is one level of decisions; the kernel may leave the oracle only at
        # the decision where the two paths part, and only if that decision's own margin is inside one fp16 ulp of p
        from helpers import note_escape, split_margin
        m = split_margin(succ, gt, want["slots"], [int(x) for x in r[8:8 + r[1]]], margins)
        assert m is not None and abs(m) < 1e-3, f"seed {seed}: paths split at a decision with margin {m}"
        note_escape(f"test_verify_probe_matches_oracle_on_star_trees seed={seed}", m)
        return
    assert r[0] == want["accept_len"] and r[1] == want["n_tree"] and r[3] == want["terminal"]
    assert (r[6] - 1 if r[1] else -1) == (want["last_node"] - 1 if want["n_tree"] else -1)
    assert np.array_equal(d_tok.cpu().numpy()[:want["accept_len"]], o_tok[:want["accept_len"]])
    if r[1] == 0 and want["bonus"] >= 0:       # all rejected: the bonus comes from the 8-fold residual
        assert abs(int(r[2]) - want["bonus"]) <= 3 or r[2] == want["bonus"]


@pytest.mark.parametrize("name", ["P_spectest", "Q_greedytest"])
def test_gpu_probe_classes_replay_reference_traces(name):
    steps, z, meta, draft, target = replay_probe(name, DEV)
    same = 0
    for s, rec in enumerate(steps):
        if rec["abt"] != rec["ref_abt"] or not np.array_equal(rec["valid"], rec["ref_valid"]):
            break
        same += 1
    # greedy: integer work, every step; stochastic: every step of this trace reproduces on the GPU (decisions are not
    # margin-limited: checked against the oracle's margins when the trace was generated)
    assert same == int(z["n_steps"]), f"{name}: {same} of {int(z['n_steps'])} steps reproduce ({steps[same]['abt']} vs {steps[same]['ref_abt']})"
