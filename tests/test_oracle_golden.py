"""Pin the numpy oracle (oracle/ops_np.py) against traces of the reference itself.

The fixtures under tests/golden/ were produced by oracle/gen_golden.py, which imports and runs
the reference (Tree/SpecTree.py, Tree/GreedyTree.py, utils.py, Engine/*) on CPU.  Every test
feeds the oracle the *inputs the reference saw* and requires the reference's outputs:
bit-exact for token / index / integer results, and within one fp16 ulp for probabilities
(the only source of difference is the exp() implementation inside torch's softmax).
"""
import numpy as np
import pytest

from conftest import (COMPACT_TRACES, GOLDEN, LARGE_COMPACT, LARGE_GREEDY, LARGE_STOCHASTIC, LARGE_TRACES, STOCHASTIC_TRACES,
                      TOPP_TRACES, TRACE_NAMES, load_trace)

COMPACT_STOCHASTIC = COMPACT_TRACES + TOPP_TRACES + ["B_7b", "D_13b_w4", "E_70b_w2", "D_13b", "E_70b_w8"] + LARGE_COMPACT        # + the headline-dims SpecTree trace (68m -> 7B dims)
from oracle import ops_np as O


@pytest.mark.parametrize("name", TRACE_NAMES + LARGE_TRACES)
def test_bitmask_equals_growmap_mask(name):
    z, meta = load_trace(name)
    succ = meta["successors"]
    bm = O.bitmask_from_successors(succ)
    n = len(succ)
    gt = int(z["step0/gt"])
    # the reference's first attention window (Tree/SpecTree.py:57-58), 0 / -65504
    win = z["mask_window0"]
    tot = gt + n - 1
    dense = O.tree_mask_dense(0, tot, tot, gt, n, bm)
    assert dense.shape == win.shape
    assert np.array_equal(dense, win)


def sampler_inputs(z, meta, s, lvl, roots):
    """(logits, rand) the reference's sampler of level lvl saw in step s: stored, or -- lean traces -- the rows
    draft_logits_pre[roots[lvl]] / rand[roots[lvl]] (Tree/SpecTree.py:103: the rows are final once their level ran)."""
    if f"step{s}/samp{lvl}/logits" in z:
        return z[f"step{s}/samp{lvl}/logits"], (z[f"step{s}/samp{lvl}/rand"] if meta["mode"] == "stochastic" else None)
    assert meta.get("lean")
    idx = roots[lvl]
    return z[f"step{s}/draft_logits_pre"][idx], (z["rand"][idx] if meta["mode"] == "stochastic" else None)


@pytest.mark.parametrize("name", TRACE_NAMES + LARGE_TRACES)
def test_sampler_matches_reference(name):
    from sequoia_amd.growmap import GrowMap
    z, meta = load_trace(name)
    succ = meta["successors"]
    roots = GrowMap.from_successors(succ).roots
    n_steps = int(z["n_steps"])
    checked = 0
    for s in range(n_steps):
        lvl = 0
        while f"step{s}/samp{lvl}/out" in z:
            logits, rnd = sampler_inputs(z, meta, s, lvl, roots)
            want = z[f"step{s}/samp{lvl}/out"]
            k = want.shape[0] // logits.shape[0]
            if meta["mode"] == "stochastic":
                got = O.sample_wor(logits, rnd, k, meta["T"])
                keys = O.sample_keys(logits, rnd, meta["T"])
            else:
                got = O.topk_ids(logits, k)
                keys = logits
            want = want.reshape(got.shape)
            # fp16 keys can tie exactly (typically at -inf when fewer than k tokens have a
            # representable key); torch.topk orders ties arbitrarily, the oracle by token id.
            # Require identity wherever the key is unique, key-equality inside a tie class.
            for r in range(got.shape[0]):
                for c in range(k):
                    if got[r, c] != want[r, c]:
                        assert keys[r, got[r, c]] == keys[r, want[r, c]], f"{name} step {s} level {lvl} row {r}"
            checked += 1
            lvl += 1
    assert checked > 0


def test_sampler_rows_full_vocab():
    z = np.load(f"{GOLDEN}/rows_v32000.npz")
    for i in range(4):
        logits, rand, k = z[f"wor{i}/logits"], z[f"wor{i}/rand"], int(z[f"wor{i}/k"])
        got = O.sample_wor(logits, rand, k, 0.6)
        want = z[f"wor{i}/out"].reshape(2, k)
        # keys are fp16: exact ties inside the top-k are ordered by torch's sort, which is
        # unspecified; require equality wherever the key is unique.
        keys = O.sample_keys(logits, rand, 0.6)
        for r in range(2):
            for s in range(k):
                if got[r, s] != want[r, s]:
                    assert keys[r, got[r, s]] == keys[r, want[r, s]], (i, r, s)
        # sampling_argmax (utils.py:29-32): identical wherever the fp16 logit is unique, same logit inside a tie class
        got_a, want_a = O.topk_ids(logits, k), z[f"wor{i}/argmax_out"].reshape(2, k)
        for r in range(2):
            for s in range(k):
                if got_a[r, s] != want_a[r, s]:
                    assert logits[r, got_a[r, s]] == logits[r, want_a[r, s]], (i, r, s)
        # residual (utils.py:5-8): within 1 fp16 ulp of the reference's
        p = O.scaled_softmax_f16(logits[0], 0.6)
        q = O.scaled_softmax_f16(logits[1], 0.6)
        res, _ = O.residual_f16(p, q)
        want_res = z[f"wor{i}/residual"]
        a = res.view(np.int16).astype(np.int32)
        b = want_res.view(np.int16).astype(np.int32)
        assert np.abs(a - b).max() <= 2
        assert (a != b).mean() < 0.01


@pytest.mark.parametrize("name", STOCHASTIC_TRACES + LARGE_STOCHASTIC)
def test_verify_stochastic_matches_reference(name):
    z, meta = load_trace(name)
    succ = meta["successors"]
    n = len(succ)
    u24 = z["bonus_u24"]
    r16 = z["r"]
    n_steps = int(z["n_steps"])
    for s in range(n_steps):
        gt = int(z[f"step{s}/gt"])
        tokens = z[f"step{s}/tokens_pre"].copy()
        draft = z[f"step{s}/draft_logits_pre"].copy()
        res = O.verify_stochastic(z[f"step{s}/target_logits"], draft, tokens, r16, succ, gt, meta["T"], int(u24[s]))
        a = int(z[f"step{s}/accept_len"])
        assert res["accept_len"] == a, f"{name} step {s}"
        assert res["terminal"] == int(z[f"step{s}/terminal"])
        valid = z[f"step{s}/valid_tokens"]
        assert np.array_equal(tokens[:valid.shape[0]], valid), f"{name} step {s}"
        # the -65504 writes into the draft rows of the walked nodes (Tree/SpecTree.py:156)
        if f"step{s}/draft_logits_post" in z:      # (lean traces keep the rows once)
            post = z[f"step{s}/draft_logits_post"]
            # (row 0 is overwritten by prepare_for_next_iter's 1-token draft forward, :279)
            walked = [sl - (gt - 1) for sl in res["slots"]]
            for t in walked:
                if len(succ[t]):
                    assert np.array_equal(draft[t], post[t]), f"{name} step {s} node {t}"
        if f"step{s}/residual" in z:
            a16 = res["final_p"].view(np.int16).astype(np.int32)
            b16 = z[f"step{s}/residual"].view(np.int16).astype(np.int32)
            assert np.abs(a16 - b16).max() <= 2


@pytest.mark.parametrize("name", COMPACT_STOCHASTIC)
def test_verify_stochastic_matches_reference_full_vocab(name):
    """V = 32000: the oracle's verifier on the reference's own logits.  The trace keeps the full target / draft rows
    of the nodes the reference walked (the verifier touches no other row) and every step must reproduce the
    reference's accept length, committed tokens and bonus token; rejections (residual updates) must occur."""
    margins = check_compact_verify(*load_trace(name), name)
    assert sum(m <= 0 for m in margins) >= 3 and sum(m > 0 for m in margins) >= 3


@pytest.mark.parametrize("name", TOPP_TRACES)
def test_top_p_filter_on_the_rows_of_a_reference_trace(name):
    """SpecTree at top_p = 0.9 (the harness default): the reference filtered the target rows in place; the trace keeps the
    rows of the walked path raw and filtered.  The oracle's filter on the raw rows == the reference's filtered rows (up to
    the identity of exactly equal logits at the cut), and -- test_verify_stochastic_matches_reference_full_vocab -- the
    oracle's verifier on the filtered rows walks the reference's path in every step."""
    from helpers import assert_top_p_equal_up_to_ties
    z, meta = load_trace(name)
    assert meta["top_p"] < 1.0
    rows = 0
    for s in range(int(z["n_steps"])):
        raw, filt = z[f"step{s}/path_target_rows_raw"], z[f"step{s}/path_target_rows"]
        assert np.isinf(filt).any() and not np.isinf(raw).any()
        assert_top_p_equal_up_to_ties(raw, O.top_p_filter(raw, meta["top_p"], meta["T"]), filt, f"{name} step {s}")
        rows += raw.shape[0]
    assert rows >= 20


def check_compact_verify(z, meta, name):
    """(also applied to fresh V = 32000 traces of the live reference: tests/test_oracle_live_reference_cpu.py)"""
    succ = meta["successors"]
    n, V = len(succ), meta["vocab"]
    margins = []
    for s in range(int(z["n_steps"])):
        gt = int(z[f"step{s}/gt"])
        nodes = z[f"step{s}/path_nodes"]
        tl = np.zeros((n, V), dtype=np.float16)
        dl = np.zeros((n, V), dtype=np.float16)
        tl[nodes] = z[f"step{s}/path_target_rows"]
        dl[nodes] = z[f"step{s}/path_draft_rows"]
        tokens = z[f"step{s}/tokens_pre"].copy()
        res = O.verify_stochastic(tl, dl, tokens, z["r"], succ, gt, meta["T"], int(z["bonus_u24"][s]), margins=margins)
        assert [0] + [sl - (gt - 1) for sl in res["slots"]] == list(nodes), f"{name} step {s}: walked path differs"
        assert res["accept_len"] == int(z[f"step{s}/accept_len"]), f"{name} step {s}"
        valid = z[f"step{s}/valid_tokens"]
        assert np.array_equal(tokens[:valid.shape[0]], valid), f"{name} step {s}"
        if f"step{s}/residual" in z:
            a16 = res["final_p"].view(np.int16).astype(np.int32)
            b16 = z[f"step{s}/residual"].view(np.int16).astype(np.int32)
            assert np.abs(a16 - b16).max() <= 2
    return margins


@pytest.mark.parametrize("name", COMPACT_STOCHASTIC)
def test_sampler_matches_reference_full_vocab(name):
    """V = 32000: sampling without replacement of every tree level of step 0 against the reference's outputs.  The
    level's input rows are the trace's draft rows where they were kept (root row); the noise is regenerated from the
    recorded seed (same CPU-generator draws as Tree/SpecTree.py:60,84) and checked on the recorded probe columns."""
    assert check_compact_sampler(*load_trace(name), name) >= 3


def check_compact_sampler(z, meta, name):
    import torch
    n, V, M = len(meta["successors"]), meta["vocab"], meta["M"]
    torch.manual_seed(meta["seed"] + 7)
    r = torch.rand(M, dtype=torch.float16).numpy()
    rand = torch.empty((n, V), dtype=torch.float16).uniform_().numpy()
    assert np.array_equal(r, z["r"]) and np.array_equal(rand[:, ::meta["compact"]], z["rand_probe"])
    from sequoia_amd.growmap import GrowMap
    g = GrowMap.from_successors(meta["successors"])
    checked = 0
    for s in range(int(z["n_steps"])):
        row0 = z[f"step{s}/path_draft_rows"][0]                  # the root's draft row before any -65504 write
        k = g.levels[0].k
        got = O.sample_wor(row0[None], rand[0:1], k, meta["T"])[0]
        want = z[f"step{s}/samp0/out"].reshape(-1)[:k]
        keys = O.sample_keys(row0[None], rand[0:1], meta["T"])[0]
        for c in range(k):
            if got[c] != want[c]:
                assert keys[got[c]] == keys[want[c]], f"{name} step {s} rank {c}"
        checked += 1
    return checked


def test_verify_greedy_matches_reference_headline_dims():
    """C_7b (GreedyTree, 8x8 growmap, 68m -> 7B dims, V = 32000): the oracle's greedy walk on the reference's own target
    rows of the walked nodes (the only rows it reads) reproduces the accepted tokens of every step."""
    z, meta = load_trace("C_7b")
    succ = meta["successors"]
    n, V = len(succ), meta["vocab"]
    accepted = 0
    for s in range(int(z["n_steps"])):
        gt = int(z[f"step{s}/gt"])
        nodes = z[f"step{s}/path_nodes"]
        tl = np.full((n, V), -60000.0, dtype=np.float16)
        tl[nodes] = z[f"step{s}/path_target_rows"]
        tokens = z[f"step{s}/tokens_pre"].copy()
        # rows off the walked path are never compared by the walk: give them an argmax no child carries
        tl[[t for t in range(n) if t not in set(nodes.tolist())], 1] = 1.0
        res = O.verify_greedy(tl, tokens, succ, gt)
        assert res["accept_len"] == int(z[f"step{s}/accept_len"]), f"step {s}"
        valid = z[f"step{s}/valid_tokens"]
        assert np.array_equal(tokens[:valid.shape[0]], valid), f"step {s}"
        accepted += res["n_tree"]
    assert accepted >= 1


@pytest.mark.parametrize("name", ["C_greedy8x8"] + LARGE_GREEDY)
def test_verify_greedy_matches_reference(name):
    z, meta = load_trace(name)
    succ = meta["successors"]
    for s in range(int(z["n_steps"])):
        gt = int(z[f"step{s}/gt"])
        tokens = z[f"step{s}/tokens_pre"].copy()
        res = O.verify_greedy(z[f"step{s}/target_logits"], tokens, succ, gt)
        assert res["accept_len"] == int(z[f"step{s}/accept_len"])
        valid = z[f"step{s}/valid_tokens"]
        assert np.array_equal(tokens[:valid.shape[0]], valid)


def test_kv_compact_matches_reference_semantics():
    rng = np.random.RandomState(0)
    L, H, M, D = 2, 3, 40, 8
    k = rng.randn(L, H, M, D).astype(np.float16)
    v = rng.randn(L, H, M, D).astype(np.float16)
    k0, v0 = k.copy(), v.copy()
    slots, off = [13, 14, 19], 12
    O.kv_compact(k, v, slots, off, M)
    import torch
    tk, tv = torch.from_numpy(k0.copy()), torch.from_numpy(v0.copy())
    # Engine/Llama_KV.py:60-68 verbatim semantics expressed with torch indexing
    tk[..., off:off + 3, :] = tk[..., slots, :]
    tv[..., off:off + 3, :] = tv[..., slots, :]
    tk[..., off + 3:, :] = 0.0
    tv[..., off + 3:, :] = 0.0
    assert np.array_equal(k, tk.numpy()) and np.array_equal(v, tv.numpy())


def test_inverse_cdf_is_exact_and_distribution_correct():
    p = np.array([0.25, 0.0, 0.5, 0.25], dtype=np.float16)
    counts = np.zeros(4)
    for u in range(0, 1 << 24, 4099):
        counts[O.inverse_cdf(p, u)] += 1
    frac = counts / counts.sum()
    assert frac[1] == 0
    assert np.allclose(frac, [0.25, 0, 0.5, 0.25], atol=1e-3)
    assert O.inverse_cdf(p, 0) == 0 and O.inverse_cdf(p, (1 << 24) - 1) == 3


def test_top_p_filter_matches_reference():
    """utils.get_sampling_logits (utils.py:65-77) on V = 32000 rows: the oracle removes exactly the reference's tokens,
    except for WHICH of several exactly-equal logits sit before the cut (torch's CPU sort of fp16 is not stable).  The
    cumulative sums themselves are reproduced exactly: fp16 probabilities are multiples of 2^-24, so torch's sequential
    fp32 cumsum is exact below 1 and equals the integer-grid sum."""
    from helpers import assert_top_p_equal_up_to_ties
    z = np.load(f"{GOLDEN}/rows_v32000.npz")
    exact = 0
    for i in range(4):
        for tp, key in ((0.9, "topp09"), (0.5, "topp05")):
            got = O.top_p_filter(z[f"wor{i}/logits"], tp, 0.6)
            want = z[f"wor{i}/{key}"]
            exact += assert_top_p_equal_up_to_ties(z[f"wor{i}/logits"], got, want, f"wor{i} P={tp}") == 0
            # the kept mass straddles top_p like the reference's
            p = O.scaled_softmax_f16(z[f"wor{i}/logits"], 0.6).astype(np.float64)
            assert (np.where(np.isinf(got), 0, p).sum(1) >= min(tp, p.max(1).min()) - 2e-3).all()
    assert exact >= 5          # most rows have no tie at the cut and match token for token (5 of 8 on the committed rows)
