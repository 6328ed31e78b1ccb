"""Host logic on CPU: the framework's engines + trees (sequoia_amd.Engine / sequoia_amd.Tree)
driven with the numpy-oracle ops adapter must reproduce the *reference's own* speculation
traces (tests/golden/trace_*.npz: same weights, prompt, noise, bonus uniforms).

This pins the index algebra, KV protocol, growmap handling and step bookkeeping of the host
code against the reference; the HIP kernels themselves are checked on the GPU
(tests/test_hip_kernels.py, tests/test_e2e_gpu.py)."""
import numpy as np
import pytest
import torch

from conftest import LARGE_TRACES, TOPP_TRACES, TRACE_NAMES
from helpers import check_replay, replay_trace


@pytest.fixture()
def oracle_ops():
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    ops.set_ops_for_testing(OracleOps())
    yield
    ops.set_ops_for_testing(None)


# (+ V = 32000 under the harness's default top_p = 0.9: 30 s; + the reference's 193- / 256- / 512-node growmaps)
@pytest.mark.parametrize("name", TRACE_NAMES + TOPP_TRACES + LARGE_TRACES)
def test_native_loop_reproduces_reference_trace(oracle_ops, name):
    steps, tree, draft, target, z, meta = replay_trace(name, "cpu")
    assert len(steps) == int(z["n_steps"])
    matched, diverged = check_replay(steps, z, meta)
    # on CPU every committed trace reproduces the reference's accepted tokens in every step
    assert diverged is None and matched == len(steps), f"{name}: diverged at step {diverged}"
    # KV protocol: offsets after the last step equal the reference's
    last = len(steps) - 1
    assert draft.engine.kv_cache.kv_offset == int(z[f"step{last}/kv_draft"][2])
    assert target.engine.kv_cache.kv_offset == int(z[f"step{last}/kv_target"][2])
    assert np.array_equal(tree.position_ids.numpy(), z[f"step{last}/position_ids_post"])


def test_kv_cache_bytes_match_reference_with_full_zero_policy(oracle_ops, monkeypatch):
    """With the reference's zero policy the target cache is identical up to attention rounding;
    the *structure* (which slots are live, which are zero) is identical exactly."""
    from sequoia_amd.Engine.Llama_KV import KV_Cache
    monkeypatch.setattr(KV_Cache, "ZERO_POLICY", "full")
    steps, tree, draft, target, z, meta = replay_trace("B_seq128", "cpu")
    ref_k = z["final/target_k"]
    got_k = target.engine.kv_cache.k_cache.numpy()
    assert got_k.shape == ref_k.shape
    assert np.array_equal(np.abs(got_k).sum(-1) == 0, np.abs(ref_k).sum(-1) == 0)
    live = np.abs(ref_k).sum(-1) != 0
    # layer 0 keys depend only on embeddings + RoPE (no attention upstream): bit-exact
    assert np.array_equal(got_k[0][live[0]], ref_k[0][live[0]])
    assert np.abs(got_k.astype(np.float32) - ref_k.astype(np.float32)).max() < 5e-2


def test_foreign_sampling_callable_is_honoured(oracle_ops):
    """A caller-supplied sampler (the reference's injection seam, tests/testbed.py:269-285) is used
    verbatim when it is not one of ours."""
    from conftest import load_trace
    from helpers import build_engines, make_tree
    z, meta = load_trace("demo4")
    draft, target = build_engines(z, meta, "cpu")
    tree = make_tree(z, meta, draft, target, "cpu")
    calls = []

    def fake(logits, rand):
        calls.append(logits.shape)
        return torch.full((logits.shape[0],), 7, dtype=torch.long)
    tree.sampling_callables = {i: fake for i in range(tree.draft_step - 1)}
    tree.sample_gather_indices = {i: torch.zeros(1, dtype=torch.long) for i in range(tree.draft_step - 1)}
    gt = tree.ground_truth_len
    tree.construct_grow_map()
    assert len(calls) == tree.draft_step - 1
    assert (tree.tokens[gt:gt + tree.tree_size - 1] == 7).all()


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the HIP ops object rejects host tensors loudly."""
    from sequoia_amd import native, ops
    ops.set_ops_for_testing(None)
    o = ops.get_ops()
    assert o.name == "hip"
    x = torch.zeros(4, 8, dtype=torch.float16)
    with pytest.raises(native.SequoiaNativeError):
        o.rmsnorm(x, torch.ones(8, dtype=torch.float16), torch.empty_like(x), 1e-6)


def test_last_step_before_max_length_completes_and_the_next_is_refused(oracle_ops):
    """A prompt that leaves room for exactly one tree: verify() must finish the step (the caller's loop ends on
    its own length test, tests/testbed.py:80); only a further speculation step is an error (README.md:47)."""
    from conftest import load_trace
    from helpers import build_engines, make_tree
    z, meta = load_trace("B_seq128")
    draft, target = build_engines(z, meta, "cpu")
    n = len(meta["successors"])
    M = meta["M"]
    import numpy as np
    z2 = dict(z)
    rng = np.random.default_rng(0)
    z2["prompt"] = rng.integers(3, meta["vocab"], M - n + 1).astype(np.int64)

    class Z(dict):
        files = list(z.files)
    zz = Z(z2)
    tree = make_tree(zz, meta, draft, target, "cpu")
    tree.construct_grow_map()
    valid, a, _, terminal = tree.verify()
    assert valid.shape[0] >= M - n + 1 + (0 if terminal else 1)
    if not terminal:
        assert tree._no_room is not None
        with pytest.raises(ValueError):
            tree.construct_grow_map()


def test_loop_run_prompts_counts_whole_prompts_and_prefill_steps():
    """harness.Loop.run_prompts (bench.py's `value_reference_metric`): whole prompts from the prefill-bearing first step to
    max_new tokens, as tests/testbed.py:78-95 times them; the prefill steps and their wall time are accounted separately."""
    from conftest import load_trace
    from helpers import build_engines
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    from sequoia_amd.growmap import GrowMap
    from sequoia_amd.harness import Loop
    ops.set_ops_for_testing(OracleOps())
    try:
        z, meta = load_trace("demo4")
        draft, target = build_engines(z, meta, "cpu")
        gm = GrowMap.from_successors(meta["successors"])
        cfg = dict(mode="stochastic", M=meta["M"])
        prompts = [[int(t) for t in z["prompt"]], [int(t) for t in z["prompt"][::-1]]]
        loop = Loop(cfg, draft, target, gm, "cpu", prompts, use_graphs=False, max_new=len(prompts[0]) + 12, vocab=meta["vocab"])
        secs, toks, steps = loop.run_steps(2)                # leaves a prompt half-way
        assert steps == 2 and loop.prefill_steps == 1 and loop.prompts_done == 0
        secs, toks, steps = loop.run_prompts(2)
        assert loop.prompts_done == 2 and loop.prefill_steps == 3 and loop.tree is None
        assert steps >= 2 and toks >= 2 * 12 and 0 < loop.prefill_seconds <= secs + 1.0
        # bench.py's timed window: start_fresh_prompt() drops a half-done prompt, so the next K steps begin with a
        # prefill-bearing step whose time and tokens are accounted apart (`value` vs `value_steady`)
        loop.run_steps(1)
        assert loop.tree is not None
        loop.start_fresh_prompt()
        assert loop.tree is None
        p0, s0, t0 = loop.prefill_steps, loop.prefill_seconds, loop.prefill_tokens
        secs, toks, steps = loop.run_steps(3)
        assert steps == 3 and loop.prefill_steps - p0 == 1 and loop.prefill_seconds > s0
        assert 1 <= loop.prefill_tokens - t0 <= toks
    finally:
        ops.set_ops_for_testing(None)


def test_collective_fault_word_stops_the_loop():
    """Engine/xgmi_allreduce.py: a collective kernel whose bounded spin ran out ORs its status bits into a word of pinned
    host memory; raise_on_fault() -- called by verify() / collect_step() at every step -- turns that into an exception
    instead of letting the job decode on stale partial sums.  Host logic only: the word is set by hand."""
    import weakref
    from sequoia_amd.Engine import xgmi_allreduce as XA

    class Fake(XA.XgmiAllReduce):
        def __init__(self):                       # no workspace, no device: only the fault word and the registry entry
            self.rank, self.fault = 3, torch.zeros(16, dtype=torch.int32)
            XA._LIVE.append(weakref.ref(self))

    ar = Fake()
    try:
        XA.raise_on_fault()                       # clean word: nothing happens
        ar.fault[0] = 1 | 8
        with pytest.raises(XA.XgmiCollectiveTimeout) as e:
            XA.raise_on_fault()
        assert "rank 3" in str(e.value) and "phase 1" in str(e.value) and "'read' flag" in str(e.value)
        ar.clear_fault()
        XA.raise_on_fault()
    finally:
        XA._LIVE[:] = [r for r in XA._LIVE if r() is not None and r() is not ar]
    del ar
    XA.raise_on_fault()                           # dead entries are pruned, not dereferenced
