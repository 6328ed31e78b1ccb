"""Device-driven speculation step, host logic on CPU: the pipelined API (begin_pipeline / enqueue_step /
collect_step, Tree/step_graph.py) driven with the oracle ops adapter must commit exactly the tokens of the
synchronous reference API (construct_grow_map + verify) on the reference's own traces."""
import numpy as np
import pytest

from conftest import load_trace
from helpers import build_engines, make_tree


@pytest.fixture()
def oracle_ops():
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    ops.set_ops_for_testing(OracleOps())
    yield
    ops.set_ops_for_testing(None)


@pytest.mark.parametrize("name", ["demo4", "D_160m13b", "C_greedy8x8"])
def test_pipelined_steps_equal_synchronous_steps(oracle_ops, name):
    z, meta = load_trace(name)
    n_steps = int(z["n_steps"])
    # synchronous run
    draft, target = build_engines(z, meta, "cpu")
    tree = make_tree(z, meta, draft, target, "cpu")
    want = []
    for s in range(n_steps):
        tree.construct_grow_map()
        valid, a, _, term = tree.verify()
        want.append((int(a), valid.numpy().copy()))
    assert [w[0] for w in want] == [int(z[f"step{s}/accept_len"]) for s in range(n_steps)]
    # pipelined run: step 0 synchronous (it carries the target prefill), then the device-driven loop
    draft, target = build_engines(z, meta, "cpu")
    tree = make_tree(z, meta, draft, target, "cpu", step_graph=True)
    assert tree.state is not None
    tree.construct_grow_map()
    valid, a, _, term = tree.verify()
    assert int(a) == want[0][0] and np.array_equal(valid.numpy(), want[0][1])
    tree.begin_pipeline()
    got = []
    horizon = meta["M"]
    s = 1
    while s < n_steps:
        while tree.can_enqueue(horizon) and len(tree._pipe["inflight"]) < 2 and s + len(tree._pipe["inflight"]) < n_steps:
            tree.enqueue_step()
        a, n_acc, bonus, term = tree.collect_step()
        got.append(a)
        assert a == want[s][0], f"step {s}"
        assert bonus == int(want[s][1][-1])
        s += 1
    tree.end_pipeline()
    assert np.array_equal(tree.tokens[:want[-1][0] + 1].numpy(), want[-1][1])
    # the synchronous API keeps working after a pipelined stretch
    assert tree.ground_truth_len == want[-1][0] + 1
    assert draft.engine.kv_cache.kv_offset == tree.ground_truth_len
    assert target.engine.kv_cache.kv_offset == tree.ground_truth_len - 1
    if tree.ground_truth_len + tree.tree_size - 1 <= meta["M"]:
        tree.construct_grow_map()
        tree.verify()


@pytest.mark.parametrize("name", ["C_greedy8x8", "D_160m13b"])
def test_eos_inside_the_pipeline_keeps_the_finished_text(oracle_ops, name):
    """A prompt that ends on an accepted EOS while another step is already in flight: the step behind the
    terminal one must commit nothing -- tree.tokens[:a], the host mirrors and the KV offsets are those of the
    synchronous run."""
    from helpers import build_renamed, find_eos_case, pipelined_run
    z, meta = load_trace(name)
    x, want = find_eos_case(z, meta, "cpu")
    assert x is not None, "no token of this trace ends the prompt at step >= 2 when renamed to EOS"
    a_end = want[-1][0]
    draft, target, tree = build_renamed(z, meta, "cpu", x, step_graph=True)
    assert tree.state is not None
    got, behind = pipelined_run(tree, max_steps=len(want) + 3)
    assert [g[0] for g in got] == [w[0] for w in want] and got[-1][1]
    assert behind >= 1, "the terminal step was collected with nothing in flight behind it: the case is not exercised"
    assert np.array_equal(tree.tokens[:a_end].cpu().numpy(), want[-1][1])
    assert tree.ground_truth_len == a_end
    assert draft.engine.kv_cache.kv_offset == a_end and target.engine.kv_cache.kv_offset == a_end - 1
