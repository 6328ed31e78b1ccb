"""Device-driven speculation step on the GPU: the whole step (draft expansion, target forward, verification, KV
compaction, next root) replayed as ONE hipGraph with the ground-truth length on the device must commit exactly the
tokens of the synchronous reference API, on the reference's own traces and on the config-B architecture pair."""
import numpy as np
import pytest
import torch

from conftest import load_trace
from helpers import build_engines, make_tree

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sync_run(z, meta, n_steps):
    draft, target = build_engines(z, meta, DEV)
    tree = make_tree(z, meta, draft, target, DEV)
    out = []
    for s in range(n_steps):
        tree.construct_grow_map()
        valid, a, _, term = tree.verify()
        out.append((int(a), valid.cpu().numpy().copy()))
    return out


@pytest.mark.parametrize("name", ["demo4", "B_seq128", "D_160m13b", "C_greedy8x8", "V32k_seq128", "B_7b", "C_7b", "D_13b_w4", "E_70b_w2"])
def test_step_graph_equals_synchronous_steps(name):
    z, meta = load_trace(name)
    n_steps = int(z["n_steps"])
    want = _sync_run(z, meta, n_steps)
    draft, target = build_engines(z, meta, DEV)
    tree = make_tree(z, meta, draft, target, DEV, step_graph=True)
    assert tree.state is not None and tree.state.graph is not None
    tree.construct_grow_map()
    valid, a, _, term = tree.verify()
    assert int(a) == want[0][0] and np.array_equal(valid.cpu().numpy(), want[0][1])
    tree.begin_pipeline()
    s = 1
    while s < n_steps:
        while (tree.can_enqueue(meta["M"]) and len(tree._pipe["inflight"]) < 2
               and s + len(tree._pipe["inflight"]) < n_steps):
            tree.enqueue_step()
        a, n_acc, bonus, term = tree.collect_step()
        assert a == want[s][0] and bonus == int(want[s][1][-1]), f"{name} step {s}"
        s += 1
    tree.end_pipeline()
    torch.cuda.synchronize()
    assert np.array_equal(tree.tokens[:want[-1][0] + 1].cpu().numpy(), want[-1][1])
    # a second prompt on the same engines adopts the same static buffers and the same graph
    draft.clear_kv(); target.clear_kv()
    tree2 = make_tree(z, meta, draft, target, DEV, step_graph=True)
    assert tree2.state is tree.state
    tree2.construct_grow_map()
    valid, a, _, term = tree2.verify()
    assert int(a) == want[0][0] and np.array_equal(valid.cpu().numpy(), want[0][1])


def test_pipelined_loop_matches_synchronous_loop_on_config_b():
    """harness.Loop in pipelined mode == synchronous mode: same accepted tokens per step over two prompts of the
    68m -> 7B architecture pair (random-init: short accepted paths, long run of steps)."""
    from sequoia_amd.harness import MODELS, Loop, build, load_prompts
    cfg = dict(MODELS["B"])
    draft, target, gm = build(cfg, DEV, "calibrated")
    prompts = load_prompts()[:2]
    seqs = []
    for pipelined in (False, True):
        torch.manual_seed(123)
        draft.clear_kv(); target.clear_kv()
        loop = Loop(cfg, draft, target, gm, DEV, prompts, pipelined=pipelined)
        lens = []
        loop.run_steps(40, on_accept=lambda a: lens.append(a))
        seqs.append(lens)
    assert seqs[0] == seqs[1] and len(seqs[0]) == 40


@pytest.mark.parametrize("name", ["C_greedy8x8", "D_160m13b", "B_seq128"])
def test_eos_inside_the_pipeline_keeps_the_finished_text(name):
    """A prompt ends on an accepted EOS while the next whole-step graph is already in flight: the walker of that step
    sees SQ_STEP_ACTIVE == 0 and commits nothing, the host drops its record -- tokens[:a], the host mirrors and the KV
    offsets equal the synchronous run's (the overshoot step used to overwrite the committed text)."""
    from helpers import build_renamed, find_eos_case, pipelined_run
    z, meta = load_trace(name)
    x, want = find_eos_case(z, meta, DEV)
    assert x is not None, "no token of this trace ends the prompt at step >= 2 when renamed to EOS"
    a_end = want[-1][0]
    draft, target, tree = build_renamed(z, meta, DEV, x, step_graph=True)
    assert tree.state is not None and tree.state.graph is not None
    got, behind = pipelined_run(tree, max_steps=len(want) + 3)
    torch.cuda.synchronize()
    assert [g[0] for g in got] == [w[0] for w in want] and got[-1][1]
    assert behind >= 1, "nothing was in flight behind the terminal step: the case is not exercised"
    assert np.array_equal(tree.tokens[:a_end].cpu().numpy(), want[-1][1])
    assert tree.ground_truth_len == a_end
    assert draft.engine.kv_cache.kv_offset == a_end and target.engine.kv_cache.kv_offset == a_end - 1
    # the committed KV rows survive too: the target cache of the pipelined run equals the synchronous run's on [0, a - 1)
    d2, t2, tree2 = build_renamed(z, meta, DEV, x)
    from helpers import sync_run
    sync_run(tree2, len(want))
    # (GreedyTree's synchronous API leaves the KV rows of a terminal step uncompacted, Tree/GreedyTree.py:206-209; the
    # device-driven step always compacts: compare the rows both runs define -- everything before the terminal step's gt)
    upto = a_end - 1 if tree._compact_when_terminal else want[-2][0]
    ka, kb = target.engine.kv_cache.k_cache[..., :upto, :], t2.engine.kv_cache.k_cache[..., :upto, :]
    assert torch.equal(ka, kb)
