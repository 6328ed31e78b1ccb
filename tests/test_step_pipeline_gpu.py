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


@pytest.mark.parametrize("name", ["demo4", "B_seq128", "D_160m13b", "C_greedy8x8", "V32k_seq128"])
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
