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


@pytest.mark.parametrize("name", ["demo4", "B_seq128", "D_160m13b", "C_greedy8x8", "V32k_seq128", "B_topp09", "B_7b", "C_7b", "D_13b_w4",
                                  "E_70b_w2", "D_13b", "E_70b_w8", "L_8x24", "L_8x24_greedy", "L_S256", "L_S512", "L_S256_v32k"])
def test_step_graph_equals_synchronous_steps(name):
    """(L_*: the reference's 193- / 256- / 512-node growmaps.  Their verify forward has more than 144 rows, so inside the
    captured step it runs on the general path -- sq_stage_tree_inputs as a launch of its own, hipBLASLt projections captured
    in the graph -- while the draft levels (<= 116 rows) stay on the tall-skinny kernel where the model qualifies.)"""
    z, meta = load_trace(name)
    n_steps = int(z["n_steps"])
    want = _sync_run(z, meta, n_steps)
    draft, target = build_engines(z, meta, DEV)
    tree = make_tree(z, meta, draft, target, DEV, step_graph=True)
    assert tree.state is not None and tree.state.graph is not None
    assert tree.state.fwd_target.tree.stage is not None        # staging inside the forwards' first launch (default)
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


@pytest.mark.parametrize("rows,hidden,frag,advance", [(34, 768, True, False), (1, 768, True, True), (128, 4096, True, False),
                                                       (19, 1024, False, False), (129, 8192, True, False)])
def test_embed_stage_rmsnorm_equals_stage_then_embed(rows, hidden, frag, advance):
    """sq_embed_stage_rmsnorm_f16 (the first launch of a forward in the device-driven step) == sq_stage_tree_inputs followed by
    sq_embed_rmsnorm_f16: the same staged ids / positions / slots / context, step block and rows, bit for bit."""
    from sequoia_amd.native import SQ_STEP_GT, SQ_STEP_INDEX, SQ_STEP_INTS, SQ_STEP_NEXT_GT
    from sequoia_amd.ops import HipOps
    hip = HipOps()
    g = torch.Generator().manual_seed(rows * 31 + hidden)
    V, n_tree, M = 5000, 140, 600
    embed = (torch.randn(V, hidden, generator=g) * 0.5).half().to(DEV)
    w = (1.0 + 0.1 * torch.randn(hidden, generator=g)).half().to(DEV)
    tokens = torch.randint(0, V, (M,), generator=g).to(DEV)
    tokens[260] = V + 7                                              # out-of-range id: clamped like sq_embed_rmsnorm_f16
    depth = torch.randint(0, 9, (n_tree,), generator=g, dtype=torch.int32).to(DEV)
    rel_slot0, rel_kv = (-1, 0) if rows == 1 else (3, 3 + rows)
    outs = []
    for fused in (False, True):
        step = torch.zeros(SQ_STEP_INTS, dtype=torch.int32, device=DEV)
        step[SQ_STEP_GT], step[SQ_STEP_NEXT_GT], step[SQ_STEP_INDEX] = 250, 257, 11
        ids = torch.full((rows,), -5, dtype=torch.long, device=DEV)
        pos, sto = ids.clone(), ids.clone()
        ctx = torch.zeros(3, dtype=torch.int32, device=DEV)
        x = torch.zeros(rows, hidden, dtype=torch.float16, device=DEV)
        h = torch.zeros(hip.frag_shape(rows, hidden) if frag else (rows, hidden), dtype=torch.float16, device=DEV)
        stage = (ids, pos, sto, ctx, tokens, depth, n_tree, rel_slot0, rel_kv, step, advance)
        if fused:
            hip.embed_stage_rmsnorm(stage, embed, w, x, h, 1e-5, out_frag=frag)
        else:
            hip.stage_tree_inputs(*stage)
            hip.embed_rmsnorm(ids, embed, w, x, h, 1e-5, out_frag=frag)
        torch.cuda.synchronize()
        outs.append([t.cpu() for t in (ids, pos, sto, ctx, step, x, h)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    gt = 257 if advance else 250
    assert int(outs[1][4][SQ_STEP_GT]) == gt and int(outs[1][4][SQ_STEP_INDEX]) == (12 if advance else 11)
    assert outs[1][2].tolist() == list(range(gt + rel_slot0, gt + rel_slot0 + rows))


def test_step_graph_with_separate_staging_launches(monkeypatch):
    """SEQUOIA_FUSE_STAGE=0: sq_stage_tree_inputs in front of every forward instead of the staging inside the forward's first
    launch (the default, which test_step_graph_equals_synchronous_steps covers on every trace) -- same committed tokens."""
    from sequoia_amd.Tree import step_graph
    monkeypatch.setattr(step_graph, "FUSE_STAGE", False)
    z, meta = load_trace("B_seq128")
    n_steps = int(z["n_steps"])
    want = _sync_run(z, meta, n_steps)
    draft, target = build_engines(z, meta, DEV)
    tree = make_tree(z, meta, draft, target, DEV, step_graph=True)
    assert tree.state is not None and tree.state.graph is not None and tree.state.fwd_target.tree.stage is None
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
        assert a == want[s][0] and bonus == int(want[s][1][-1]), f"step {s}"
        s += 1
    tree.end_pipeline()
    torch.cuda.synchronize()
    assert np.array_equal(tree.tokens[:want[-1][0] + 1].cpu().numpy(), want[-1][1])


def test_step_graph_with_chunked_verify_in_exclusive_mode(monkeypatch):
    """A target whose fragment-major images are its only weights (SEQUOIA_TS_EXCLUSIVE=1: the 70B-on-few-GPUs mode) has no
    hipBLASLt path: a verify forward of more than 144 rows (the 256-node growmap) runs as consecutive <= 144-row chunks of the
    tall-skinny kernel, each reading the captured forward's context block shifted by its rows.  Whole-step graphs must commit
    the tokens of the synchronous steps of the ordinary (two-copy) engines -- which replay the reference's trace."""
    name = "L_S256_v32k"
    z, meta = load_trace(name)
    n_steps = int(z["n_steps"])
    want = _sync_run(z, meta, n_steps)
    monkeypatch.setenv("SEQUOIA_TS_EXCLUSIVE", "1")
    draft, target = build_engines(z, meta, DEV)
    assert target.engine.model.ts is not None and target.engine.model.ts.exclusive
    tree = make_tree(z, meta, draft, target, DEV, step_graph=True)
    assert tree.state is not None and tree.state.graph is not None
    tree.construct_grow_map()
    valid, a, _, term = tree.verify()
    assert int(a) == want[0][0] and np.array_equal(valid.cpu().numpy(), want[0][1])
    tree.begin_pipeline()
    s = 1
    while s < n_steps:
        while (tree.can_enqueue(meta["M"]) and len(tree._pipe["inflight"]) < 2 and s + len(tree._pipe["inflight"]) < n_steps):
            tree.enqueue_step()
        a, n_acc, bonus, term = tree.collect_step()
        assert a == want[s][0] and bonus == int(want[s][1][-1]), f"{name} step {s}"
        s += 1
    tree.end_pipeline()
    torch.cuda.synchronize()
    assert np.array_equal(tree.tokens[:want[-1][0] + 1].cpu().numpy(), want[-1][1])
