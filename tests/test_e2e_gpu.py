"""GPU end-to-end parity: the native loop (HIP kernels + PyTorch-ROCm GEMMs) replays the
reference's recorded speculation traces — same weights, prompt, noise and bonus uniforms —
and must emit the reference's accepted token sequence.

Greedy (GreedyTree): tokens are integer work -> bit-exact.  Stochastic (SpecTree): logits on
the GPU differ from the reference's CPU run by <= a few fp16 ulps (GEMM / attention
accumulation order), so a sampled or accepted token may legitimately differ only where the
decision margin is inside that tolerance; every trace committed here reproduces exactly.
"""
import numpy as np
import pytest
import torch

from conftest import (COMPACT_TRACES, DEPTH_TRACES, HEADLINE_TRACES, LARGE_COMPACT, LARGE_TRACES, TOPP_TRACES, TRACE_NAMES, WIDTH_TRACES,
                      load_trace)
from helpers import assert_replay_complete, build_engines, check_replay, make_tree, replay_trace

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", TRACE_NAMES + COMPACT_TRACES + TOPP_TRACES + HEADLINE_TRACES + WIDTH_TRACES + DEPTH_TRACES + LARGE_TRACES
                         + LARGE_COMPACT)
def test_gpu_loop_reproduces_reference_tokens(name):
    """(round 5: + the reference's 193- / 256- / 512-node growmaps -- L_*: 4 and 8 ancestor-bitmask words in the attention
    kernel, the verifier at n = 512, sampler levels of up to 116 rows x 32 children, verify forwards of 193-512 rows, which
    leave the <= 144-row tall-skinny path for the hipBLASLt forward)
    All steps of every trace (configs A-E shapes, the demo tree, the V = 32000 trace, the same pair under the harness's
    default nucleus filter top_p = 0.9 -- sq_top_p_filter_f16 in front of the verifier --, and the two traces at the
    headline model dims: 68m -> Llama-2-7b architectures, SpecTree 128-node growmap and GreedyTree 8x8).  Logits agree within
    tolerance in every compared step (asserted inside check_replay); the committed tokens are identical in every
    step -- a stochastic run may leave the reference only at a decision whose margin is proven to be inside one fp16
    ulp (assert_replay_complete)."""
    steps, tree, draft, target, z, meta = replay_trace(name, DEV)
    matched, diverged = check_replay(steps, z, meta)
    assert_replay_complete(name, steps, tree, z, meta, matched, diverged)
    print(f"{name}: {matched}/{int(z['n_steps'])} steps token-identical to the reference"
          + ("" if diverged is None else f" (margin-limited decision at step {diverged})"))


def test_logits_close_to_reference_forward():
    """Target logits of the first verify call vs the reference's recorded ones (fp16, logit
    tolerance 3e-2 absolute on logits of magnitude ~8: a few fp16 ulps)."""
    z, meta = load_trace("B_seq128")
    draft, target = build_engines(z, meta, DEV)
    tree = make_tree(z, meta, draft, target, DEV)
    got0 = tree.draft_logits[0].float().cpu().numpy()
    assert np.abs(got0 - z["draft_logits0_prefill"].astype(np.float32)).max() < 3e-2
    tree.construct_grow_map()
    same = (tree.tokens.cpu().numpy()[:tree.num_nodes] == z["step0/tokens_pre"][:tree.num_nodes])
    tree.verify()
    got = tree.target_logits.float().cpu().numpy()
    ref = z["step0/target_logits"].astype(np.float32)
    # rows whose token path matches the reference's are comparable
    succ = meta["successors"]
    gt = int(z["step0/gt"])
    ok_rows = [0]
    parent = {c: p for p, ch in enumerate(succ) for c in ch}
    for t in range(1, len(succ)):
        if parent[t] in ok_rows and same[t + gt - 1]:
            ok_rows.append(t)
    assert len(ok_rows) > len(succ) // 2
    assert np.abs(got[ok_rows] - ref[ok_rows]).max() < 3e-2


def test_graphs_match_eager():
    """hipGraph replay (device-resident {q_slot0, gt, kv_len}) == eager launches, bit for bit."""
    z, meta = load_trace("B_seq128")
    outs = []
    for use_graph in (False, True):
        draft, target = build_engines(z, meta, DEV)
        tree = make_tree(z, meta, draft, target, DEV)
        if use_graph:
            lens = sorted({lv.total for lv in tree.gm.levels} | {1})
            draft.initialize_cuda_graph(lens, tree_bitmask=tree.gdev["bitmask"], n_tree=tree.tree_size)
            target.initialize_cuda_graph([tree.tree_size], tree_bitmask=tree.gdev["bitmask"], n_tree=tree.tree_size)
            # clear_kv() inside initialize_cuda_graph wiped the prefill: rebuild the tree
            tree = make_tree(z, meta, draft, target, DEV)
        seq = []
        for s in range(3):
            tree.construct_grow_map()
            valid, a, _, term = tree.verify()
            seq.append(valid.cpu().numpy().copy())
        outs.append(seq)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_dense_mask_signature_matches_tree_context():
    """The reference's call signature (dense additive attn_mask) and the implicit-mask fast path
    produce identical logits."""
    from sequoia_amd.Engine.Llama_modules import TreeContext
    from sequoia_amd.ops import get_ops
    z, meta = load_trace("E_64x2")
    draft, target = build_engines(z, meta, DEV)
    tree = make_tree(z, meta, draft, target, DEV)
    tree.construct_grow_map()
    gt, n, M = tree.ground_truth_len, tree.tree_size, meta["M"]
    tot = gt + n - 1
    ids = tree.tokens[:tot].unsqueeze(0)
    pos = tree.position_ids[:tot].unsqueeze(0)
    sid = tree.storage_ids[:tot]
    ctx = TreeContext(q_slot0=0, gt=gt, n_tree=n, bitmask=tree.gdev["bitmask"], kv_len=tot)
    a = target.inference(input_ids=ids, storage_ids=sid, position_ids=pos, attn_mask=None, tree=ctx)
    target.clear_kv()
    mask = torch.empty(tot, tot, dtype=torch.float16, device=DEV)
    get_ops().tree_mask_dense(mask, 0, gt, n, tree.gdev["bitmask"])
    b = target.inference(input_ids=ids, storage_ids=sid, position_ids=pos, attn_mask=mask[None, None])
    assert torch.equal(a, b)
    # wrong mask shape -> the reference's ValueError (Engine/Llama_modules.py:238-242)
    target.clear_kv()
    with pytest.raises(ValueError):
        target.inference(input_ids=ids, storage_ids=sid, position_ids=pos, attn_mask=mask[None, None, :, :-1])


def test_tall_skinny_draft_forward_matches_general_path():
    """68m draft architecture: a tree level on the tall-skinny projections (Engine/ts_linear.py) and on the
    hipBLASLt + glue forward give the same logits up to accumulation order (few fp16 ulps) and the same KV
    cache rows."""
    from sequoia_amd.Engine.Engine import GraphInferenceEngine
    from sequoia_amd.Engine.Llama_modules import TreeContext
    from sequoia_amd.growmap import GrowMap
    M = 384
    eng = GraphInferenceEngine(max_length=M, model_name_or_path="random:JackFram/llama-68m:seed=5:gain=20",
                               dtype=torch.float16, device=DEV)
    g = GrowMap.load("A100-CNN-68m-7b-stochastic")
    bm = g.device_tensors(DEV)["bitmask"]
    torch.manual_seed(0)
    ids = torch.randint(3, 32000, (1, 160), device=DEV)
    outs, caches = [], []
    model = eng.engine.model
    ts = model.ts
    assert ts is not None
    for use in (True, False):
        model.ts = ts if use else None
        eng.clear_kv()
        pos = torch.arange(160, device=DEV)
        # prefill 126 tokens (general path in both runs), then a 34-token tree level
        eng.inference(input_ids=ids[:, :126], storage_ids=pos[:126], position_ids=pos[None, :126], attn_mask=None,
                      tree=TreeContext(0, 126, g.size, bm, 126))
        lv = eng.inference(input_ids=ids[:, 126:160], storage_ids=pos[126:160], position_ids=pos[None, 126:160],
                           attn_mask=None, tree=TreeContext(126, 126, g.size, bm, 160))
        outs.append(lv.float())
        caches.append(eng.engine.kv_cache.k_cache[:, :, :, :160].float().clone())
    model.ts = ts
    assert any(v is not None for v in ts.plan(34).values())
    assert (outs[0] - outs[1]).abs().max() < 6e-2          # logits of magnitude ~10: a few fp16 ulps
    assert (outs[0].argmax(-1) == outs[1].argmax(-1)).float().mean() > 0.9
    assert (caches[0] - caches[1]).abs().max() < 2e-2


@pytest.mark.parametrize("name", ["B_seq128", "V32k_seq128", "L_S256"])
def test_lossless_commit_order_replay_matches_oracle(name):
    """The package DEFAULT commit order (`lossless`: accepted tokens gathered first, then the bonus token; ADVICE r04: the
    default path must have trace coverage of its own).  Every step of the native loop on a trace's weights / prompt / noise
    must equal the oracle's verification of the step's own inputs with gather_first=True, and the committed text must be
    exactly the accepted path + the bonus token -- no bonus id over an accepted token, whichever slot it sat in."""
    from oracle import ops_np as O
    z, meta = load_trace(name)
    draft, target = build_engines(z, meta, DEV)
    tree = make_tree(z, meta, draft, target, DEV, commit_order="lossless")
    succ, n, T = meta["successors"], len(meta["successors"]), meta["T"]
    r16 = tree.r.cpu().numpy()
    quirk_candidates = 0
    for s in range(int(z["n_steps"])):
        tree.construct_grow_map()
        gt = tree.ground_truth_len
        tokens_pre = tree.tokens.cpu().numpy().copy()
        dl = tree.draft_logits[:n].cpu().numpy().copy()
        valid, a, _, term = tree.verify()
        tl = tree.target_logits.cpu().numpy()
        want_tokens = tokens_pre.copy()
        res = O.verify_stochastic(tl, dl.copy(), want_tokens, r16, succ, gt, T, int(z["bonus_u24"][s]), gather_first=True)
        assert res["accept_len"] == int(a) and bool(res["terminal"]) == bool(term), f"{name} step {s}"
        got = valid.cpu().numpy()
        assert np.array_equal(got, want_tokens[:got.shape[0]]), f"{name} step {s}: committed tokens differ from the oracle's lossless order"
        slots = res["slots"]
        assert np.array_equal(got[gt:a], tokens_pre[slots]), f"{name} step {s}: the committed text is not the accepted path"
        if not term:
            assert got[a] == res["bonus"]
        quirk_candidates += int(a in slots)              # steps where the reference's order would have committed a bonus id
        if term:
            break
    assert tree.quirk_steps == 0
    print(f"{name}: lossless replay == oracle in every step; {quirk_candidates} step(s) where the reference order would differ")


@pytest.mark.parametrize("arch,rows", [("JackFram/llama-68m", 34), ("princeton-nlp/Sheared-LLaMA-1.3B", 13)])
def test_kv_only_forward_writes_the_full_forward_kv_rows(arch, rows):
    """The draft forward over the last tree level needs only its KV rows (TreeContext.need_logits = False: the forward stops after
    its last layer's RoPE + KV write and returns None).  Every K / V row of every layer must equal, bit for bit, what the full
    forward writes -- they are what the next step's context is built from when a leaf is accepted."""
    from sequoia_amd.Engine.Engine import GraphInferenceEngine
    from sequoia_amd.Engine.Llama_modules import TreeContext
    from sequoia_amd.growmap import GrowMap
    M = 384
    eng = GraphInferenceEngine(max_length=M, model_name_or_path=f"random:{arch}:seed=5:gain=20", dtype=torch.float16, device=DEV)
    assert eng.engine.model.ts is not None
    g = GrowMap.load("A100-CNN-68m-7b-stochastic")
    bm = g.device_tensors(DEV)["bitmask"]
    torch.manual_seed(1)
    n0 = 126
    ids = torch.randint(3, 32000, (1, n0 + rows), device=DEV)
    pos = torch.arange(n0 + rows, device=DEV)
    caches = []
    for need in (True, False):
        eng.clear_kv()
        eng.inference(input_ids=ids[:, :n0], storage_ids=pos[:n0], position_ids=pos[None, :n0], attn_mask=None,
                      tree=TreeContext(0, n0, g.size, bm, n0))
        ctx = TreeContext(n0, n0, g.size, bm, n0 + rows, contiguous_slots=True, need_logits=need)
        out = eng.inference(input_ids=ids[:, n0:], storage_ids=pos[n0:], position_ids=pos[None, n0:], attn_mask=None, tree=ctx)
        assert (out is None) == (not need)
        assert eng.engine.kv_cache.kv_offset == n0 + rows
        kc = eng.engine.kv_cache
        caches.append((kc.k_cache[..., :n0 + rows, :].clone(), kc.v_cache[..., :n0 + rows, :].clone()))
    assert torch.equal(caches[0][0], caches[1][0]) and torch.equal(caches[0][1], caches[1][1])
    assert caches[0][0][..., n0:, :].abs().sum() > 0
