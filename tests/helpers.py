"""Shared test helpers: build engines from a golden trace's recorded weights and replay the
speculation loop."""
from __future__ import annotations

import numpy as np
import torch

from conftest import load_trace


def state_dict_of(z, prefix):
    sd = {}
    for k in z.files:
        if k.startswith(prefix + "/"):
            sd[k[len(prefix) + 1:]] = torch.from_numpy(z[k])
    return sd


def dims_dict(dims5, vocab):
    hidden, inter, layers, heads, kv = dims5
    return dict(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                num_attention_heads=heads, num_key_value_heads=kv, max_position_embeddings=2048)


def trace_state_dicts(z, meta):
    """(draft, target) state dicts of a trace: stored arrays, or regenerated from the recorded seeds
    (oracle/seeded_weights.py) and checked against the recorded checksums."""
    sm = meta.get("seeded")
    if not sm:
        return state_dict_of(z, "draft"), state_dict_of(z, "target")
    from oracle import seeded_weights as SW
    sd_t = SW.seeded_state_dict(tuple(meta["target_dims"]), meta["vocab"], sm["target_seed"], meta["logit_gain"],
                                  branch_scale=sm.get("branch_scale", 1.0))
    sd_d = SW.seeded_state_dict(tuple(meta["draft_dims"]), meta["vocab"], sm["draft_seed"], meta["logit_gain"],
                                  branch_scale=sm.get("branch_scale", 1.0))
    if sm["share_vocab"] > 0.0:
        SW.correlate(sd_d, sd_t, sm["share_vocab"], sm["share_seed"])
    assert str(SW.checksum(sd_d)) == sm["draft_checksum"] and str(SW.checksum(sd_t)) == sm["target_checksum"], \
        "seeded weights differ from the ones the reference trace was generated with (torch CPU generator drift)"
    return sd_d, sd_t


def build_engines(z, meta, device):
    from sequoia_amd.Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG
    M = meta["M"]
    sd_d, sd_t = trace_state_dicts(z, meta)
    dspec = dict(state_dict=sd_d, config=dims_dict(meta["draft_dims"], meta["vocab"]))
    tspec = dict(state_dict=sd_t, config=dims_dict(meta["target_dims"], meta["vocab"]))
    draft = GraphInferenceEngine(max_length=M, model_name_or_path=dspec, dtype=torch.float16, device=device)
    target = GraphInferenceEngineTG(max_length=M, model_name_or_path=tspec, dtype=torch.float16, device=device)
    return draft, target


def make_tree(z, meta, draft, target, device, cls=None, step_graph=None):
    from sequoia_amd.growmap import GrowMap
    from sequoia_amd.Tree.GreedyTree import GreedyTree
    from sequoia_amd.Tree.SpecTree import SpecTree
    M = meta["M"]
    g = GrowMap.from_successors(meta["successors"]).to_reference_dict()
    if cls is None:
        from sequoia_amd.Tree.GreedySTree import GreedySTree
        from sequoia_amd.Tree.SpecInferTree import SpecInferTree
        cls = {"stochastic": SpecTree, "greedy": GreedyTree, "specinfer": SpecInferTree, "greedys": GreedySTree}[meta["mode"]]
    torch.manual_seed(meta["seed"] + 7)          # same noise seed as oracle/gen_golden.py
    tree = cls(prefix=torch.from_numpy(z["prompt"]), device=device, temperature=meta["T"], top_p=1.0, draft_kv_len=0,
               target_kv_len=0, draft_model_engine=draft, target_model_engine=target, max_length=M, max_target_seq=M,
               grow_map=g, attn_mask=torch.full((M, M), torch.finfo(torch.float16).min, dtype=torch.float16, device=device),
               sequence=None, new_tokens_buffer=None, parents_buffer=None,
               position_ids=torch.zeros(M, device=device).long(), residual_graph=None, sampling_callables=None,
               sample_gather_indices=None, vocab_size=meta["vocab"],
               bonus_uniforms=[int(x) for x in z["bonus_u24"]], step_graph=step_graph,
               commit_order="reference")        # the traces are runs of the reference itself (bonus stored before the gather)
    if meta["mode"] == "specinfer":
        tree.draw_uniforms = [z["draw_u24"][i] for i in range(z["draw_u24"].shape[0])]     # the trace's uniforms, per step
    if meta["mode"] == "greedys":
        tree.target_uniforms = [z["target_u24"][i] for i in range(z["target_u24"].shape[0])]
    return tree


def replay_trace(name, device, max_steps=None):
    """Run the native loop on a trace's weights / prompt / noise.  Returns per-step records
    (valid tokens, accept length) next to the reference's."""
    z, meta = load_trace(name)
    draft, target = build_engines(z, meta, device)
    tree = make_tree(z, meta, draft, target, device)
    n_steps = int(z["n_steps"]) if max_steps is None else min(max_steps, int(z["n_steps"]))
    out = []
    for s in range(n_steps):
        tree.construct_grow_map()
        tokens_pre = tree.tokens.cpu().numpy().copy()
        draft_logits = tree.draft_logits.float().cpu().numpy().copy()
        valid, a, _, terminal = tree.verify()
        out.append(dict(valid=valid.cpu().numpy().copy(), accept_len=int(a), terminal=bool(terminal),
                        tokens_pre=tokens_pre, draft_logits=draft_logits,
                        target_logits=tree.target_logits.float().cpu().numpy().copy(),
                        ref_valid=z[f"step{s}/valid_tokens"],
                        ref_tokens_pre=z[f"step{s}/tokens_pre"], ref_accept_len=int(z[f"step{s}/accept_len"]),
                        gt=int(z[f"step{s}/gt"])))
        if terminal:
            break
    return out, tree, draft, target, z, meta


def check_replay(steps, z, meta, logit_tol=4e-2):
    """Compare a replay with the reference trace, step by step.

    Requirements (the parity statement of DESIGN.md §3):
      * committed tokens entering every compared step are identical;
      * for every tree node whose token path equals the reference's, the draft and target
        logits agree within `logit_tol` (a few fp16 ulps at |logit| ~ 8: GEMM / attention
        accumulation order is the only difference);
      * the accepted tokens of the step are identical -- unless the run is stochastic and the
        step is the first one where a decision margin fell inside that tolerance, in which case
        the comparison stops there (returned as `diverged_at`).
    Returns (n_matched_steps, diverged_at or None)."""
    import numpy as np
    succ = meta["successors"]
    n = len(succ)
    parent = {c: p for p, ch in enumerate(succ) for c in ch}
    for s, rec in enumerate(steps):
        gt = rec["gt"]
        assert np.array_equal(rec["tokens_pre"][:gt], rec["ref_tokens_pre"][:gt]), f"step {s}: committed tokens differ"
        same_tok = rec["tokens_pre"][gt - 1:gt + n - 1] == rec["ref_tokens_pre"][gt - 1:gt + n - 1]
        ok = [0]
        okset = {0}
        for t in range(1, n):
            if parent[t] in okset and same_tok[t]:
                ok.append(t); okset.add(t)
        ref_d = z[f"step{s}/draft_logits_pre"].astype(np.float32)
        ref_t = z[f"step{s}/target_logits"].astype(np.float32)
        stride = meta.get("compact") or 1          # compact traces keep every stride-th logit column
        internal = [t for t in ok if len(succ[t])]
        dd = np.abs(rec["draft_logits"][internal][:, ::stride] - ref_d[internal]).max() if internal else 0.0
        dt = np.abs(rec["target_logits"][ok][:, ::stride] - ref_t[ok]).max()
        assert dd <= logit_tol and dt <= logit_tol, f"step {s}: logits off by {dd:.4f} / {dt:.4f}"
        if rec["accept_len"] == rec["ref_accept_len"] and np.array_equal(rec["valid"], rec["ref_valid"]):
            continue
        assert meta["mode"] != "greedy", f"greedy step {s} must be bit-exact"
        return s, s
    return len(steps), None


def assert_replay_complete(name, steps, tree, z, meta, matched, diverged, commit_order="reference"):
    """Every step of a trace must reproduce the reference's committed tokens.  A run with sampled decisions may leave
    the reference only where that is attributable to the (asserted) logit tolerance: at the first differing step the
    oracle, fed the NATIVE run's own logits / tokens / noise, must reproduce the native run's decisions -- i.e. the
    kernels are exact on their inputs and only the inputs differ within tolerance -- or a decision margin
    |p[tok] - r q[tok]| is below 1e-3 (one fp16 ulp of p).  Greedy traces must match in every step."""
    from oracle import ops_np as O
    n_steps = int(z["n_steps"])
    if diverged is None:
        assert matched == n_steps, f"{name}: replay stopped after {matched} of {n_steps} steps"
        return
    mode = meta["mode"]
    assert mode != "greedy", f"{name}: greedy step {diverged} must be bit-exact"
    rec = steps[diverged]
    succ, gt, n, T = meta["successors"], rec["gt"], len(meta["successors"]), meta["T"]
    dl, tl = rec["draft_logits"].astype(np.float16), rec["target_logits"].astype(np.float16)
    got_t = rec["tokens_pre"][gt - 1:gt + n - 1]
    # 1. draft expansion: the native children of every internal node are what the oracle draws from the native rows
    for t in range(n):
        k = len(succ[t])
        if not k:
            continue
        kids = got_t[succ[t]]
        if mode == "stochastic":
            keys = O.sample_keys(dl[t][None], tree.rand[t].cpu().numpy()[None], T)[0]
            want = O.sample_wor(dl[t][None], tree.rand[t].cpu().numpy()[None], k, T)[0]
            for a_, b_ in zip(kids, want):
                ka, kb = int(keys[a_].view(np.int16)), int(keys[b_].view(np.int16))
                assert a_ == b_ or abs(ka - kb) <= 1, f"{name} step {diverged}: child of node {t} off by more than a key ulp"
        elif mode == "specinfer":
            want = O.sample_iid(dl[t][None], z["draw_u24"][diverged][t][None, :k], k, T)[0]
            assert (kids != want).sum() <= 1, f"{name} step {diverged}: draws of node {t} differ from the oracle's"
        else:                                           # greedys: top-k children
            assert np.array_equal(kids, O.topk_ids(dl[t][None], k)[0]), f"{name} step {diverged}: top-k of node {t}"
    # 2. verification on the native inputs
    tokens = rec["tokens_pre"].copy()
    margins = []
    if mode == "greedys":
        tt = O.sample_iid(tl[:n], z["target_u24"][diverged][:, None], 1, T)[:, 0]
        res = O.verify_tokens(tt, tokens, succ, gt)
    else:
        res = O.verify_stochastic(tl, dl, tokens, z["r"], succ, gt, T, int(z["bonus_u24"][diverged]), margins=margins,
                                  replace=(mode == "specinfer"), gather_first=(commit_order == "lossless"))
    same = res["accept_len"] == rec["accept_len"] and np.array_equal(tokens[:len(rec["valid"])], rec["valid"])
    tight = bool(margins) and min(abs(m) for m in margins) < 1e-3
    assert same or tight, (f"{name} step {diverged}: native decisions differ from the oracle on the native run's own "
                           f"inputs and no decision margin is below 1e-3 ({margins})")
