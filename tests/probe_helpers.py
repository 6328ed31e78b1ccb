"""The loop of the reference's tests/test_accept.py:36-140 on the native acceptance probes, replaying a probe trace
(oracle/gen_golden.py::run_probe_case): same seeded weights, prompt, noise stream and bonus uniforms."""
import json

import numpy as np
import torch

from conftest import load_trace


def probe_engines(meta, device):
    from oracle import seeded_weights as SW
    from sequoia_amd.Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG
    from helpers import dims_dict
    sm, dims, vocab, M = meta["seeded"], tuple(meta["dims"]), meta["vocab"], meta["M"]
    sd_t = SW.seeded_state_dict(dims, vocab, sm["target_seed"], meta["logit_gain"])
    sd_d = SW.correlate(SW.seeded_state_dict(dims, vocab, sm["draft_seed"], meta["logit_gain"]), sd_t, meta["noise"], sm["share_seed"])
    assert str(SW.checksum(sd_d)) == sm["draft_checksum"] and str(SW.checksum(sd_t)) == sm["target_checksum"]
    cfg = dims_dict(list(dims), vocab)
    draft = GraphInferenceEngine(max_length=M, model_name_or_path=dict(state_dict=sd_d, config=cfg), dtype=torch.float16, device=device)
    target = GraphInferenceEngineTG(max_length=M, model_name_or_path=dict(state_dict=sd_t, config=cfg), dtype=torch.float16, device=device)
    return draft, target


def replay_probe(name, device):
    """Returns per-step records (native 5-tuple + sampled children) next to the reference's."""
    from sequoia_amd.Tree.GreedyTree import GreedyTreeTest
    from sequoia_amd.Tree.SpecTree import SpecTreeTest
    z, meta = load_trace(name)
    draft, target = probe_engines(meta, device)
    M, w, T = meta["M"], meta["width"], meta["T"]
    cls = SpecTreeTest if meta["mode"] == "spectest" else GreedyTreeTest
    attn_mask = torch.full((M, M), torch.finfo(torch.float16).min, dtype=torch.float16, device=device)
    position_ids = torch.zeros(M, dtype=torch.long, device=device)
    input_ids = torch.from_numpy(z["prompt"])
    torch.manual_seed(meta["seed"] + 7)
    dkv = tkv = 0
    out = []
    for s in range(int(z["n_steps"])):
        kw = dict(bonus_uniforms=[int(z["bonus_u24"][s])]) if meta["mode"] == "spectest" else {}
        tree = cls(prefix=input_ids, device=device, temperature=T, top_p=1.0, draft_kv_len=dkv, target_kv_len=tkv,
                   draft_model_engine=draft, target_model_engine=target, max_length=M, attn_mask=attn_mask, sequence=None,
                   new_tokens_buffer=None, parents_buffer=None, position_ids=position_ids, max_width=w, **kw)
        gt = len(input_ids)
        children = tree.tokens[gt:gt + w].cpu().numpy().copy()
        valid, a, a2, b, terminal = tree.verify(benchmark=True)
        ra, rb, rt = (int(x) for x in z[f"step{s}/a_b_terminal"])
        out.append(dict(children=children, ref_children=z[f"step{s}/tokens_pre"][gt:gt + w], valid=valid.cpu().numpy().copy(),
                        ref_valid=z[f"step{s}/valid_tokens"], abt=(int(a), int(b), int(bool(terminal))), ref_abt=(ra, rb, rt),
                        a2=int(a2)))
        input_ids = valid.clone().cpu()
        dkv = tkv = int(a)
        if terminal:
            break
    return out, z, meta, draft, target
