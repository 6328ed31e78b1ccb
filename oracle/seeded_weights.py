"""Seeded Llama weights for golden traces that are too large to store (test infrastructure).

A trace at the real vocabulary (V = 32000) or with 13B-like head counts would carry tens of MB of
random fp16 weights.  Instead the trace records (dims, vocab, seed, gain, share) and this module
regenerates the state dict bit for bit: every tensor is drawn on the CPU generator in a fixed name
order.  `oracle/gen_golden.py` loads the same state dict into the REFERENCE model
(`load_state_dict`), so the reference run and the native replay share identical weights; the
trace's meta carries a checksum so a drifting generator is detected instead of silently
producing a different model.
"""
from __future__ import annotations

import numpy as np
import torch


def _names(layers: int):
    yield "model.embed_tokens.weight"
    for i in range(layers):
        p = f"model.layers.{i}."
        for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj",
                  "mlp.up_proj", "mlp.down_proj"):
            yield p + n + ".weight"
        yield p + "input_layernorm.weight"
        yield p + "post_attention_layernorm.weight"
    yield "model.norm.weight"
    yield "lm_head.weight"


def _shape(name: str, dims, vocab: int):
    hidden, inter, _, heads, kv = dims
    d = hidden // heads
    if name.endswith("embed_tokens.weight") or name == "lm_head.weight":
        return (vocab, hidden)
    if "layernorm" in name or name == "model.norm.weight":
        return (hidden,)
    if "q_proj" in name:
        return (heads * d, hidden)
    if "k_proj" in name or "v_proj" in name:
        return (kv * d, hidden)
    if "o_proj" in name:
        return (hidden, heads * d)
    if "gate_proj" in name or "up_proj" in name:
        return (inter, hidden)
    if "down_proj" in name:
        return (hidden, inter)
    raise KeyError(name)


def seeded_state_dict(dims, vocab: int, seed: int, logit_gain: float = 1.0, std: float = 0.02,
                      branch_scale: float = 1.0, lead=None) -> dict:
    """dims = (hidden, inter, layers, heads, kv_heads).  fp16 tensors, HF parameter names.
    lead = (cols, factor): the first `cols` hidden dimensions of the embedding and of lm_head are scaled by `factor`, so
    that a narrower draft built from those leading slices (`correlate`) predicts this model to a useful degree (the
    headline-dims traces B_7b / C_7b: 768-wide draft, 4096-wide target)."""
    gen = torch.Generator()
    gen.manual_seed(int(seed))
    sd = {}
    for name in _names(dims[2]):
        shape = _shape(name, dims, vocab)
        if len(shape) == 1:
            sd[name] = torch.ones(shape, dtype=torch.float16)
        else:
            sd[name] = torch.empty(shape, dtype=torch.float32).normal_(0.0, std, generator=gen).half()
    if logit_gain != 1.0:
        sd["lm_head.weight"] = (sd["lm_head.weight"].float() * logit_gain).half()
    if lead is not None:
        cols, factor = int(lead[0]), float(lead[1])
        for k in ("model.embed_tokens.weight", "lm_head.weight"):
            t = sd[k].float()
            t[:, :cols] *= factor
            sd[k] = t.half()
    if branch_scale != 1.0:            # damp the attention / MLP branches: the residual stream stays embedding-dominated
        for k in sd:
            if "o_proj" in k or "down_proj" in k:
                sd[k] = (sd[k].float() * branch_scale).half()
    return sd


def correlate(draft_sd: dict, target_sd: dict, noise: float, seed: int):
    """Correlated draft: every draft tensor that also exists in the target becomes the target's (leading slice, when
    the draft is narrower) plus relative Gaussian noise, so that the two models agree to a controllable degree."""
    gen = torch.Generator()
    gen.manual_seed(int(seed))
    for k in sorted(draft_sd):
        if k not in target_sd or draft_sd[k].dim() != 2:
            continue
        t = target_sd[k].float()
        r, c = draft_sd[k].shape
        if t.shape[0] < r or t.shape[1] < c:
            continue
        base = t[:r, :c]
        draft_sd[k] = (base + torch.empty(base.shape).normal_(0.0, 1.0, generator=gen) * base.std() * noise).half()
    return draft_sd


def scale_lm_head(draft_sd: dict, scale: float):
    """lm_head *= scale (after `correlate`): a draft narrower than its target normalises the residual stream over fewer
    dimensions, so the shared lm_head slice yields logits that are flatter by a constant factor; see gen_golden.run_case."""
    if scale != 1.0:
        draft_sd["lm_head.weight"] = (draft_sd["lm_head.weight"].float() * float(scale)).half()
    return draft_sd


def checksum(sd: dict) -> int:
    """Order-dependent 64-bit checksum of the fp16 bit patterns (wrap-around uint64 arithmetic)."""
    acc = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for k in sorted(sd):
            v = sd[k].contiguous().view(torch.int16).numpy().reshape(-1).astype(np.uint64)
            probe = v[::97]
            s = v.sum(dtype=np.uint64) + (probe * np.arange(1, probe.size + 1, dtype=np.uint64)).sum(dtype=np.uint64)
            acc = (acc * np.uint64(1099511628211)) ^ s
    return int(acc)
