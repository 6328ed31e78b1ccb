"""Calibrated synthetic draft/target pair for benchmarking without checkpoints.

No model weights exist offline, and two independently random-initialised Llamas agree on
nothing: the draft's proposals are accepted with probability ~0 and every speculation step
yields exactly one (bonus) token, which makes "accepted tokens/sec" degenerate.  This module
builds a *synthetic language* that both models know to different degrees, so that the
acceptance statistics are non-trivial (SURVEY.md §7 "Hard parts"):

  * a token u has K ranked successors s_1(u) .. s_K(u) (fixed random permutations);
  * both models' residual streams are dominated by the last token's embedding (transformer
    branch weights are scaled down, so attention / MLP act as a context-dependent perturbation
    whose size differs between the 32-layer target and the 2-layer draft);
  * lm_head[v] = sum_k (a_k / hidden) * embed[s_k^-1(v)], hence logit(v | u) ~ a_k when
    v = s_k(u) and ~0 otherwise: a Zipf-like next-token distribution with controllable
    peakedness; the draft uses the first `hidden_draft` coordinates of the target's embedding
    (a projection) and blurred rank weights, i.e. it is a weaker model of the same language.

Architectures, dtypes, shapes and therefore every kernel's work are exactly those of the named
models; only the weight values are synthetic.  bench.py reports which pair it ran.
"""
from __future__ import annotations

import math

import torch

from .Engine.Llama_model import KNOWN_ARCHS, LlamaDims, LlamaWeights

# rank probabilities after temperature 0.6 the successor logits are solved for (tail mass is
# spread uniformly over the vocabulary); roughly the reference's acceptance-rate vector shape
RANK_PROBS = [0.52, 0.15, 0.08, 0.05, 0.035, 0.025, 0.02, 0.015, 0.01, 0.008]


def _rank_logits(vocab, temperature=0.6, probs=RANK_PROBS):
    tail_each = (1.0 - sum(probs)) / vocab
    return [temperature * math.log(p / tail_each) for p in probs]


def _weights(arch, device, seed, vocab, embed_full, perms, rank_logits, branch_scale, blur, tp_world=1, tp_rank=0):
    dims = LlamaDims(vocab_size=vocab, tp_world=tp_world, tp_rank=tp_rank, **KNOWN_ARCHS[arch])
    w = LlamaWeights.random(dims, torch.float16, device, seed)
    h = dims.hidden_size
    emb = embed_full[:, :h].contiguous()                         # projection of the shared embedding
    w.embed = emb.half()
    gen = torch.Generator(device=device); gen.manual_seed(seed + 99)
    head = torch.zeros((vocab, h), dtype=torch.float32, device=device)
    for k, (perm_inv, a) in enumerate(zip(perms, rank_logits)):
        a_k = a * (1.0 + blur * float(torch.randn((), generator=gen, device=device)))
        head += (a_k / h) * emb[perm_inv].float()
    if blur > 0:
        head += blur * (rank_logits[0] / h) * torch.randn(head.shape, generator=gen, device=device)
    full_head = head.half()
    if tp_world > 1:
        n = vocab // tp_world
        full_head = full_head[tp_rank * n:(tp_rank + 1) * n].contiguous()
    w.lm_head = full_head
    for lw in w.layers:
        lw.wo.mul_(branch_scale)
        lw.w_down.mul_(branch_scale)
    return w


def calibrated_pair_specs(draft_arch, target_arch, device, vocab=32000, seed=7, tp_world=1, tp_rank=0,
                          target_branch=0.04, draft_branch=0.5, draft_blur=0.12, draft_tp=False):
    """Returns (draft_spec, target_spec) accepted by the engines' `model_name_or_path`.  tp_world / tp_rank shard the
    target (and, with draft_tp, the draft) Megatron-style; every rank derives its shard from the same full matrices."""
    hmax = max(KNOWN_ARCHS[draft_arch]["hidden_size"], KNOWN_ARCHS[target_arch]["hidden_size"])
    gen = torch.Generator(device=device); gen.manual_seed(seed)
    embed_full = torch.randn((vocab, hmax), generator=gen, device=device, dtype=torch.float32)
    perms = []
    for k in range(len(RANK_PROBS)):
        perm = torch.randperm(vocab, generator=gen, device=device)       # s_k(u) = perm[u]
        inv = torch.empty_like(perm); inv[perm] = torch.arange(vocab, device=device)
        perms.append(inv)                                               # s_k^-1(v)
    logits = _rank_logits(vocab)
    wt = _weights(target_arch, device, 2, vocab, embed_full, perms, logits, target_branch, 0.0, tp_world, tp_rank)
    wd = _weights(draft_arch, device, 1, vocab, embed_full, perms, logits, draft_branch, draft_blur,
                  tp_world if draft_tp else 1, tp_rank if draft_tp else 0)
    return dict(weights=wd), dict(weights=wt)
