"""Static (teacher-forced) acceptance-rate vector: the estimator of the reference's tests/fast_test.py:36-108 on the
native forwards ("static variant").

For every text position i >= `start` of a tokenised row, with p = softmax(target_logits[i] / T) (optionally top-p filtered)
and the draft logits of the same position, k children are drawn from the draft WITHOUT replacement -- each draw from the
softmax of the draft logits with the earlier draws masked to -inf -- and the probability that child j is the accepted one
is accumulated:  a_0 = min(1, p[s_0] / q[s_0]);  a_j = (1 - sum_{l<j} a_l) * min(1, p_res[s_j] / q_j[s_j])  with p_res the
running residual normalize(relu(p - q)).  The mean over positions is the vector the growmap search reads
(tree_search.py:14), stored with a leading 0 like the reference (:166-168).

`acceptance_from_logits` is the arithmetic on logits tensors (any device; on the CPU generator it reproduces the
reference's draws bit for bit: tests/test_acceptance_static_cpu.py, fixture from oracle/gen_fast_test_golden.py);
`static_acceptance_vector` runs the teacher-forced forwards on the engines.  The dynamic estimator -- real speculation steps
on a star tree, tests/test_accept.py -- is growmap_tuning.measure_acceptance_vector.
"""
from __future__ import annotations

import torch


def _nucleus_(logits: torch.Tensor, top_p: float, T: float) -> torch.Tensor:
    """In place: tokens outside the top-p nucleus of softmax(logits / T) -> -inf (last dim)."""
    order_vals, order_idx = torch.sort(logits, descending=True)
    mass = torch.cumsum(torch.nn.functional.softmax(order_vals / T, dim=-1), dim=-1)
    drop = mass > top_p
    drop[..., 1:] = drop[..., :-1].clone()
    drop[..., 0] = 0
    logits[drop.scatter(-1, order_idx, drop)] = float("-inf")
    return logits


def _residual(p: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    r = p - q
    r[r < 0] = 0.0
    return r / (r.sum(dim=-1).unsqueeze(-1) + 1e-9)


@torch.no_grad()
def acceptance_from_logits(target_logits: torch.Tensor, draft_logits: torch.Tensor, labels, k: int, T: float = 0.6,
                           top_p: float = 0.9, draft_top_p: float = 0.99, start: int = 128, acc=None):
    """One tokenised row.  target_logits / draft_logits: [1, L, V] (mutated like the reference mutates its model outputs);
    labels: [L] or None (positions labelled -100 / 0 are skipped, :64); acc: running sum over earlier rows (the reference
    keeps ONE fp32 accumulator over the whole dataset, :42,106).  Returns (acc + sum of per-position vectors [k], #positions)."""
    softmax = torch.nn.functional.softmax
    if top_p < 1.0:
        _nucleus_(target_logits, top_p, T)
    p_all = softmax(target_logits / T, dim=-1).squeeze(0)
    total = torch.zeros(k) if acc is None else acc
    count = 0
    for i in range(start, p_all.shape[0]):
        if labels is not None and (labels[i] == -100 or labels[i] == 0):
            continue
        count += 1
        a = torch.zeros(k)
        p = p_all[i]
        dl = draft_logits[0][i]
        if draft_top_p < 1.0:
            _nucleus_(dl, draft_top_p, T)
        q = softmax(dl / T, dim=-1).squeeze(0)
        s = q.multinomial(num_samples=1, replacement=True)
        a[0] = min(1.0, (p[s] / q[s]))
        p = _residual(p, q)
        for j in range(k - 1):
            dl[s] = -torch.inf
            q = softmax(dl / T, dim=-1).squeeze(0)
            if torch.isnan(q).long().sum() >= 1:
                break
            q = q / q.sum(-1)
            s = q.multinomial(num_samples=1, replacement=True)
            a[j + 1] = (1 - a.sum()) * min(1, p[s] / q[s])
            p = _residual(p, q)
        total = total + a
    return total, count


@torch.no_grad()
def static_acceptance_vector(draft_engine, target_engine, rows, k: int = 16, T: float = 0.6, top_p: float = 1.0,
                             draft_top_p: float = 1.1, start: int = 128, device: str = "cuda:0"):
    """rows: iterable of token-id lists (the reference evaluates 256-token c4 rows, positions 128..255).  Each row is one
    teacher-forced forward per engine (causal: a 1-node tree context); returns the [k + 1] vector with the leading 0."""
    from .Engine.Llama_modules import TreeContext
    total = torch.zeros(k)
    n = 0
    one = torch.ones((1, 1), dtype=torch.int64, device=device)
    for ids in rows:
        ids = torch.as_tensor(ids, dtype=torch.long, device=device)
        L = ids.shape[0]
        pos = torch.arange(L, device=device)
        outs = []
        for eng in (target_engine, draft_engine):
            eng.clear_kv()
            outs.append(eng.inference(input_ids=ids.unsqueeze(0), storage_ids=pos, position_ids=pos.unsqueeze(0), attn_mask=None,
                                      tree=TreeContext(q_slot0=0, gt=L, n_tree=1, bitmask=one, kv_len=L)).float())
            eng.clear_kv()
        # logits at position i predict token i + 1: the reference's HF models return the same alignment (labels = ids)
        total, c = acceptance_from_logits(outs[0], outs[1], None, k, T, top_p, draft_top_p, start, acc=total)
        n += c
    vec = torch.zeros(k + 1)
    vec[1:] = total / max(n, 1)
    return vec
