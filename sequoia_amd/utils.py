"""Harness-facing helpers with the reference's names (utils.py): the functions the reference
wraps into CUDA graphs, and the `cuda_graph_for_*` factories the harness injects into the trees
(tests/testbed.py:256,269-276).  Here every factory returns a callable backed by one HIP
kernel; no graph is needed to make them launch-cheap.
"""
from __future__ import annotations

import dataclasses

import torch

from .ops import get_ops


def _mark_native(fn):
    fn._sequoia_native = True
    return fn


def get_residual(p: torch.Tensor, q: torch.Tensor):
    """relu(p - q) / sum(relu(p - q))  (utils.py:5-8).  Plain tensor expression: it is only used
    outside the hot path (the native verifier fuses it)."""
    residual = (p - q).relu_()
    return residual / residual.sum(dim=-1).unsqueeze(-1)


def sampling_without_replacement(sampling_logits: torch.Tensor, rand: torch.Tensor, num_samples: int,
                                 temperature: float):
    """utils.py:10-18 on the native sampler: int64 [n_rows * num_samples]."""
    logits = sampling_logits.reshape(-1, sampling_logits.shape[-1])
    out = torch.empty(logits.shape[0] * num_samples, dtype=torch.long, device=logits.device)
    get_ops().sample_wor(logits, rand.reshape(-1, rand.shape[-1]), None, num_samples, temperature, out)
    return out


def sampling_with_replacement(sampling_logits: torch.Tensor, num_samples: int, temperature: float):
    """utils.py:20-28.  Despite its name the reference draws WITHOUT replacement here
    (`multinomial(num_samples, replacement=False)`); so does this: the exponential-race sampler on fresh uniform noise
    (log(u)/q keys, top-k) is an exact draw without replacement from softmax(logits / T)."""
    rand = torch.empty_like(sampling_logits).uniform_()
    return sampling_without_replacement(sampling_logits, rand, num_samples, temperature)


def sampling_argmax(sampling_logits: torch.Tensor, num_samples: int):
    """utils.py:29-32 on the native top-k."""
    logits = sampling_logits.reshape(-1, sampling_logits.shape[-1])
    out = torch.empty(logits.shape[0] * num_samples, dtype=torch.long, device=logits.device)
    get_ops().topk(logits, None, num_samples, out)
    return out


def get_sampling_logits(logits: torch.Tensor, top_p: float, T: float, replicate=False):
    """Nucleus filter (utils.py:65-77); identity at top_p = 1.0 (all reference scripts)."""
    if replicate:
        logits = logits.clone()
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cumulative = torch.cumsum(torch.nn.functional.softmax(sorted_logits / T, dim=-1), dim=-1)
        drop = cumulative > top_p
        drop[..., 1:] = drop[..., :-1].clone()
        drop[..., 0] = 0
        logits[drop.scatter(-1, sorted_indices, drop)] = float("-inf")
    return logits


@dataclasses.dataclass
class ChildrenAccept:
    accept_mark: int = None
    token: int = None
    position: int = None
    successor_order: int = -1
    residual: torch.FloatTensor = None


def _make_causal_mask(input_ids_shape, dtype: torch.dtype, device):
    """[tgt, tgt] additive causal mask (utils.py:95-107), produced by the tree-mask kernel with an
    empty tree on HIP devices."""
    _, tgt_len = input_ids_shape
    if str(device).startswith("cuda") and dtype == torch.float16:
        out = torch.empty((tgt_len, tgt_len), dtype=dtype, device=device)
        get_ops().tree_mask_dense(out, 0, tgt_len, 1, None)
        return out
    mask = torch.full((tgt_len, tgt_len), torch.finfo(dtype).min, device=device, dtype=dtype)
    return torch.triu(mask, diagonal=1)


def cuda_graph_for_residual(device="cuda:0", dtype=torch.float16, dim=32000, n_warmups=3, mempool=None):
    return _mark_native(lambda p, q: get_residual(p, q))


def cuda_graph_for_sampling_without_replacement(device="cuda:0", dtype=torch.float16, dim=32000, max_length=384,
                                                n_warmups=3, mempool=None, idx_len=8, num_samples=16,
                                                temperature=0.6, tree_size=64):
    def run(draft_logits, rand_vector):
        return sampling_without_replacement(draft_logits, rand_vector, num_samples, temperature)
    return _mark_native(run)


def cuda_graph_for_sampling_argmax(device="cuda:0", dtype=torch.float16, dim=32000, max_length=384, n_warmups=3,
                                   mempool=None, idx_len=8, num_samples=16, temperature=0.6, tree_size=64):
    def run(draft_logits):
        return sampling_argmax(draft_logits, num_samples)
    return _mark_native(run)


def cuda_graph_for_sampling_with_replacement(device="cuda:0", dtype=torch.float16, dim=32000, max_length=384, n_warmups=3,
                                             mempool=None, idx_len=8, num_samples=16, temperature=0.6, tree_size=64):
    """utils.py:214-248 (used by tests/test_specinfer.py): one callable per tree level, logits -> drawn positions."""
    def run(draft_logits):
        return sampling_with_replacement(draft_logits, num_samples, temperature)
    return _mark_native(run)
