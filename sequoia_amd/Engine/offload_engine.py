"""OffloadEngine slot of the reference API (Engine/offload_engine.py:416-451).

The reference fits Llama-2-70B on one 48 GB GPU by streaming layer weights from pinned host
memory.  An MI355X node has 8 x 288 GB of HBM3E, so the 70B target is instead sharded
tensor-parallel across the GPUs of one node with RCCL all-reduce over xGMI (BASELINE.json
north_star); see tp_engine.TPEngine.  `OffloadEngine` keeps the reference's
constructor and method set and delegates to it.
"""
from __future__ import annotations

from .tp_engine import TPEngine


class OffloadEngine(TPEngine):
    def __init__(self, max_length: int, model_name_or_path, dtype=None, device="cuda:0", stay_layers=None, **kw):
        import torch
        super().__init__(max_length=max_length, model_name_or_path=model_name_or_path,
                         dtype=dtype or torch.float16, device=device, **kw)
