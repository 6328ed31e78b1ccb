"""GEMM algorithm selection for the dense projections.

For the projections that stay on PyTorch-ROCm (prompt prefill, tensor-parallel shards, and the few tree-forward
shapes where hipBLASLt beats the tall-skinny kernel, Engine/ts_linear.py) what can be chosen is *which*
hipBLASLt / rocBLAS solution PyTorch dispatches for each (M, N, K).  The library heuristics pick
tiles that leave CUs idle for the skinny verify shapes (M = tree size 128, N = 4096: o_proj and
down_proj run at 1.4-1.5 TB/s of weight streaming); PyTorch's TunableOp times the candidate
solutions once and records the winner.  `enable()` turns it on with an in-tree results file as
the starting point (entries for config B measured on MI355X are shipped); shapes not in the
file are tuned on first use, during engine warm-up and graph warm-up, never inside a capture.
"""
from __future__ import annotations

import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
DEFAULT_RESULTS = os.path.join(_PKG, "tunableop_gfx950_configB.csv")


def enable(results_file: str | None = None, tune_missing: bool = True, max_tuning_ms: int = 300) -> bool:
    """Returns True when TunableOp is active.  Safe to call on builds without it."""
    try:
        tun = torch.cuda.tunable
    except AttributeError:
        return False
    src = results_file or DEFAULT_RESULTS
    tun.enable(True)
    tun.tuning_enable(bool(tune_missing))
    tun.set_max_tuning_duration(int(max_tuning_ms))
    if hasattr(tun, "set_rotating_buffer_size"):
        tun.set_rotating_buffer_size(512)      # MB: time candidates HBM-cold (weights never sit in the 256 MiB MALL)
    # read the shipped winners; write newly tuned shapes to a scratch file, not into the package
    scratch = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"sequoia_tunableop_{os.getpid()}.csv")
    tun.set_filename(scratch)
    if os.path.exists(src):
        try:
            tun.read_file(src)
        except Exception:
            pass
    if hasattr(tun, "write_file_on_exit"):
        tun.write_file_on_exit(False)
    return True
