"""Tensor parallelism at world size 2 on the HIP kernels with one GPU: two ranks share cuda:0 and exchange over gloo
(RCCL refuses two ranks on one device), so every all-reduce / all-gather hook of Engine/tp_engine.py, the KV-head split
and the vocabulary-parallel lm_head run on hardware against the reference's trace -- every step, on every rank, with
identical decisions on both ranks.  (World size 1 over RCCL: tests/test_tp_native_gpu.py; N real GPUs: bench.py --gpus N.)"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from test_tp_gloo_cpu import _worker

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["A_2chain", "E_64x2"])
def test_tp2_one_gpu_matches_reference_trace(name, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    n_steps = {"A_2chain": 6, "E_64x2": 2}
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, name, str(tmp_path), "cuda:0"), nprocs=world, join=True)
    for r in range(world):
        matched, diverged = np.load(tmp_path / f"r{r}.npy")
        assert diverged == -1 and matched == n_steps[name], f"rank {r}: {matched} steps, diverged at {diverged}"
