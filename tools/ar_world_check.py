"""The all-reduce / all-gather / all-reduce + RMSNorm checks of tests/test_xgmi_allreduce_gpu.py::_ar_worker at any world size
(the suite runs worlds 2 and 4), every rank on GPU 0:   python tools/ar_world_check.py 8"""
import os
import sys
import tempfile

import numpy as np
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
from test_xgmi_allreduce_gpu import _ar_worker  # noqa: E402

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    d = tempfile.mkdtemp()
    mp.spawn(_ar_worker, args=(world, 33900 + world, d), nprocs=world, join=True)
    res = [tuple(int(x) for x in np.load(os.path.join(d, f"ar{r}.npy"))) for r in range(world)]
    print("world", world, "(sizes, burst, graph, calls) per rank:", res)
    assert all(r[0] == 7 and r[1] == 1 and r[2] == 1 for r in res)
