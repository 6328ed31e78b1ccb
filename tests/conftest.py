import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_trace(name):
    z = np.load(os.path.join(GOLDEN, f"trace_{name}.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


TRACE_NAMES = ["A_2chain", "B_seq128", "demo4", "C_greedy8x8", "E_64x2"]
STOCHASTIC_TRACES = ["A_2chain", "B_seq128", "demo4", "E_64x2"]
BASELINE_TRACES = ["F_specinfer", "G_greedys"]        # the paper's comparison baselines (SpecInferTree, GreedySTree)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
