"""Static acceptance-rate estimator (sequoia_amd/acceptance_static.py) against the reference's own `evaluate`
(tests/fast_test.py:36-108), run on synthetic logits by oracle/gen_fast_test_golden.py: same CPU-generator draws, same
arithmetic -> the same vector, bit for bit, for plain / top-p / draft-top-p / T = 1 settings and skipped labels."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.mark.parametrize("name", ["plain", "topp", "hot"])
def test_static_estimator_reproduces_the_reference(name):
    from sequoia_amd.acceptance_static import acceptance_from_logits
    z = np.load(os.path.join(GOLDEN, "fast_test.npz"))
    k, top_p, dtp, T = z[f"{name}/params"]
    k = int(k)
    tl, dl, labels = z[f"{name}/target"], z[f"{name}/draft"], z[f"{name}/labels"]
    torch.manual_seed(99)                                  # the generator state the reference run started from
    total, n = torch.zeros(k), 0
    for r in range(tl.shape[0]):
        total, c = acceptance_from_logits(torch.from_numpy(tl[r].copy()), torch.from_numpy(dl[r].copy()),
                                          torch.from_numpy(labels[r][0]), k, float(T), float(top_p), float(dtp), acc=total)
        n += c
    got = (total / n).numpy()
    assert n == 2 * 12 - 2                                 # positions 128..139 of two rows, two labels skipped
    assert np.array_equal(got, z[f"{name}/out"]), (got, z[f"{name}/out"])


def test_vector_layout_feeds_the_growmap_search():
    """[0, a_1 .. a_k]: the layout tree_search reads (tree_search.py:14), produced by the engine-level wrapper's tail."""
    from sequoia_amd import tree_search
    from sequoia_amd.acceptance_static import acceptance_from_logits
    torch.manual_seed(5)
    V, L, k = 256, 136, 8
    tl = torch.randn(1, L, V) * 3
    dl = tl + torch.randn(1, L, V)
    total, n = acceptance_from_logits(tl, dl, None, k, 0.6, 1.0, 1.1)
    vec = torch.zeros(k + 1)
    vec[1:] = total / n
    assert n == L - 128 and 0.0 < float(vec[1]) <= 1.0 and float(vec.sum()) <= 1.0 + 1e-5
    cfg = dict(acceptance_rate_vector=vec.tolist() + [max(0.0, 1.0 - float(vec.sum()))], max_depth=4, max_budget=16,
               draft_time=1e-4, valid_budget=[1, 2, 4, 8, 16], target_time=[3e-3, 3.1e-3, 3.2e-3, 3.3e-3, 3.5e-3])
    g, report = tree_search.search(cfg)
    assert report["budget"] in (2, 4, 8, 16) and report["expected_accepted"] > 1.0
