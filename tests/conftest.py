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


def pytest_sessionstart(session):
    """The ABI tests load sequoia_amd/lib/libsequoia_hip.so.  The library is a build product (git-ignored): a fresh
    checkout builds it here when hipcc is present (cross-compiles gfx950 without a GPU, about a minute); a stale
    library is rebuilt by the same dependency check __graft_entry__.build() uses."""
    try:
        from sequoia_amd.build import build, hipcc
        hipcc()
    except Exception:
        return                  # no ROCm toolchain: the library must already be in the tree (GPU box snapshot)
    build(force=False, verbose=False)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How many margin-limited decisions the session ACCEPTED (tests/helpers.py::ESCAPES).  Every committed fixture is
    fail-closed (helpers.KNOWN_INPUT_LIMITED is empty since round 5), so every entry comes from fresh random inputs or live
    traces of the reference on new seeds."""
    try:
        import helpers
    except Exception:
        return
    esc = helpers.ESCAPES
    terminalreporter.write_line(f"margin-limited decisions accepted: {len(esc)} (every committed fixture is fail-closed: entries come "
                                "from fresh random inputs or live traces of the reference on new seeds)")
    for label, m in esc:
        terminalreporter.write_line(f"  {label}: margin {m:.3e}")
    for tree, (same, total, expl, unexpl) in sorted(getattr(helpers, "HARNESS_RATE", {}).items()):
        terminalreporter.write_line(f"reference-harness runs token-identical on this GPU (UNSCREENED seeds, tree {tree}): {same} / {total}; "
                                    f"{len(expl)} part from the reference's run at ONE decision that two fp16 ulps per logit (or an exact tie) "
                                    f"flip: {expl}; {len(unexpl)} otherwise: {unexpl}")
    if helpers.LOGIT_EXCESS:
        terminalreporter.write_line("logit distance to the reference's recorded logits beyond 4 fp16 ulps (draft / target / tolerance):")
        for (name, layers), (dd, dt, tol) in sorted(helpers.LOGIT_EXCESS.items()):
            terminalreporter.write_line(f"  {name} ({layers} target layers): {dd:.4f} / {dt:.4f} / {tol:.4f}")


def load_trace(name):
    z = np.load(os.path.join(GOLDEN, f"trace_{name}.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


TRACE_NAMES = ["A_2chain", "B_seq128", "demo4", "C_greedy8x8", "E_64x2", "D_160m13b"]
STOCHASTIC_TRACES = ["A_2chain", "B_seq128", "demo4", "E_64x2", "D_160m13b"]
COMPACT_TRACES = ["V32k_seq128"]        # V = 32000, seeded weights, subsampled logits + full rows of the walked path
# the headline dims (BASELINE.json configs[1] / [2]): 68m-dims draft -> Llama-2-7b-dims target, V = 32000, M = 384, 128-token
# prompt; seeded weights (13.5 GB regenerated on the GPU box), compact logits.  B_7b: SpecTree on the
# A100-CNN-68m-7b-stochastic growmap; C_7b: GreedyTree on 8x8-tree (+ recorded top-k / top-2 margins)
HEADLINE_TRACES = ["B_7b", "C_7b"]
# the same V = 32000 pair and growmap as V32k_seq128 under the reference harness's DEFAULT nucleus filter (tests/testbed.py:28:
# --P 0.9; utils.get_sampling_logits, Tree/SpecTree.py:196): the trace keeps the walked target rows raw AND filtered
TOPP_TRACES = ["B_topp09"]
# configuration D at its real widths (1.3B-dims draft -> 13B-dims target: hidden 5120, 40 heads, inter 13824), 4 layers each
# configuration E at its real widths (7B-dims draft -> 70B-dims target: hidden 8192, GQA 64:8, inter 28672), 2 layers each
WIDTH_TRACES = ["D_13b_w4", "E_70b_w2"]
# configuration D at FULL DEPTH: Sheared-LLaMA-1.3B dims (24 layers) -> Llama-2-13b dims (40 layers), 26 GB of seeded weights
# (two to three minutes of CPU generation on the GPU box, shared by the tests of one process)
DEPTH_TRACES = ["D_13b", "E_70b_w8"]
# E_70b_w8 (round 6): configuration E at 70B WIDTHS and 8 LAYERS each side (17 GB of seeded fp16 weights): E_70b_w2 pins the
# widths, this one the accumulation of rounding over depth at those widths
# Round 5: the reference's LARGE growmaps -- 193 nodes / depth 24 (L40_growmaps/8x24-tree.pt, SpecTree and GreedyTree), 256 and
# 512 nodes (A100-CNN-68m-13b-stochastic-S256 / -S512: 4 / 8 ancestor-bitmask words, up to 116 parents x 32 children per
# level, verify forwards of 193-512 rows).  "lean" traces: the samplers' inputs are draft_logits_pre[roots[i]], rand[roots[i]]
LARGE_STOCHASTIC = ["L_8x24", "L_S256", "L_S512"]
LARGE_GREEDY = ["L_8x24_greedy"]
LARGE_TRACES = LARGE_STOCHASTIC + LARGE_GREEDY
LARGE_COMPACT = ["L_S256_v32k"]          # S256 at V = 32000 (68m-dims -> 160m-dims, seeded weights, compact logits)
BASELINE_TRACES = ["F_specinfer", "G_greedys"]        # the paper's comparison baselines (SpecInferTree, GreedySTree)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
