"""The drop-in claim on the reference's OWN harness source (CPU; needs the reference checkout, skipped elsewhere).

`setup_seed`, `simulation_fast`, the engine / growmap / sampler set-up block and the `simulation_fast(...)` call of the
reference's tests/testbed.py (:35-40, :45-95, :250-285, :297-298) are compiled from the file's AST (oracle/ref_harness.py)
and executed unmodified twice: in a subprocess against the reference's own Engine / Tree / utils (its top-level modules never
enter this process), and here against `sequoia_amd.dropin` with the numpy oracle standing in for the HIP library.  Same
seeded weights, prompts and noise on both sides; the tokens every `verify()` hands back to the harness loop and the value
the harness returns must be identical.  tests/golden/harness_simulation_fast_<seed>.npz are records of the same runs
(oracle/gen_harness_golden.py) that the GPU suite replays through the library (tests/test_dropin_harness_gpu.py).

Seeds: 24 is one of the 5 in 6 unscreened seeds whose whole run (two prompts decoded to 256 tokens, ~40 verify calls, every
bonus draw) is token-identical between the two arithmetics; 27 is the sixth -- one bonus draw lands on the other side of a
CDF boundary of the residual distribution (a normalised difference of nearly equal fp16 probabilities) -- and is kept as the
test of that classification: the runs agree on every token up to that draw, differ in that ONE token, and the reference's
uniform lies within 2 % of mass of the drop-in token's interval under the reference's own residual."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTBED = "/root/reference/tests/testbed.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF_TESTBED), reason="reference checkout not present")


@pytest.fixture(scope="module")
def oracle_ops():
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    ops.set_ops_for_testing(OracleOps())
    yield
    ops.set_ops_for_testing(None)


def _reference_run(tmp, seed):
    out = str(tmp / f"ref{seed}.npz")
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "ref_harness.py"), "reference", out, str(seed)], env=env,
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return out


def test_harness_lines_are_the_references():
    """What gets compiled is what the file holds at the cited lines (a moved or edited harness fails here, not silently)."""
    from oracle import ref_harness as RH
    code = RH.compile_harness()
    src = open(REF_TESTBED).read().splitlines()
    (s0, s1), (f0, f1) = code["lines"]["defs"]
    assert src[s0 - 1].startswith("def setup_seed(") and src[f0 - 1].startswith("def simulation_fast(")
    assert "return num_decoding_steps / num_large_model_steps" in src[f1 - 1]
    a, b = code["lines"]["setup"]
    assert "draft_model = GraphInferenceEngine(" in src[a - 1] and "sample_gather_indices[i] = ith_gather_list" in src[b - 1]
    c0, c1 = code["lines"]["call"]
    assert src[c0 - 1].strip().startswith("simulation_fast(target_model=target_model")


def test_reference_harness_runs_on_the_dropin_token_identical(tmp_path, oracle_ops):
    from oracle import ref_harness as RH
    z, meta = RH.load_record(_reference_run(tmp_path, 24))
    assert meta["n_verify"] >= 20 and meta["value"] > 2.0           # a real run: two prompts decoded to 256 tokens, deep paths
    value, log = RH.run_dropin(z, meta)
    assert len(log) == meta["n_verify"], (len(log), meta["n_verify"])
    for j, (p, toks) in enumerate(log):
        assert p == int(z["verify_prompt"][j])
        assert np.array_equal(toks, RH.record_tokens(z, j)), f"verify call {j} (prompt {p}): the drop-in's tokens differ from the reference's"
    assert RH.classify_run(z, meta, log) == ("identical", None)
    assert value == meta["value"]                                    # num_decoding_steps / num_large_model_steps
    # the third prompt's labels end in -100: the harness builds its tree and never steps it (tests/testbed.py:64,80)
    assert sorted({p for p, _ in log}) == [0, 1]
    # the committed record the GPU suite replays is this very run
    zc, mc = RH.load_record(os.path.join(REPO, "tests", "golden", "harness_simulation_fast_24.npz"))
    assert mc["weight_checksums"] == meta["weight_checksums"] and mc["n_verify"] == meta["n_verify"] and mc["lines"] == meta["lines"]
    for j in range(mc["n_verify"]):
        assert np.array_equal(RH.record_tokens(zc, j), RH.record_tokens(z, j))


def test_boundary_bonus_draw_is_classified(tmp_path, oracle_ops):
    from helpers import note_escape
    from oracle import ref_harness as RH
    z, meta = RH.load_record(_reference_run(tmp_path, 27))
    value, log = RH.run_dropin(z, meta)
    kind, info = RH.classify_run(z, meta, log)
    if kind == "identical":
        return                                        # (another torch build may round the other way: identity is the better outcome)
    assert kind == "boundary", (kind, info)           # the runs part at ONE bonus draw, nowhere else
    call, dist = info
    assert dist <= 2e-2, dist
    note_escape(f"reference harness seed 27, verify call {call}: bonus draw at a CDF boundary", dist)
