"""Records of the reference's own harness (tests/testbed.py `simulation_fast` + set-up, compiled from the file's AST and run on
the reference's classes: oracle/ref_harness.py) for the GPU suite to replay: prompts, bonus uniforms, the tokens every
verify() returned, the sparse residual every bonus token was drawn from, the harness's return value, the growmap's Successors
and the checksums of the seeded weights (a few KB per record).  Runs only where /root/reference exists.

    python oracle/gen_harness_golden.py                 # the whole committed set:
        tests/golden/harness_simulation_fast_{24,25,28}.npz     33-node tree (L40_growmaps/4x8-tree.pt); seeds whose CPU
                                                                 drop-in run is token-identical (tests/test_reference_harness_cpu.py)
        tests/golden/harness_unscreened_4x8_{100..119}.npz      the same tree, 20 seeds taken AS THEY COME
        tests/golden/harness_unscreened_s128_{100..119}.npz     the 128-node A100-CNN-68m-7b-stochastic growmap, 20 seeds as they come
The unscreened sets are what the GPU suite's summary line "reference-harness seeds token-identical on this GPU" counts: nothing
was selected, every miss must be ONE bonus draw at a CDF boundary of the reference's own residual (ref_harness.classify_run).
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = [("harness_simulation_fast_{seed}", "L40_growmaps/4x8-tree.pt", [24, 25, 28]),
        ("harness_unscreened_4x8_{seed}", "L40_growmaps/4x8-tree.pt", list(range(100, 120))),
        ("harness_unscreened_s128_{seed}", "A100_growmaps/68m_7b/growmaps/A100-CNN-68m-7b-stochastic.pt", list(range(100, 120)))]

if __name__ == "__main__":
    only = sys.argv[1:]
    for pattern, growmap, seeds in SETS:
        if only and not any(o in pattern for o in only):
            continue
        for seed in seeds:
            out = os.path.join(REPO, "tests", "golden", pattern.format(seed=seed) + ".npz")
            subprocess.run([sys.executable, os.path.join(REPO, "oracle", "ref_harness.py"), "reference", out, str(seed), growmap],
                           check=True, env=dict(os.environ, PYTHONPATH=REPO), cwd=REPO, capture_output=True)
            print("->", os.path.relpath(out, REPO), os.path.getsize(out), "bytes", flush=True)
