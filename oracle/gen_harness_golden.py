"""Record of the reference's own harness (tests/testbed.py `simulation_fast` + set-up, compiled from the file's AST and run on
the reference's classes: oracle/ref_harness.py) for the GPU suite to replay: prompts, bonus uniforms, the tokens every
verify() returned, the harness's return value, the growmap's Successors and the checksums of the seeded weights.  Runs only
where /root/reference exists.

    python oracle/gen_harness_golden.py [seed ...]      # -> tests/golden/harness_simulation_fast_<seed>.npz
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    for seed in [int(a) for a in sys.argv[1:]] or [24]:
        out = os.path.join(REPO, "tests", "golden", f"harness_simulation_fast_{seed}.npz")
        subprocess.run([sys.executable, os.path.join(REPO, "oracle", "ref_harness.py"), "reference", out, str(seed)], check=True,
                       env=dict(os.environ, PYTHONPATH=REPO), cwd=REPO)
        print("->", out, os.path.getsize(out), "bytes")
