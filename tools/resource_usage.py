#!/usr/bin/env python
"""Register / scratch / LDS usage of every kernel of a HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage
remarks (cross-compiles gfx950 without a GPU).  `python tools/resource_usage.py ts_linear.hip [--all]` prints the kernels
that spill or exceed 256 VGPRs (or all of them)."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from sequoia_amd.build import CSRC, EXTRA_FLAGS, FLAGS, hipcc  # noqa: E402


def usage(src):
    path = src if os.path.exists(src) else os.path.join(CSRC, src)
    cmd = [hipcc()] + FLAGS + EXTRA_FLAGS.get(os.path.basename(path), []) + ["-Rpass-analysis=kernel-resource-usage", "-c", path,
                                                                             "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    out = []
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = b.split()[0]
        g = lambda k: int(re.search(re.escape(k) + r": (\d+)", b).group(1))
        out.append(dict(name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("TotalSGPRs"), vgpr_spill=g("VGPRs Spill"),
                        sgpr_spill=g("SGPRs Spill"), scratch=g("ScratchSize [bytes/lane]"), lds=g("LDS Size [bytes/block]"),
                        occupancy=g("Occupancy [waves/SIMD]")))
    names = subprocess.run(["c++filt"] + [r["name"] for r in out], capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(out, names):
        r["demangled"] = n.strip()
    return out


if __name__ == "__main__":
    show_all = "--all" in sys.argv
    for src in [a for a in sys.argv[1:] if not a.startswith("--")]:
        for r in usage(src):
            if show_all or r["vgpr_spill"] or r["scratch"] or r["vgpr"] + r["agpr"] > 256:
                print(f"{r['demangled'][:80]:80s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} spill {r['vgpr_spill']:3d} "
                      f"scratch {r['scratch']:4d} lds {r['lds']:6d} occ {r['occupancy']}")
