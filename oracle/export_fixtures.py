"""Export DATA fixtures from the reference checkout (build container only).

Growmaps are data, not code: a growmap is fully determined by its
`Successors` list, so only that is stored (JSON); sequoia_amd.growmap rebuilds roots / branches /
mask / depth from it and tests/test_growmap.py checks the rebuild against the original .pt
contents recorded here (sha of mask/depth).  Prompts: the first rows of the reference's
pre-tokenised dataset/c4_small.json (Llama-2 vocabulary ids), first 128 tokens each.
"""
import hashlib
import json
import os

import torch

REF = os.environ.get("SEQUOIA_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GROWMAPS = {
    "2-chain": "L40_growmaps/2-chain.pt",
    "A100-CNN-68m-7b-stochastic": "A100_growmaps/68m_7b/growmaps/A100-CNN-68m-7b-stochastic.pt",
    "A100-CNN-68m-7b-greedy": "A100_growmaps/68m_7b/growmaps/A100-CNN-68m-7b-greedy.pt",
    "8x8-tree": "L40_growmaps/8x8-tree.pt",
    "A100-CNN-160m-13b-stochastic": "A100_growmaps/160m_13b/growmaps/A100-CNN-160m-13b-stochastic.pt",
    "64x2-tree": "L40_growmaps/64x2-tree.pt",
    "demo_tree": "demo_tree.pt",
    "16x8-tree": "L40_growmaps/16x8-tree.pt",
    # round 6: the large-tree path gets a bench line (other_configs.L: 68m -> Llama-2-13b dims, 256 nodes, M = 512)
    "A100-CNN-68m-13b-stochastic-S256": "A100_growmaps/68m_13b/growmaps/A100-CNN-68m-13b-stochastic-S256.pt",
}


def main():
    out = os.path.join(REPO, "sequoia_amd", "growmaps")
    os.makedirs(out, exist_ok=True)
    for name, rel in GROWMAPS.items():
        g = torch.load(os.path.join(REF, rel), weights_only=False)
        rec = dict(
            name=name, source=rel, size=int(g["size"]), Successors=g["Successors"],
            check=dict(
                roots=g["roots"], branches=g["branches"],
                mask_sha=hashlib.sha256(g["mask"].numpy().tobytes()).hexdigest(),
                depth=g["depth"].tolist()),
        )
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(rec, f, separators=(",", ":"))
        print(name, g["size"])
    with open(os.path.join(REF, "dataset", "c4_small.json")) as f:
        first = f.read(1)
        f.seek(0)
        rows = []
        if first == "[":
            data = json.load(f)
            rows = data[:32]
        else:
            for line in f:
                rows.append(json.loads(line))
                if len(rows) >= 32:
                    break
    prompts = []
    for r in rows:
        ids = r["input_ids"] if isinstance(r, dict) else r
        prompts.append([int(x) for x in ids[:128]])
    with open(os.path.join(REPO, "sequoia_amd", "growmaps", "c4_small_prompts.json"), "w") as f:
        json.dump(dict(source="dataset/c4_small.json", n=len(prompts), prompts=prompts), f, separators=(",", ":"))
    print("prompts", len(prompts), len(prompts[0]))


if __name__ == "__main__":
    main()
