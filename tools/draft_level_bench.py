"""One draft-level forward of the 68m architecture (q rows after a 160-token context), graph-replayed: fused small-draft
sequence vs the tall-skinny sequence (SEQUOIA_DRAFT_FUSED).  python tools/draft_level_bench.py [rows ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequoia_amd.Engine import ts_linear  # noqa: E402
from sequoia_amd.Engine.Engine import GraphInferenceEngine  # noqa: E402
from sequoia_amd.Engine.Llama_modules import TreeContext  # noqa: E402
from sequoia_amd.growmap import GrowMap  # noqa: E402

DEV = "cuda:0"
rows = [int(a) for a in sys.argv[1:]] or [1, 19, 34]
arch = os.environ.get("DRAFT_ARCH", "JackFram/llama-68m")
eng = GraphInferenceEngine(max_length=384, model_name_or_path=f"random:{arch}:seed=5:gain=20", dtype=torch.float16, device=DEV)
g = GrowMap.load("A100-CNN-68m-7b-stochastic")
bm = g.device_tensors(DEV)["bitmask"]
ids = torch.randint(3, 32000, (1, 384), device=DEV)
pos = torch.arange(384, device=DEV)
eng.inference(input_ids=ids[:, :160], storage_ids=pos[:160], position_ids=pos[None, :160], attn_mask=None,
              tree=TreeContext(0, 160, g.size, bm, 160))
for fused in (True, False):
    ts_linear.SMALL_FUSED = fused
    for q in rows:
        def fwd():
            return eng.inference(input_ids=ids[:, 160:160 + q], storage_ids=pos[160:160 + q], position_ids=pos[None, 160:160 + q],
                                 attn_mask=None, tree=TreeContext(160, 160, g.size, bm, 160 + q))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fwd()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.inference_mode():
            with torch.cuda.graph(gph):
                for _ in range(8):
                    fwd()
        gph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            gph.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"fused={int(fused)} rows={q:3d}: {e0.elapsed_time(e1) * 1e3 / 160:7.2f} us / forward", flush=True)
