"""What would a run-ahead weight prefetch buy the 128-row 7B projections?  Per layer: [small kernel, projection] with and
without a toucher of the projection's first weight lines in between (sq_linear_ts_prefetch), weights rotating over 32
layers; run under rocprofv3 --kernel-trace --stats and compare the projection's average duration in the two halves.
    python tools/prefetch_probe.py [qkv|o|gate_up|down] [depth]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequoia_amd.native import check  # noqa: E402
from sequoia_amd.ops import get_ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "qkv"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shape = {"qkv": (12288, 4096, False, 128, 2), "o": (4096, 4096, False, 64, 4), "gate_up": (11008, 4096, True, 230, 1),
         "down": (4096, 11008, False, 64, 4)}[name]
n_out, k, silu, tiles, splits = shape
dev, L, rows = "cuda:0", 32, 128
ops = get_ops()
torch.manual_seed(0)
ws = [ops.repack_weight((torch.randn((2 if silu else 1) * n_out, k, device=dev) * 0.02).half()) for _ in range(L)]
x = (torch.randn(rows, k, device=dev) * 0.5).half()
xf = ops.repack_rows(x)
g = torch.ones(k, dtype=torch.float16, device=dev)
h = torch.empty(ops.frag_shape(rows, k), dtype=torch.float16, device=dev)
out = torch.empty(ops.frag_shape(rows, n_out) if silu else (rows, n_out), dtype=torch.float16, device=dev)
slab = torch.empty(8 * rows * n_out, dtype=torch.float32, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)


def seq(prefetch):
    for li in range(L):
        if prefetch:
            check(ops.lib.sq_linear_ts_prefetch(ws[li].data_ptr(), n_out, k, 1 if silu else 0, tiles, splits, depth, 128,
                                                sink.data_ptr(), torch.cuda.current_stream().cuda_stream), "prefetch")
        ops.rmsnorm_frag(x, g, h, 1e-6)                     # the small kernel in front of the projection
        ops.linear_ts(xf, ws[li], rows, n_out, k, out=out if splits == 1 else None, silu=silu, out_frag=silu, tiles=tiles,
                      splits=splits, slab=slab if splits > 1 else None)


modes = {'0': (False,), '1': (True,)}.get(os.environ.get('PROBE_MODE', ''), (False, True))
for prefetch in modes:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        seq(prefetch)
        s.synchronize()
    torch.cuda.current_stream().wait_stream(s)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        seq(prefetch)
    gph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        gph.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name} depth {depth} prefetch={int(prefetch)}: {e0.elapsed_time(e1) * 1e3 / (10 * L):7.2f} us per [small kernel + projection"
          f"{' + toucher' if prefetch else ''}]", flush=True)
