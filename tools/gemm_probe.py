"""Probe: how fast does PyTorch-ROCm (hipBLASLt / rocBLAS) run the 7B verify projections at M=128?
Prints us and effective weight-streaming TB/s per shape and call form."""
import os, sys
import torch
import torch.nn.functional as F

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008), "lm_head": (32000, 4096)}


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print("M =", M, "tunable:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"), "blas pref:", torch.backends.cuda.preferred_blas_library())
tot = {}
for name, (N, K) in shapes.items():
    x = torch.randn(M, K, device=dev).half()
    # rotate over 8 distinct weight copies so the 256 MiB Infinity Cache cannot hold them
    ws = [torch.randn(N, K, device=dev).half() * 0.02 for _ in range(6)]
    wts = [w.t().contiguous() for w in ws]
    i = [0]
    def lin():
        i[0] += 1; return F.linear(x, ws[i[0] % 6])
    def mm_nn():
        i[0] += 1; return torch.mm(x, wts[i[0] % 6])
    for form, fn in (("F.linear(TN)", lin), ("mm(NN)", mm_nn)):
        t = timeit(fn)
        tot[form] = tot.get(form, 0) + t * (1 if name == "lm_head" else 32)
        print(f"{name:8s} {form:13s} {t:8.1f} us  {N * K * 2 / t / 1e6:6.2f} TB/s  {2 * M * N * K / t / 1e6:7.1f} TFLOP/s")
print("per-verify totals (32 layers + lm_head):", {k: f"{v / 1e3:.2f} ms" for k, v in tot.items()})
