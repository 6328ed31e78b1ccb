"""Probe: does a split-K formulation (batched GEMM over K slices + sum) beat F.linear for the
N = 4096 verify projections (o_proj K=4096, down_proj K=11008) at M = 128?"""
import torch, torch.nn.functional as F
dev = "cuda:0"
M = 128
def timeit(fn, reps=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, (N, K) in {"o": (4096, 4096), "down": (4096, 11008)}.items():
    x = torch.randn(M, K, device=dev).half()
    ws = [torch.randn(N, K, device=dev).half() * 0.02 for _ in range(8)]
    i = [0]
    def lin():
        i[0] += 1; return F.linear(x, ws[i[0] % 8])
    print(name, "F.linear", round(timeit(lin), 1), "us")
    for S in (2, 4, 8):
        if K % S: continue
        Ks = K // S
        xb = x.view(M, S, Ks).transpose(0, 1)                      # [S, M, Ks] (strided)
        wbs = [w.view(N, S, Ks).permute(1, 2, 0) for w in ws]      # [S, Ks, N] (strided views, no copy)
        def sk():
            i[0] += 1
            return torch.bmm(xb, wbs[i[0] % 8]).sum(0)
        def sk32():
            i[0] += 1
            return torch.bmm(xb, wbs[i[0] % 8])
        print(name, f"bmm split-K S={S} (+sum)", round(timeit(sk), 1), "us;  bmm only", round(timeit(sk32), 1), "us")
