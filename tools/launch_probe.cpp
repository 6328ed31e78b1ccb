// What a dependent kernel boundary costs as a function of the kernel's resource footprint (MI355X).
//   hipcc -O2 --offload-arch=gfx950 tools/launch_probe.cpp -o tools/launch_probe && tools/launch_probe
// A chain of N identical, dependent launches (one stream, captured into a hipGraph and replayed): time per launch of
//   an empty 256-thread kernel, the same with a large dynamic LDS allocation, the same with a 512-register footprint,
//   and with both -- the footprint of ts_linear_kernel<8,6,3,true> (512 VGPRs, 139 KB of LDS).
// Question: the tall-skinny projections measure 3.5 us for "nothing in the kernel", a trivial kernel
// boundary is 1.45 us; which resource is the difference made of?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

extern __shared__ float dyn_lds[];

__global__ void __launch_bounds__(256) k_small(float* out, int flag) {
    if (flag == 12345) out[threadIdx.x] = 1.f;
}
__global__ void __launch_bounds__(256) k_lds(float* out, int flag) {
    if (flag == 12345) { dyn_lds[threadIdx.x] = 1.f; __syncthreads(); out[threadIdx.x] = dyn_lds[255 - threadIdx.x]; }
}
// a register footprint of ~500 VGPRs that the compiler cannot shrink: the values are live across an opaque asm
template <bool LDS>
__global__ void __launch_bounds__(256) k_regs(float* out, int flag) {
    if (flag == 12345) {
        float v[480];
#pragma unroll
        for (int i = 0; i < 480; ++i) { v[i] = out[i * 256 + threadIdx.x]; }
        asm volatile("" ::: "memory");
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 480; ++i) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(v[i])); }
        if (LDS) { dyn_lds[threadIdx.x] = s; __syncthreads(); s = dyn_lds[255 - threadIdx.x]; }
        out[threadIdx.x] = s;
    }
}

template <typename F>
static double chain(F launch, int n, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    double best = 1e9;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e3 / n < best) best = ms * 1e3 / n;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float* out; CK(hipMalloc(&out, 480 * 256 * 4 + 4096));
    const int n = 400;
    const int lds_big = 139 * 1024;
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_regs<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int wgs : {64, 230, 256, 512}) {
        printf("workgroups %4d:", wgs);
        printf("  empty %.2f us", chain([&] { hipLaunchKernelGGL(k_small, dim3(wgs), dim3(256), 0, st, out, 0); }, n, st));
        for (int kb : {32, 64, 80, 139, 160})
            printf("  lds%dK %.2f", kb, chain([&] { hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(256), kb * 1024, st, out, 0); }, n, st));
        printf("  regs512 %.2f", chain([&] { hipLaunchKernelGGL(k_regs<false>, dim3(wgs), dim3(256), 0, st, out, 0); }, n, st));
        printf("  regs512+lds139K %.2f us\n", chain([&] { hipLaunchKernelGGL(k_regs<true>, dim3(wgs), dim3(256), lds_big, st, out, 0); }, n, st));
    }
    return 0;
}
