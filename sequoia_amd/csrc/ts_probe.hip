// Measurement aid for profiles/r03_prefetch_probe.md (kept out of ts_linear.hip so that the PMC record's source stamp of the
// production kernel does not move with it).
#include "common.h"
#define TS_WAVES 4

// ---- experiment (tools/prefetch_probe.py): touch the cache lines a projection's workgroups load first ---------------------------
// One dword per 128-byte line of the first `depth` k-steps of every (workgroup, wave, column tile) of the launch
// (tiles x splits), issued from workgroup b of THIS launch for projection workgroups b, b + grid, ... -- with grid a multiple
// of 8 the toucher runs on the XCD (b % 8) whose L2 the projection's workgroup will read from.  Measures what a run-ahead
// weight prefetch inside the preceding small kernel could buy (profiles/r03_prefetch_probe.md).
__global__ void __launch_bounds__(256) ts_prefetch_kernel(const char* __restrict__ w, int n_units, int ksteps, int silu, int tiles,
                                                          int splits, int nt_max, int depth, int* sink) {
    const int tid = threadIdx.x;
    int acc = 0;
    for (int b = blockIdx.x; b < tiles * splits; b += gridDim.x) {
        const int tile = b % tiles, split = b / tiles;
        const int u0 = (int)((long)tile * n_units / tiles), u1 = (int)((long)(tile + 1) * n_units / tiles);
        const int tpu = silu ? 2 : 1;
        const int ntl = (u1 - u0) * tpu;
        const int parts = splits * TS_WAVES;
        const int per = (ksteps + parts - 1) / parts;
        // items: (wave, d, t, line)  ->  4 * depth * ntl * 8
        const int items = TS_WAVES * depth * ntl * 8;
        for (int it = tid; it < items; it += 256) {
            const int line = it & 7, r = it >> 3;
            const int t = r % ntl, r2 = r / ntl;
            const int d = r2 % depth, wave = r2 / depth;
            const int ks0 = min(ksteps, (split * TS_WAVES + wave) * per), ks1 = min(ksteps, ks0 + per);
            const int nst = ks1 - ks0;
            if (nst <= 0) continue;
            const int rot = (int)(((unsigned)tile * 2654435761u >> 8) % (unsigned)nst);
            int i_ = min(d, nst - 1) + rot; i_ = i_ >= nst ? i_ - nst : i_; i_ = i_ >= nst ? i_ - nst : i_;
            const int ks = ks0 + i_;
            const int wtile = u0 + t / tpu + ((silu && (t & 1)) ? n_units : 0);
            acc += *(const int*)(w + ((size_t)wtile * ksteps + ks) * 1024 + line * 128);
        }
    }
    if (acc == 0x7fffffff) *sink = acc;
}

extern "C" int sq_linear_ts_prefetch(const void* w_frag, int n_out, int k, int silu, int tiles, int splits, int depth, int grid,
                                     void* sink, void* stream) {
    if (!w_frag || !sink || n_out <= 0 || k <= 0 || tiles < 1 || splits < 1 || depth < 1 || grid < 1) return SQ_EINVAL;
    const int units = n_out / 16;
    if (tiles > units) tiles = units;
    hipLaunchKernelGGL(ts_prefetch_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)w_frag, units, k >> 5, silu,
                       tiles, splits, 0, depth, (int*)sink);
    return sq_check_launch();
}
