// Skinny linear layers of the draft forward: out[M, N] = f(A)[M, K] . W[N, K]^T for M <= 64 rows
// (a tree level: 1..34 tokens for the 128-node growmap), with the row-wise glue fused in:
//   prologue  NORM : A row -> weight * h(x * rsqrt(mean(x^2) + eps))   (LlamaRMSNorm_FI, fp16 rounding
//                    points of Engine/Llama_modules.py:282-288), statistics recomputed per workgroup
//                    (the whole activation is <= 64 x 3072 fp16 and L2-resident);
//   prologue  ADD  : x = a + residual first (fp16 add), the sum is written back by workgroup 0;
//   epilogue  SILU : W holds gate rows [0, N) and up rows [N, 2N); out = h(h(silu(g)) * u);
//   epilogue  RES  : out = h(acc) + residual (fp16 add, the decoder layer's skip connection).
// hipBLASLt runs these M <= 34 GEMMs at 6-16 us each plus one launch per glue op; here one launch
// streams the weight rows straight into MFMA B-fragments.
//
// Decomposition: workgroup = 16 output columns, 4 waves split K (each wave owns a contiguous K/4 slice
// and issues all its loads up front), partial sums merged through LDS.  Both operands are consumed in
// their natural row-major layout: v_mfma_f32_16x16x32_f16 wants, per lane, 8 consecutive k of one row
// (A: activation row m = lane & 15, B: weight row n = lane & 15) -- a 16-byte load each, no LDS staging.
#include "common.h"

#define SK_WAVES 4
#define SK_THREADS (SK_WAVES * 64)
#define SK_MAXM 64
#define SK_BN 16

struct SkinnyParams {
    const half_t* a;        // [M][K] activations (row stride lda)
    const half_t* res_in;   // ADD prologue: residual added to a before the norm   (may be null)
    half_t* sum_out;        // ADD prologue: where a + res_in is written (workgroup 0)
    const half_t* ln_w;     // NORM prologue: [K]
    const half_t* w;        // [N or 2N][K]
    const half_t* res_out;  // RES epilogue: [M][N] added to the result
    half_t* out;            // [M][ldo]
    int m, n, k, lda, ldo;
    float eps;
};

template <bool NORM, bool ADD, bool SILU, bool RES>
__global__ void __launch_bounds__(SK_THREADS) skinny_linear_kernel(const SkinnyParams P) {
    constexpr int MT = SK_MAXM / 16;                 // up to 4 row tiles
    __shared__ float s_inv[SK_MAXM];
    __shared__ float s_part[SK_WAVES][SILU ? 2 : 1][MT][16 * 16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * SK_BN;
    const int mt_used = (P.m + 15) >> 4;

    // ---- prologue statistics: inv_rms per row (every workgroup, redundant by design) ------------------
    if (NORM) {
        // 256 threads: 4 threads per row for up to 64 rows; each sums K/4 squares
        const int row = tid >> 2, part = tid & 3;
        float ss = 0.f;
        if (row < P.m) {
            const int kq = P.k >> 2;                 // K is a multiple of 32
            const half_t* ar = P.a + (size_t)row * P.lda + part * kq;
            const half_t* rr = ADD ? P.res_in + (size_t)row * P.lda + part * kq : nullptr;
            for (int c = 0; c < kq; c += 8) {
                half8 v = *(const half8*)(ar + c);
                if (ADD) {
                    const half8 r = *(const half8*)(rr + c);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] + (float)r[j]);
                    if (blockIdx.x == 0) *(half8*)(P.sum_out + (size_t)row * P.lda + part * kq + c) = v;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
            }
        }
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        if (part == 0 && row < SK_MAXM) s_inv[row] = (row < P.m) ? rsqrtf(ss / (float)P.k + P.eps) : 0.f;
        __syncthreads();
    }

    // ---- main loop: this wave's K slice -----------------------------------------------------------------
    const int ksteps = P.k >> 5;                      // 32-wide MFMA steps
    const int per_wave = (ksteps + SK_WAVES - 1) / SK_WAVES;
    const int ks0 = wave * per_wave, ks1 = min(ksteps, ks0 + per_wave);
    floatx4 acc[SILU ? 2 : 1][MT];
#pragma unroll
    for (int s = 0; s < (SILU ? 2 : 1); ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[s][t] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int wn = n0 + r16;                          // weight row feeding output column wn
    const bool wn_ok = wn < P.n;
    const half_t* wrow0 = P.w + (size_t)(wn_ok ? wn : 0) * P.k;
    const half_t* wrow1 = SILU ? P.w + (size_t)((wn_ok ? wn : 0) + P.n) * P.k : nullptr;
    float inv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) inv[t] = NORM ? s_inv[t * 16 + r16] : 1.f;

#pragma unroll 4
    for (int ks = ks0; ks < ks1; ++ks) {
        const int k0 = ks * 32 + g * 8;
        half8 wf0 = *(const half8*)(wrow0 + k0);
        half8 wf1;
        if (SILU) wf1 = *(const half8*)(wrow1 + k0);
        half8 lw;
        if (NORM) lw = *(const half8*)(P.ln_w + k0);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            if (t < mt_used) {
                int row = t * 16 + r16; if (row >= P.m) row = P.m - 1;
                half8 af = *(const half8*)(P.a + (size_t)row * P.lda + k0);
                if (ADD) {
                    const half8 r = *(const half8*)(P.res_in + (size_t)row * P.lda + k0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) af[j] = (half_t)((float)af[j] + (float)r[j]);
                }
                if (NORM) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const half_t nx = (half_t)((float)af[j] * inv[t]);
                        af[j] = (half_t)((float)lw[j] * (float)nx);
                    }
                }
                // D[m = activation row][n = weight row]: A operand = activations, B operand = weights
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, wf0, acc[0][t], 0, 0, 0);
                if (SILU) acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, wf1, acc[1][t], 0, 0, 0);
            }
        }
    }
    // ---- merge the K slices: C layout col = lane & 15 (output column), row = 4 g + r (activation row) --
#pragma unroll
    for (int s = 0; s < (SILU ? 2 : 1); ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (t < mt_used)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_part[wave][s][t][(g * 4 + r) * 16 + r16] = acc[s][t][r];
    __syncthreads();
    // 256 threads -> (row tile rows 16) x 16 columns per tile pass
    for (int t = 0; t < mt_used; ++t) {
        const int row = t * 16 + (tid >> 4), col = tid & 15;
        if (row < P.m && n0 + col < P.n) {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < SK_WAVES; ++w2) {
                v0 += s_part[w2][0][t][(tid >> 4) * 16 + col];
                if (SILU) v1 += s_part[w2][1][t][(tid >> 4) * 16 + col];
            }
            half_t o = (half_t)v0;
            if (SILU) {
                const float gf = (float)o;
                const half_t sg = (half_t)(gf / (1.0f + expf(-gf)));
                o = (half_t)((float)sg * (float)(half_t)v1);
            }
            if (RES) o = (half_t)((float)o + (float)P.res_out[(size_t)row * P.ldo + n0 + col]);
            P.out[(size_t)row * P.ldo + n0 + col] = o;
        }
    }
}

extern "C" int sq_linear_skinny_f16(const void* a, int lda, const void* res_in, void* sum_out, const void* ln_w,
                                    float eps, const void* w, const void* res_out, void* out, int ldo, int m, int n,
                                    int k, int silu, void* stream) {
    if (!a || !w || !out || m <= 0 || n <= 0 || k <= 0 || lda < k || ldo < n) return SQ_EINVAL;
    if (m > SK_MAXM || (k & 127) || (lda & 7) || ((uintptr_t)a & 15) || ((uintptr_t)w & 15)) return SQ_EUNSUPPORTED;
    if (res_in && (!ln_w || !sum_out)) return SQ_EINVAL;
    SkinnyParams P;
    P.a = (const half_t*)a; P.res_in = (const half_t*)res_in; P.sum_out = (half_t*)sum_out; P.ln_w = (const half_t*)ln_w;
    P.w = (const half_t*)w; P.res_out = (const half_t*)res_out; P.out = (half_t*)out;
    P.m = m; P.n = n; P.k = k; P.lda = lda; P.ldo = ldo; P.eps = eps;
    dim3 grid((n + SK_BN - 1) / SK_BN), block(SK_THREADS);
    hipStream_t st = (hipStream_t)stream;
    const bool norm = ln_w != nullptr, add = res_in != nullptr, res = res_out != nullptr;
#define SK_GO(N_, A_, S_, R_) hipLaunchKernelGGL((skinny_linear_kernel<N_, A_, S_, R_>), grid, block, 0, st, P)
    if (silu) {
        if (res) return SQ_EUNSUPPORTED;
        if (norm && add) SK_GO(true, true, true, false);
        else if (norm) SK_GO(true, false, true, false);
        else SK_GO(false, false, true, false);
    } else if (res) {
        if (norm) return SQ_EUNSUPPORTED;
        SK_GO(false, false, false, true);
    } else {
        if (norm && add) SK_GO(true, true, false, false);
        else if (norm) SK_GO(true, false, false, false);
        else SK_GO(false, false, false, false);
    }
#undef SK_GO
    return sq_check_launch();
}
