// Shared device/host helpers for libsequoia_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sequoia_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define SQ_WAVE 64

// Fragment-major activation image consumed by the tall-skinny linear layer (ts_linear.hip): the 8-element chunk
// c (columns 8c .. 8c+7) of row r of an [rows][cols] matrix lives at
// [c / 4][r / 16][(c % 4) * 16 + r % 16][8], mtp = ceil(rows / 16) row tiles.  Returns the element offset.
__device__ __forceinline__ size_t frag_chunk_offset(size_t row, int c, int mtp) {
    return ((((size_t)(c >> 2)) * mtp + (row >> 4)) * 64 + (c & 3) * 16 + (row & 15)) * 8;
}

// error plumbing (host side) ----------------------------------------------------------------
void sq_set_error(hipError_t e);
static inline int sq_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sq_set_error(e); return SQ_ELAUNCH; }
    return SQ_OK;
}

// fp16 <-> sortable unsigned 16-bit (larger float => larger unsigned; NaN largest, like torch.topk)
__device__ __forceinline__ uint32_t f16_to_ordered(half_t x) {
    const uint16_t b = __builtin_bit_cast(uint16_t, x);
    const uint32_t o = (b & 0x8000u) ? (uint16_t)~b : (uint16_t)(b | 0x8000u);
    return ((b & 0x7fffu) > 0x7c00u) ? 0xffffu : o;         // NaN (selects only: callers unroll this per element)
}

// Correctly rounded fp32 division for operands whose quotient cannot over/underflow (both here are
// fp16-derived values): one v_rcp + two fma correction steps instead of the ~10-instruction IEEE
// sequence.  Returns exactly RN(a / b) for normal a, b (Markstein): q1 = q0 + r*(a - q0*b).
__device__ __forceinline__ float div_rn(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float r = __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0);   // refine 1/b to <= 0.5 ulp-ish
    const float q0 = a * r;
    const float e = __builtin_fmaf(-q0, b, a);
    const float q1 = __builtin_fmaf(e, r, q0);
    // +-inf / NaN numerators propagate like IEEE division (a select, not an early return: callers unroll this over 8-32
    // elements and an early return became one exec-mask branch per element)
    return (__builtin_fabsf(q0) < INFINITY) ? q1 : q0;
}

// exp / log on the hardware transcendental units (v_exp_f32 / v_log_f32 are base-2, 1 ulp) with the
// base conversion carried in two-float precision, so the result stays within ~1.5 ulp of the exact
// value -- the same class as libm's expf/logf (<= 1 ulp) at a third of the instructions.  The callers
// round the result to fp16 (11 bits), where a 1-ulp fp32 difference is visible for ~1 element in 5000.
__device__ __forceinline__ float exp_fast(float x) {           // x <= 0 in every caller
#ifdef SQ_ACCURATE_TRANSCENDENTALS
    return expf(x);
#else
    x = fmaxf(x, -104.0f);                                    // exp(-104) == 0 in fp32; avoids inf - inf below
    const float t = x * 1.44269502e+00f;                      // hi part of x * log2(e)
    const float lo = __builtin_fmaf(x, 1.44269502e+00f, -t) + x * 1.92596299e-08f;
    const float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, lo * 6.93147182e-01f, r);        // r * 2^lo, 2^lo ~ 1 + lo ln 2
#endif
}
__device__ __forceinline__ float log_fast(float u) {           // u >= 0
#ifdef SQ_ACCURATE_TRANSCENDENTALS
    return logf(u);
#else
    const float l2 = __builtin_amdgcn_logf(u);                // log2(u); log2(0) = -inf
    const float hi = l2 * 6.93147182e-01f;
    const float r = __builtin_fmaf(l2, -1.90465421e-09f, hi + __builtin_fmaf(l2, 6.93147182e-01f, -hi));
    return (l2 > -INFINITY) ? r : l2;                         // a select: no branch per element in unrolled callers
#endif
}

// DPP wave reductions: rotate within 16-lane rows (row_ror 8,4,2,1), then combine the four rows
// through SGPRs (v_readlane).  ~10x cheaper than a ds_bpermute butterfly.  Result is wave-uniform.
#define SQ_DPP_ROW_ROR(n) (0x120 + (n))
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
#define SQ_STEP(n) { uint32_t w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, SQ_DPP_ROW_ROR(n), 0xf, 0xf, false); v = v > w ? v : w; }
    SQ_STEP(8) SQ_STEP(4) SQ_STEP(2) SQ_STEP(1)
#undef SQ_STEP
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
__device__ __forceinline__ float wave_max_f32_dpp(float v) {
#define SQ_STEP(n) { float w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), SQ_DPP_ROW_ROR(n), 0xf, 0xf, false)); v = fmaxf(v, w); }
    SQ_STEP(8) SQ_STEP(4) SQ_STEP(2) SQ_STEP(1)
#undef SQ_STEP
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
#define SQ_STEP(n) { float w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), SQ_DPP_ROW_ROR(n), 0xf, 0xf, false)); v += w; }
    SQ_STEP(8) SQ_STEP(4) SQ_STEP(2) SQ_STEP(1)
#undef SQ_STEP
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (a + b) + (c + d);
}

__device__ __forceinline__ uint32_t wave_sum_u32_dpp(uint32_t v) {   // caller guarantees no overflow
#define SQ_STEP(n) { v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, SQ_DPP_ROW_ROR(n), 0xf, 0xf, false); }
    SQ_STEP(8) SQ_STEP(4) SQ_STEP(2) SQ_STEP(1)
#undef SQ_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
// exact wave sum of 64 values < 2^32 as a 64-bit result: 16-bit limbs keep every partial sum < 2^23
__device__ __forceinline__ unsigned long long wave_sum_u32_wide_dpp(uint32_t v) {
    const uint32_t lo = wave_sum_u32_dpp(v & 0xffffu), hi = wave_sum_u32_dpp(v >> 16);
    return ((unsigned long long)hi << 16) + lo;
}

// wave-level reductions over 64 lanes (all lanes get the result) ------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
        unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = v > w ? v : w;
    }
    return v;
}

// block-level reductions for blocks of NW waves; `scratch` must hold NW entries of the type and
// is reusable after the call returns (two barriers inside).
template <int NW>
__device__ __forceinline__ float block_max_f32(float v, float* scratch) {
    v = wave_max_f32_dpp(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[l < NW ? l : 0];
    r = wave_max_f32_dpp(r);
    __syncthreads();
    return r;
}
template <int NW>
__device__ __forceinline__ float block_sum_f32(float v, float* scratch) {
    v = wave_sum_f32_dpp(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = (l < NW) ? scratch[l] : 0.0f;
    r = wave_sum_f32_dpp(r);   // fixed order => deterministic
    __syncthreads();
    return r;
}
template <int NW>
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* scratch) {
    v = wave_sum_u64(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) scratch[w] = v;
    __syncthreads();
    unsigned long long r = (l < NW) ? scratch[l] : 0ull;
    r = wave_sum_u64(r);
    __syncthreads();
    return r;
}
template <int NW>
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, unsigned long long* scratch) {
    v = wave_max_u64(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) scratch[w] = v;
    __syncthreads();
    unsigned long long r = (l < NW) ? scratch[l] : 0ull;
    r = wave_max_u64(r);
    __syncthreads();
    return r;
}

// Sum of `splits` fp32 split-K partials (csrc/ts_linear.hip slabs) of 8 consecutive elements at NP places: ((p0 + p1) + p2)
// + ... in split order.  Up to 4 splits every load is issued before the first add: a run-time loop `a += load` waits for
// each split's data before it issues the next load -- one memory round trip PER SPLIT in consumers that are nothing but
// latency (norm, RoPE, SwiGLU, all-reduce input).  lo[i] / hi[i]: elements 0-3 / 4-7 at sp[i].
template <int S, int NP>
__device__ __forceinline__ void slab_sum8_n(const float* const (&sp)[NP], size_t stride, floatx4 (&lo)[NP], floatx4 (&hi)[NP]) {
    floatx4 av[NP][S], bv[NP][S];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            av[i][s] = *(const floatx4*)(sp[i] + s * stride);
            bv[i][s] = *(const floatx4*)(sp[i] + s * stride + 4);
        }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        lo[i] = av[i][0]; hi[i] = bv[i][0];
#pragma unroll
        for (int s = 1; s < S; ++s) { lo[i] += av[i][s]; hi[i] += bv[i][s]; }
    }
}
template <int NP>
__device__ __forceinline__ void slab_sum8(const float* const (&sp)[NP], int splits, size_t stride, floatx4 (&lo)[NP],
                                          floatx4 (&hi)[NP]) {
    switch (splits) {                                       // wave-uniform
    case 1: slab_sum8_n<1, NP>(sp, stride, lo, hi); break;
    case 2: slab_sum8_n<2, NP>(sp, stride, lo, hi); break;
    case 3: slab_sum8_n<3, NP>(sp, stride, lo, hi); break;
    case 4: slab_sum8_n<4, NP>(sp, stride, lo, hi); break;
    case 6: slab_sum8_n<6, NP>(sp, stride, lo, hi); break;
    case 8: slab_sum8_n<8, NP>(sp, stride, lo, hi); break;
    case 12:                                               // per-head partials of the fused draft attention block (12 / 16 heads,
        if constexpr (NP == 1) { slab_sum8_n<12, NP>(sp, stride, lo, hi); break; }      // csrc/draft_block.hip): one place per thread
    case 16:
        if constexpr (NP == 1) { if (splits == 16) { slab_sum8_n<16, NP>(sp, stride, lo, hi); break; } }
    default:
        slab_sum8_n<4, NP>(sp, stride, lo, hi);
        for (int s = 4; s < splits; ++s)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                lo[i] += *(const floatx4*)(sp[i] + s * stride);
                hi[i] += *(const floatx4*)(sp[i] + s * stride + 4);
            }
    }
}

// Tree-causal visibility rule shared by the dense-mask writer and the attention kernel
// (restates the window of the reference's doubled mask: Tree/Tree.py:20-27,
// Tree/SpecTree.py:54-58).  bm_row points at the ancestor bitmask row of tree node t (or null
// when slot < gt).
__device__ __forceinline__ bool tree_visible(int slot, int col, int gt, int n_tree,
                                             const uint64_t* bm_row) {
    if (col >= gt + n_tree - 1) return false;
    if (slot < gt) return col <= slot;
    if (col < gt) return true;
    const int j = col - (gt - 1);
    return (bm_row[j >> 6] >> (j & 63)) & 1ull;
}
