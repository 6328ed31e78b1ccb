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

// error plumbing (host side) ----------------------------------------------------------------
void sq_set_error(hipError_t e);
static inline int sq_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sq_set_error(e); return SQ_ELAUNCH; }
    return SQ_OK;
}

// fp16 <-> sortable unsigned 16-bit (larger float => larger unsigned; NaN largest, like torch.topk)
__device__ __forceinline__ uint32_t f16_to_ordered(half_t x) {
    uint16_t b = __builtin_bit_cast(uint16_t, x);
    if ((b & 0x7fffu) > 0x7c00u) return 0xffffu;            // NaN
    return (b & 0x8000u) ? (uint16_t)~b : (uint16_t)(b | 0x8000u);
}

// wave-level reductions over 64 lanes (all lanes get the result) ------------------------------
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64);
        v = v > w ? v : w;
    }
    return v;
}
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
    v = wave_max_f32(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[l < NW ? l : 0];
    r = wave_max_f32(r);
    __syncthreads();
    return r;
}
template <int NW>
__device__ __forceinline__ float block_sum_f32(float v, float* scratch) {
    v = wave_sum_f32(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = (l < NW) ? scratch[l] : 0.0f;
    r = wave_sum_f32(r);   // fixed order => deterministic
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
