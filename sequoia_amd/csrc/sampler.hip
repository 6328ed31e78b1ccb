// a2: draft-expansion samplers, every logits row split over `parts` workgroups (4096 elements each) so that a tree level
// of 1-34 rows fills 8-272 compute units instead of 1-34:
//   1. logits_stats_kernel   grid (rows, parts): y = h(x / T), per-part maximum m_p and s_p = sum exp(y - m_p)
//                            (optionally also copies the row: in the speculation step this launch replaces the copy of
//                            the draft forward's logits into the tree's draft_logits rows);
//   2. sample_parts_kernel   grid (rows, parts): row statistics M = max m_p, z = sum_p s_p exp(m_p - M) (fixed order),
//                            keys log(u) / q for the part's elements, local top-k (per-wave DPP arg-max rounds, then one
//                            wave merges the 4 lists) -> candidate list of the part;
//   3. sample_merge_kernel   one wave per row: merges parts x k candidates, writes the tokens.
// The top-k sampler (greedy trees) runs 2 + 3 on the raw logits.  Rounding points follow the fp16 torch expression of the
// reference (utils.py:10-18): y = h(x/T), q = h(exp(y-max)/sum), lu = h(log u), key = h(lu/q).
//
// Ordering = 32-bit composites (ordered fp16 key << 16 | 0xffff - token id): the maximum is the largest key and, inside
// an exact tie, the lowest token id; vocabularies above 65536 use 64-bit composites (ordered key + 1 << 32 | ~id).
#include "common.h"

#define PART_THREADS 256
#define PART_WAVES (PART_THREADS / 64)
#define PART_CH 2                                  // 16-byte chunks per thread
#define PART_ELEMS (PART_THREADS * PART_CH * 8)    // 4096
#define SAMP_MAX_PARTS 32                          // vocab <= 131072
#define SAMP_FAST_K 32                             // threshold selection + counting ranks up to this many samples per row
#define SAMP_WAVE_CAP 64                           // candidate slots per wave on that path
#define SAMP_RANK_MAX 512                          // candidates the counting merge orders (parts x samples)

__host__ __device__ static inline int samp_parts(int vocab) { return (vocab + PART_ELEMS - 1) / PART_ELEMS; }

// element index of (part p, chunk c, thread t, lane-element j) -- 16-byte chunks interleaved over threads
__device__ __forceinline__ int part_elem(int p, int c, int t, int j) { return p * PART_ELEMS + (c * PART_THREADS + t) * 8 + j; }

struct SampWs {
    float* stats;       // [n_rows][parts][2]  (m_p, s_p) when the caller supplies none
    void* cand;         // [n_rows][parts][k] composites (uint32 or uint64)
};
__host__ __device__ static inline size_t samp_stats_bytes(int n_rows, int vocab) {
    return (((size_t)n_rows * samp_parts(vocab) * 2 * sizeof(float)) + 15) & ~(size_t)15;
}
extern "C" size_t sq_sample_workspace_bytes(int n_rows, int vocab, int k) {
    if (n_rows <= 0 || vocab <= 0 || k <= 0) return 0;
    return samp_stats_bytes(n_rows, vocab) + (size_t)n_rows * samp_parts(vocab) * k * sizeof(unsigned long long) + 64;
}

// chunks past the row end: defined values (the element passes below are branch-free -- they evaluate the quotient / exp / log
// of every register and select afterwards; indeterminate registers there would be undefined behaviour)
__device__ __forceinline__ half8 samp_neg_inf8() {
    const half_t n = (half_t)(-INFINITY);
    return half8{n, n, n, n, n, n, n, n};
}
__device__ __forceinline__ half8 samp_half8(float v) {
    const half_t n = (half_t)v;
    return half8{n, n, n, n, n, n, n, n};
}

// ---- 1. per-part softmax statistics (+ optional row copy) ----------------------------------------------------------
__global__ void __launch_bounds__(PART_THREADS)
logits_stats_kernel(const half_t* __restrict__ logits, int64_t ld, const int32_t* __restrict__ row_ids, int vocab,
                    float temperature, float* __restrict__ stats, int stats_by_source_row, half_t* __restrict__ copy_dst,
                    int64_t ld_dst) {
    __shared__ float s_f[PART_WAVES];
    const int t = threadIdx.x, r = blockIdx.x, p = blockIdx.y, parts = gridDim.y;
    const int64_t row = row_ids ? (int64_t)row_ids[r] : (int64_t)r;
    const half_t* x = logits + row * ld;
    float y[PART_CH * 8];
    float lmax = -INFINITY;
    half8 xv[PART_CH];
#pragma unroll
    for (int c = 0; c < PART_CH; ++c) {                     // every load of the thread in flight before the first use
        const int e0 = part_elem(p, c, t, 0);
        xv[c] = samp_neg_inf8();
        if (e0 < vocab) xv[c] = *(const half8*)(x + e0);
    }
    if (copy_dst) {
#pragma unroll
        for (int c = 0; c < PART_CH; ++c) {
            const int e0 = part_elem(p, c, t, 0);
            if (e0 < vocab) *(half8*)(copy_dst + (int64_t)r * ld_dst + e0) = xv[c];
        }
    }
#pragma unroll
    for (int c = 0; c < PART_CH; ++c) {
        const bool in = part_elem(p, c, t, 0) < vocab;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float q = (float)(half_t)div_rn((float)xv[c][j], temperature);
            const float yy = in ? q : -INFINITY;
            y[c * 8 + j] = yy;
            lmax = fmaxf(lmax, yy);
        }
    }
    const float m = block_max_f32<PART_WAVES>(lmax, s_f);
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < PART_CH * 8; ++i) lsum += exp_fast(y[i] - m);      // exp(-inf - m) = 0; m = -inf only for an all -inf part
    const float s = block_sum_f32<PART_WAVES>(lsum, s_f);
    if (t == 0) {
        float* o = stats + ((size_t)(stats_by_source_row ? row : r) * parts + p) * 2;
        o[0] = m;
        o[1] = (m > -INFINITY) ? s : 0.f;
    }
}

// row statistics from the per-part ones, identical in every consumer: M = max m_p, z = sum_p s_p exp(m_p - M) in part order.
// Lane p of every wave loads part p's pair (ONE memory round trip instead of a dependent load per part), the maximum is a
// wave reduction, the sum runs over the lanes' terms in part order (v_readlane, no memory).
__device__ __forceinline__ void combine_stats(const float* __restrict__ st, int parts, float& M, float& z) {
    const int lane = threadIdx.x & 63;
    float m = -INFINITY, sm = 0.f;
    if (lane < parts) { m = st[lane * 2]; sm = st[lane * 2 + 1]; }
    const float mx = wave_max_f32_dpp(m);
    const float term = (m > -INFINITY) ? sm * exp_fast(m - mx) : 0.f;
    float acc = 0.f;
    for (int q = 0; q < parts; ++q) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, term), q));
    M = mx; z = acc;
}

// ---- 2. keys + local top-k of one part -----------------------------------------------------------------------------
template <typename C> struct Comp;
template <> struct Comp<uint32_t> {
    typedef uint16_t key_t;
    static __device__ __forceinline__ uint32_t make(uint32_t ord, int e) { return (ord << 16) | (0xffffu - (uint32_t)e); }
    static __device__ __forceinline__ int64_t id(uint32_t c) { return (int64_t)(0xffffu - (c & 0xffffu)); }
    static __device__ __forceinline__ uint32_t wave_max(uint32_t v) { return wave_max_u32_dpp(v); }
};
template <> struct Comp<unsigned long long> {
    typedef uint32_t key_t;
    static __device__ __forceinline__ unsigned long long make(uint32_t ord, int e) {
        return ((unsigned long long)(ord + 1u) << 32) | (uint32_t)(0xffffffffu - (uint32_t)e);
    }
    static __device__ __forceinline__ int64_t id(unsigned long long c) { return (int64_t)(0xffffffffu - (uint32_t)(c & 0xffffffffu)); }
    static __device__ __forceinline__ unsigned long long wave_max(unsigned long long v) {
        // two 32-bit DPP maxima: the high words first, then the low words among the lanes that hold the maximum high word
        const uint32_t hi = wave_max_u32_dpp((uint32_t)(v >> 32));
        const uint32_t lo = wave_max_u32_dpp(((uint32_t)(v >> 32) == hi) ? (uint32_t)v : 0u);
        return ((unsigned long long)hi << 32) | lo;
    }
};

// k rounds of wave-wide arg-max over NPL composites per lane; round s stores the winner with `put(s, win)`.
// Registers >= `used` (wave-uniform) hold nothing and are skipped with scalar branches.
template <typename C, int NPL, typename Put>
__device__ __forceinline__ void wave_topk(C (&comp)[NPL], int used, int n_out, int lane, Put put) {
    C best = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) if (i < used) best = comp[i] > best ? comp[i] : best;
    for (int s = 0; s < n_out; ++s) {
        const C win = Comp<C>::wave_max(best);            // wave-uniform
        if (lane == 0) put(s, win);
        if (win == 0) continue;
        if (best == win) {                                // the owner (ids are unique) retires it and rescans its registers
            best = 0;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                if (i < used) {
                    if (comp[i] == win) comp[i] = 0;
                    best = comp[i] > best ? comp[i] : best;
                }
            }
        }
    }
}

// fp32 -> sortable unsigned 32-bit (larger float => larger unsigned; NaN largest)
__device__ __forceinline__ uint32_t f32_to_ordered(float x) {
    const uint32_t b = __builtin_bit_cast(uint32_t, x);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xfffffffeu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// WOR: 0 = top-k of the raw logits, 1 = sampling without replacement with fp16 noise (SpecTree), 2 = the acceptance
// probe's form (SpecTreeTest, Tree/SpecTree.py:349-360): fp32 noise, key = log(u) / q evaluated in fp32 (torch promotes
// fp32 / fp16), 64-bit composites with the 32-bit ordered key
template <int WOR, typename C>
__global__ void __launch_bounds__(PART_THREADS)
sample_parts_kernel(const half_t* __restrict__ logits, int64_t ld_logits, const void* __restrict__ rnd_raw, int64_t ld_rand,
                    const int32_t* __restrict__ row_ids, int vocab, int k, float temperature,
                    const float* __restrict__ stats, int stats_by_source_row, const int32_t* __restrict__ branch,
                    C* __restrict__ cand) {
    constexpr int EPT = PART_CH * 8;
    __shared__ __attribute__((aligned(16))) C s_cand[PART_WAVES * SQ_MAX_TOPK];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = blockIdx.x, p = blockIdx.y, parts = gridDim.y;
    int n_out = k;
    if (branch) { n_out = branch[r]; if (n_out > k) n_out = k; }
    if (n_out <= 0) return;                                // uniform per block
    const int64_t row = row_ids ? (int64_t)row_ids[r] : (int64_t)r;
    const half_t* x = logits + row * ld_logits;

    C comp[EPT];
    if (WOR == 2) {
        const float* u = (const float*)rnd_raw + row * ld_rand;
        half8 xv[PART_CH];
        float uv[PART_CH][8];
#pragma unroll
        for (int c = 0; c < PART_CH; ++c) {
            const int e0 = part_elem(p, c, t, 0);
            xv[c] = samp_neg_inf8();
#pragma unroll
            for (int j = 0; j < 8; ++j) uv[c][j] = 0.5f;
            if (e0 < vocab) {
                xv[c] = *(const half8*)(x + e0);
                const floatx4 a = *(const floatx4*)(u + e0), b = *(const floatx4*)(u + e0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { uv[c][j] = a[j]; uv[c][4 + j] = b[j]; }
            }
        }
        float M, z;
        combine_stats(stats + (size_t)(stats_by_source_row ? row : r) * parts * 2, parts, M, z);
#pragma unroll
        for (int c = 0; c < PART_CH; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // computed for every element and selected at the end (branch-free: chunks past the row end hold garbage)
                const int e = part_elem(p, c, t, j);
                const float yy = (float)(half_t)div_rn((float)xv[c][j], temperature);
                const half_t q = (half_t)div_rn(exp_fast(yy - M), z);
                const float lu = log_fast(uv[c][j]);
                const float kq = div_rn(lu, (float)q);
                const float key = (q == (half_t)0.0f) ? -INFINITY : kq;
                const C v = ((C)f32_to_ordered(key) << (sizeof(C) * 4)) + (C)1 + (C)(0xfffffffeu - (uint32_t)e);
                comp[c * 8 + j] = (part_elem(p, c, t, 0) < vocab) ? v : (C)0;
            }
        }
    } else if (WOR == 1) {
        const half_t* u = (const half_t*)rnd_raw + row * ld_rand;
        half8 xv[PART_CH], uv[PART_CH];
#pragma unroll
        for (int c = 0; c < PART_CH; ++c) {
            const int e0 = part_elem(p, c, t, 0);
            xv[c] = samp_neg_inf8(); uv[c] = samp_half8(0.5f);
            if (e0 < vocab) { xv[c] = *(const half8*)(x + e0); uv[c] = *(const half8*)(u + e0); }
        }
        float M, z;
        combine_stats(stats + (size_t)(stats_by_source_row ? row : r) * parts * 2, parts, M, z);
#pragma unroll
        for (int c = 0; c < PART_CH; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // computed for every element and selected at the end (branch-free; vocab % 8 == 0: chunks are in or out as
                // a whole, chunks past the row end hold garbage)
                const int e = part_elem(p, c, t, j);
                const float yy = (float)(half_t)div_rn((float)xv[c][j], temperature);
                const half_t q = (half_t)div_rn(exp_fast(yy - M), z);
                const half_t lu = (half_t)log_fast((float)uv[c][j]);
                // lu / 0 = -inf (lu < 0 always: u < 1); otherwise the correctly rounded quotient
                const half_t kq = (half_t)div_rn((float)lu, (float)q);
                const half_t key = (q == (half_t)0.0f) ? (half_t)(-INFINITY) : kq;
                const C v = Comp<C>::make(f16_to_ordered(key), e);
                comp[c * 8 + j] = (part_elem(p, c, t, 0) < vocab) ? v : (C)0;
            }
        }
    } else {
        half8 xv[PART_CH];
#pragma unroll
        for (int c = 0; c < PART_CH; ++c) {                      // both loads in flight before the first use
            const int e0 = part_elem(p, c, t, 0);
            xv[c] = samp_neg_inf8();
            if (e0 < vocab) xv[c] = *(const half8*)(x + e0);
        }
#pragma unroll
        for (int c = 0; c < PART_CH; ++c) {
            const int e0 = part_elem(p, c, t, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const C v = Comp<C>::make(f16_to_ordered(xv[c][j]), e0 + j);
                comp[c * 8 + j] = (e0 < vocab) ? v : (C)0;
            }
        }
    }
    C* dst = cand + ((size_t)r * parts + p) * k;
    if (n_out <= SAMP_FAST_K) {
        // Threshold selection: the n_out-th largest of the 64 lane maxima bounds the wave's n_out-th largest element from
        // below, so every element of the wave's top n_out has a key >= that bound.  The bound is found by bisection on the
        // key bits (one ballot + popcount per step, no data movement), the few elements at or above it are compacted
        // into LDS with ballot prefixes, and the workgroup's candidates (about n_out per wave) are ordered by counting,
        // for each candidate, how many are larger -- no k-round arg-max chain anywhere.
        constexpr int KB = sizeof(C) * 4;                           // key bits = the composite's upper half
        typedef typename Comp<C>::key_t K;
        C lmax = 0;
#pragma unroll
        for (int i = 0; i < EPT; ++i) lmax = comp[i] > lmax ? comp[i] : lmax;
        const K lkey = (K)(lmax >> KB);
        K lo = 0, hi = (K)~(K)0;                                    // largest key t with #{lanes: lkey >= t} >= n_out
        if (__builtin_popcountll(__ballot(lmax != 0)) < n_out) {
            hi = 0;                                                 // fewer live lanes than requested: everything qualifies
        } else {
            while (lo < hi) {
                const K mid = lo + ((hi - lo) >> 1) + 1;
                if (__builtin_popcountll(__ballot(lkey >= mid)) >= n_out) lo = mid; else hi = (K)(mid - 1);
            }
        }
        const K tau = hi;
        int base = 0;
        C* seg = s_cand + wave * SAMP_WAVE_CAP;
        seg[lane] = 0;                                              // unused slots read as "nothing" (SAMP_WAVE_CAP == 64)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const bool take = comp[i] != 0 && (K)(comp[i] >> KB) >= tau;
            const unsigned long long b = __ballot(take);
            const int pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (take && pos < SAMP_WAVE_CAP) seg[pos] = comp[i];
            base += __builtin_popcountll(b);
        }
        if (base > SAMP_WAVE_CAP) {                                 // many equal keys: the exact arg-max rounds for this wave
            seg[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            wave_topk<C, EPT>(comp, EPT, n_out, lane, [&](int s2, C win) { seg[s2] = win; });
        }
        __syncthreads();
        // order by counting: thread t owns slot t of the 4 x 64 slots; its output position is the number of larger
        // candidates.  Fixed-length scans with wide LDS reads: no dependent load per candidate.
        const C mine = s_cand[t];
        int rank = 0, live = 0;
        constexpr int VEC = 16 / sizeof(C);
        typedef C cvec __attribute__((ext_vector_type(VEC)));
#pragma unroll 8
        for (int j = 0; j < PART_WAVES * SAMP_WAVE_CAP / VEC; ++j) {
            const cvec v = ((const cvec*)s_cand)[j];                  // same address in every lane: LDS broadcast
#pragma unroll
            for (int e = 0; e < VEC; ++e) { rank += v[e] > mine; live += v[e] != 0; }
        }
        if (mine != 0 && rank < n_out) dst[rank] = mine;
        if (t >= live && t < n_out) dst[t] = 0;                     // fewer live candidates than requested
        return;
    }
    // more than SAMP_FAST_K samples per row: k rounds of arg-max per wave, then wave 0 merges the 4 lists
    wave_topk<C, EPT>(comp, EPT, n_out, lane, [&](int s, C win) { s_cand[wave * SQ_MAX_TOPK + s] = win; });
    __syncthreads();
    if (wave == 0) {
        constexpr int CPL = PART_WAVES * SQ_MAX_TOPK / 64;     // 8
        C c[CPL];
        const int total = PART_WAVES * n_out;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int idx = j * 64 + lane;                     // (wave w, rank s) = (idx / n_out, idx % n_out)
            c[j] = idx < total ? s_cand[(idx / n_out) * SQ_MAX_TOPK + (idx % n_out)] : (C)0;
        }
        wave_topk<C, CPL>(c, (total + 63) >> 6, n_out, lane, [&](int s, C win) { dst[s] = win; });
    }
}

// ---- 3. merge the parts' candidate lists ---------------------------------------------------------------------------
// parts x n_out <= SAMP_RANK_MAX candidates: one 256-thread workgroup per row orders them by counting (a candidate's output
// position is the number of larger candidates); otherwise one wave per row runs k arg-max rounds.
template <typename C>
__global__ void __launch_bounds__(256)
sample_merge_rank_kernel(const C* __restrict__ cand, int parts, int k, int64_t* __restrict__ out,
                         const int32_t* __restrict__ branch, const int32_t* __restrict__ out_off,
                         const int32_t* __restrict__ out_base) {
    __shared__ __attribute__((aligned(16))) C s_all[SAMP_RANK_MAX];
    const int r = blockIdx.x, t = threadIdx.x;
    int n_out = k;
    int64_t* dst = out + (int64_t)r * k;
    if (branch) {
        n_out = branch[r];
        if (n_out > k) n_out = k;
        dst = out + out_off[r];
    }
    if (out_base) dst += *out_base;
    if (n_out <= 0) return;
    const C* src = cand + (size_t)r * parts * k;
    const int total = parts * n_out;
    constexpr int VEC = 16 / sizeof(C);
    typedef C cvec __attribute__((ext_vector_type(VEC)));
    const int padded = (total + 4 * VEC - 1) / (4 * VEC) * (4 * VEC);          // zero padding: "nothing"
    for (int idx = t; idx < padded; idx += 256) s_all[idx] = idx < total ? src[(size_t)(idx / n_out) * k + (idx % n_out)] : (C)0;
    __syncthreads();
    C mine[SAMP_RANK_MAX / 256];
    int rank[SAMP_RANK_MAX / 256], live = 0;
#pragma unroll
    for (int u = 0; u < SAMP_RANK_MAX / 256; ++u) { mine[u] = (t + u * 256 < total) ? s_all[t + u * 256] : (C)0; rank[u] = 0; }
    for (int j = 0; j < padded / VEC; j += 4) {
        cvec v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ((const cvec*)s_all)[j + q];           // same address in every lane: broadcast
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                live += v[q][e] != 0;
#pragma unroll
                for (int u = 0; u < SAMP_RANK_MAX / 256; ++u) rank[u] += v[q][e] > mine[u];
            }
    }
#pragma unroll
    for (int u = 0; u < SAMP_RANK_MAX / 256; ++u)
        if (mine[u] != 0 && rank[u] < n_out) dst[rank[u]] = Comp<C>::id(mine[u]);
    if (t >= live && t < n_out) dst[t] = 0;
}

template <typename C, int CPL>
__global__ void __launch_bounds__(64)
sample_merge_kernel(const C* __restrict__ cand, int parts, int k, int64_t* __restrict__ out,
                    const int32_t* __restrict__ branch, const int32_t* __restrict__ out_off,
                    const int32_t* __restrict__ out_base) {
    const int r = blockIdx.x, lane = threadIdx.x;            // 64 * CPL >= parts * k candidates
    int n_out = k;
    int64_t* dst = out + (int64_t)r * k;
    if (branch) {
        n_out = branch[r];
        if (n_out > k) n_out = k;
        dst = out + out_off[r];
    }
    if (out_base) dst += *out_base;
    if (n_out <= 0) return;
    const C* src = cand + (size_t)r * parts * k;
    const int total = parts * n_out;
    C c[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int idx = j * 64 + lane;
        c[j] = idx < total ? src[(size_t)(idx / n_out) * k + (idx % n_out)] : (C)0;
    }
    wave_topk<C, CPL>(c, (total + 63) >> 6, n_out, lane, [&](int s, C win) { dst[s] = (win == 0) ? 0 : Comp<C>::id(win); });
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int check_rows(const void* logits, int64_t ld, int n_rows, int vocab, int k, const int64_t* out,
                      const int32_t* branch, const int32_t* out_off, const void* workspace) {
    if (!logits || !out || !workspace || n_rows < 0 || vocab <= 0 || k <= 0 || ld < vocab) return SQ_EINVAL;
    if ((branch == nullptr) != (out_off == nullptr)) return SQ_EINVAL;
    if (k > SQ_MAX_TOPK || k > vocab) return SQ_EUNSUPPORTED;
    if ((vocab & 7) || (ld & 7) || ((uintptr_t)logits & 15) || samp_parts(vocab) > SAMP_MAX_PARTS) return SQ_EUNSUPPORTED;
    return SQ_OK;
}

template <int WOR, typename C>
static int launch_sampler(const void* logits, int64_t ld_logits, const void* rnd, int64_t ld_rand, const int32_t* row_ids,
                          int n_rows, int vocab, int k, float temperature, int64_t* out, const int32_t* branch,
                          const int32_t* out_off, const int32_t* out_base, const float* stats, int by_source_row,
                          void* workspace, hipStream_t st) {
    const int parts = samp_parts(vocab);
    C* cand = (C*)((char*)workspace + samp_stats_bytes(n_rows, vocab));
    hipLaunchKernelGGL((sample_parts_kernel<WOR, C>), dim3(n_rows, parts), dim3(PART_THREADS), 0, st, (const half_t*)logits,
                       ld_logits, rnd, ld_rand, row_ids, vocab, k, temperature, stats, by_source_row, branch,
                       cand);
    int rc = sq_check_launch();
    if (rc != SQ_OK) return rc;
    if (parts * k <= SAMP_RANK_MAX && k <= SAMP_FAST_K) {
        hipLaunchKernelGGL((sample_merge_rank_kernel<C>), dim3(n_rows), dim3(256), 0, st, (const C*)cand, parts, k, out, branch,
                           out_off, out_base);
        return sq_check_launch();
    }
#define SQ_MERGE(CPL) hipLaunchKernelGGL((sample_merge_kernel<C, CPL>), dim3(n_rows), dim3(64), 0, st, (const C*)cand, parts, \
                                         k, out, branch, out_off, out_base)
    if (parts * k <= 64 * 4) SQ_MERGE(4);
    else if (parts * k <= 64 * 16) SQ_MERGE(16);
    else SQ_MERGE(SAMP_MAX_PARTS * SQ_MAX_TOPK / 64);
#undef SQ_MERGE
    return sq_check_launch();
}

extern "C" int sq_logits_stats_f16(const void* logits, int64_t ld, const int32_t* d_row_ids, int n_rows, int vocab,
                                   float temperature, float* d_stats, int stats_by_source_row, void* copy_dst,
                                   int64_t ld_dst, void* stream) {
    if (!logits || !d_stats || n_rows < 0 || vocab <= 0 || ld < vocab || !(temperature > 0.f)) return SQ_EINVAL;
    if ((vocab & 7) || (ld & 7) || ((uintptr_t)logits & 15) || samp_parts(vocab) > SAMP_MAX_PARTS) return SQ_EUNSUPPORTED;
    if (copy_dst && (ld_dst < vocab || (ld_dst & 7) || ((uintptr_t)copy_dst & 15))) return SQ_EINVAL;
    if (n_rows == 0) return SQ_OK;
    hipLaunchKernelGGL(logits_stats_kernel, dim3(n_rows, samp_parts(vocab)), dim3(PART_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)logits, ld, d_row_ids, vocab, temperature, d_stats, stats_by_source_row,
                       (half_t*)copy_dst, ld_dst);
    return sq_check_launch();
}

extern "C" int sq_sample_wor_f16(const void* logits, int64_t ld_logits, const void* rnd, int64_t ld_rand,
                                 const int32_t* d_row_ids, int n_rows, int vocab, int k, float temperature,
                                 int64_t* out, const int32_t* d_branch, const int32_t* d_out_off,
                                 const int32_t* d_out_base, const float* d_stats, void* workspace, void* stream) {
    int rc = check_rows(logits, ld_logits, n_rows, vocab, k, out, d_branch, d_out_off, workspace);
    if (rc != SQ_OK) return rc;
    if (!rnd || ld_rand < vocab || (ld_rand & 7) || ((uintptr_t)rnd & 15)) return SQ_EINVAL;
    if (!(temperature > 0.f)) return SQ_EINVAL;
    if (n_rows == 0) return SQ_OK;
    hipStream_t st = (hipStream_t)stream;
    const float* stats = d_stats;
    int by_source = 1;
    if (!stats) {                       // statistics of exactly these rows, indexed by launch row
        rc = sq_logits_stats_f16(logits, ld_logits, d_row_ids, n_rows, vocab, temperature, (float*)workspace, 0, nullptr, 0,
                                 stream);
        if (rc != SQ_OK) return rc;
        stats = (const float*)workspace;
        by_source = 0;
    }
    if (vocab <= 65536)
        return launch_sampler<1, uint32_t>(logits, ld_logits, rnd, ld_rand, d_row_ids, n_rows, vocab, k, temperature, out,
                                              d_branch, d_out_off, d_out_base, stats, by_source, workspace, st);
    return launch_sampler<1, unsigned long long>(logits, ld_logits, rnd, ld_rand, d_row_ids, n_rows, vocab, k, temperature,
                                                    out, d_branch, d_out_off, d_out_base, stats, by_source, workspace, st);
}

extern "C" int sq_topk_f16(const void* logits, int64_t ld_logits, const int32_t* d_row_ids, int n_rows, int vocab,
                           int k, int64_t* out, const int32_t* d_branch, const int32_t* d_out_off,
                           const int32_t* d_out_base, void* workspace, void* stream) {
    int rc = check_rows(logits, ld_logits, n_rows, vocab, k, out, d_branch, d_out_off, workspace);
    if (rc != SQ_OK) return rc;
    if (n_rows == 0) return SQ_OK;
    hipStream_t st = (hipStream_t)stream;
    if (vocab <= 65536)
        return launch_sampler<0, uint32_t>(logits, ld_logits, nullptr, 0, d_row_ids, n_rows, vocab, k, 1.0f, out, d_branch,
                                               d_out_off, d_out_base, nullptr, 0, workspace, st);
    return launch_sampler<0, unsigned long long>(logits, ld_logits, nullptr, 0, d_row_ids, n_rows, vocab, k, 1.0f, out,
                                                     d_branch, d_out_off, d_out_base, nullptr, 0, workspace, st);
}

// The acceptance-rate probe's sampler (SpecTreeTest.collective_grow_static, Tree/SpecTree.py:349-360): the noise is
// fp32 and torch evaluates rand.log() / q in fp32 (q is the fp16 softmax), so the keys are fp32.
extern "C" int sq_sample_wor_f32noise_f16(const void* logits, int64_t ld_logits, const float* rnd, int64_t ld_rand,
                                          const int32_t* d_row_ids, int n_rows, int vocab, int k, float temperature,
                                          int64_t* out, const int32_t* d_branch, const int32_t* d_out_off, void* workspace,
                                          void* stream) {
    int rc = check_rows(logits, ld_logits, n_rows, vocab, k, out, d_branch, d_out_off, workspace);
    if (rc != SQ_OK) return rc;
    if (!rnd || ld_rand < vocab || (ld_rand & 3) || ((uintptr_t)rnd & 15) || !(temperature > 0.f)) return SQ_EINVAL;
    if (n_rows == 0) return SQ_OK;
    rc = sq_logits_stats_f16(logits, ld_logits, d_row_ids, n_rows, vocab, temperature, (float*)workspace, 0, nullptr, 0, stream);
    if (rc != SQ_OK) return rc;
    return launch_sampler<2, unsigned long long>(logits, ld_logits, rnd, ld_rand, d_row_ids, n_rows, vocab, k, temperature, out,
                                                 d_branch, d_out_off, nullptr, (const float*)workspace, 0, workspace,
                                                 (hipStream_t)stream);
}
