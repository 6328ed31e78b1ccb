// a2: draft-expansion samplers.  One 1024-thread workgroup per logits row; the whole row lives in
// registers (EPT halves per thread, 16-byte coalesced loads), so the row is read from HBM exactly
// once: softmax statistics, the Gumbel-style key log(u)/q and the top-k selection all run on the
// register copy.  Rounding points follow the fp16 torch expression of the reference
// (utils.py:10-18): y = h(x/T), q = h(exp(y-max)/sum), lu = h(log u), key = h(lu/q).
//
// Top-k = two-level tournament on 32-bit composites (ordered fp16 key << 16 | 0xffff - token id):
// per-wave DPP arg-max rounds, then one wave merges the 16 candidate lists.  Ties resolve to the
// lower token id.
#include "common.h"

#define SAMP_THREADS 1024
#define SAMP_WAVES (SAMP_THREADS / 64)

// element index of (chunk c, thread t, lane-element j) -- 16-byte chunks interleaved over threads
__device__ __forceinline__ int elem_index(int c, int t, int j) { return (c * SAMP_THREADS + t) * 8 + j; }

template <int EPT, bool WOR>
__global__ void __launch_bounds__(SAMP_THREADS)
sample_rows_kernel(const half_t* __restrict__ logits, int64_t ld_logits, const half_t* __restrict__ rnd,
                   int64_t ld_rand, const int32_t* __restrict__ row_ids, int vocab, int k, float temperature,
                   int64_t* __restrict__ out, const int32_t* __restrict__ branch, const int32_t* __restrict__ out_off) {
    constexpr int CH = EPT / 8;
    __shared__ float s_f[SAMP_WAVES];
    const int t = threadIdx.x;
    const int r = blockIdx.x;
    const int64_t row = row_ids ? (int64_t)row_ids[r] : (int64_t)r;
    const half_t* x = logits + row * ld_logits;

    int n_out = k;
    int64_t* dst = out + (int64_t)r * k;
    if (branch) {
        n_out = branch[r];
        if (n_out > k) n_out = k;
        dst = out + out_off[r];
    }
    if (n_out <= 0) return;   // uniform per block

    half_t key[EPT];          // fp16 keys (WOR) or raw logits (top-k)
    if (WOR) {
        float y[EPT];
        float lmax = -INFINITY;
        // the noise row is fetched together with the logits row: one HBM latency instead of two
        const half_t* u = rnd + row * ld_rand;
        half8 uv[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e0 = elem_index(c, t, 0);
            if (e0 < vocab) uv[c] = *(const half8*)(u + e0);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e0 = elem_index(c, t, 0);
            half8 v;
            if (e0 < vocab) v = *(const half8*)(x + e0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float yy = -INFINITY;
                if (e0 + j < vocab) yy = (float)(half_t)div_rn((float)v[j], temperature);
                y[c * 8 + j] = yy;
                lmax = fmaxf(lmax, yy);
            }
        }
        const float mx = block_max_f32<SAMP_WAVES>(lmax, s_f);
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            y[i] = exp_fast(y[i] - mx);          // exp(-inf) = 0 for padding
            lsum += y[i];
        }
        const float z = block_sum_f32<SAMP_WAVES>(lsum, s_f);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const half_t q = (half_t)div_rn(y[c * 8 + j], z);
                const half_t lu = (half_t)log_fast((float)uv[c][j]);
                // lu / 0 = -inf (lu < 0 always: u < 1); otherwise the correctly rounded quotient
                key[c * 8 + j] = (q == (half_t)0.0f) ? (half_t)(-INFINITY) : (half_t)div_rn((float)lu, (float)q);
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e0 = elem_index(c, t, 0);
            half8 v;
            if (e0 < vocab) v = *(const half8*)(x + e0);
#pragma unroll
            for (int j = 0; j < 8; ++j) key[c * 8 + j] = v[j];
        }
    }

    // ---- top-k: two-level tournament -----------------------------------------------------------
    // composite = ordered fp16 key << 16 | (0xffff - token id): the maximum is the largest key and,
    // inside an exact tie, the lowest token id.  (Token ids need 16 bits: vocab <= 65536; wider
    // vocabularies take the 64-bit path below.)  Level 1: every wave extracts its own top-n_out with
    // DPP arg-max rounds (no barriers); level 2: wave 0 merges the 16 x n_out candidates.
    if (vocab <= 65536) {
        __shared__ uint32_t s_cand[SAMP_WAVES * SQ_MAX_TOPK];
        const int lane = t & 63, wave = t >> 6;
        // composites are built once; a retired element becomes 0 and each round is one
        // compare-select-max sweep over the thread's registers
        uint32_t comp[EPT];
        uint32_t gmax[CH];                               // running maximum of every 8-element chunk
        uint32_t best = 0u;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            uint32_t m = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = elem_index(c, t, j);
                const uint32_t v = (e < vocab) ? ((f16_to_ordered(key[c * 8 + j]) << 16) | (0xffffu - (uint32_t)e)) : 0u;
                comp[c * 8 + j] = v;
                m = v > m ? v : m;
            }
            gmax[c] = m;
            best = m > best ? m : best;
        }
        for (int s = 0; s < n_out; ++s) {
            const uint32_t win = wave_max_u32_dpp(best);  // wave-uniform (SGPR)
            if (lane == 0) s_cand[wave * SQ_MAX_TOPK + s] = win;
            if (win == 0u) continue;
            // the winner's chunk follows from its token id: only that chunk is rescanned (a scalar branch per
            // chunk, the id is wave-uniform), and only in the lane that owns it
            const int wc = (int)(((0xffffu - (win & 0xffffu)) >> 3) / SAMP_THREADS);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (c == wc) {
                    uint32_t m = 0u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t v = (comp[c * 8 + j] == win) ? 0u : comp[c * 8 + j];
                        comp[c * 8 + j] = v;
                        m = v > m ? v : m;
                    }
                    gmax[c] = m;
                }
            }
            best = 0u;
#pragma unroll
            for (int c = 0; c < CH; ++c) best = gmax[c] > best ? gmax[c] : best;
        }
        __syncthreads();
        if (wave == 0) {
            // 16 sorted lists of n_out candidates; lane l owns candidates l, l+64, ...
            constexpr int CPL = SAMP_WAVES * SQ_MAX_TOPK / 64;     // 32
            uint32_t c[CPL];
            const int total = SAMP_WAVES * n_out;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int idx = j * 64 + lane;                     // (wave w, rank s) = (idx / n_out, idx % n_out)
                c[j] = idx < total ? s_cand[(idx / n_out) * SQ_MAX_TOPK + (idx % n_out)] : 0u;
            }
            uint32_t lb = 0u;
#pragma unroll
            for (int j = 0; j < CPL; ++j) lb = c[j] > lb ? c[j] : lb;
            const int used = (total + 63) >> 6;            // candidate registers that hold anything (5 of 32 for k = 19)
            for (int s = 0; s < n_out; ++s) {
                const uint32_t win = wave_max_u32_dpp(lb);
                if (lane == 0) dst[s] = (win == 0u) ? 0 : (int64_t)(0xffffu - (win & 0xffffu));
                if (win == lb && win != 0u) {
                    lb = 0u;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        if (j < used) {
                            if (c[j] == win) c[j] = 0u;
                            lb = c[j] > lb ? c[j] : lb;
                        }
                    }
                }
            }
        }
        return;
    }
    // ---- wide-vocabulary path: k rounds of block-wide arg-max on 64-bit composites ------------------
    unsigned long long best;
    auto local_best = [&](uint32_t removed_lo, uint32_t removed_hi, uint32_t removed_2, uint32_t removed_3) {
        unsigned long long b = 0ull;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const uint32_t word = i < 32 ? removed_lo : (i < 64 ? removed_hi : (i < 96 ? removed_2 : removed_3));
            const bool gone = (word >> (i & 31)) & 1u;
            const int e = elem_index(i >> 3, t, i & 7);
            if (!gone && e < vocab) {
                const unsigned long long comp =
                    ((unsigned long long)(f16_to_ordered(key[i]) + 1u) << 32) | (uint32_t)(0xffffffffu - (uint32_t)e);
                b = comp > b ? comp : b;
            }
        }
        return b;
    };
    uint32_t rm0 = 0, rm1 = 0, rm2 = 0, rm3 = 0;
    best = local_best(rm0, rm1, rm2, rm3);
    __shared__ unsigned long long s_b[SAMP_WAVES];
    for (int s = 0; s < n_out; ++s) {
        const unsigned long long win = block_max_u64<SAMP_WAVES>(best, s_b);
        const uint32_t e = 0xffffffffu - (uint32_t)(win & 0xffffffffu);
        if (t == 0) dst[s] = (win == 0ull) ? 0 : (int64_t)e;
        if (win == best && win != 0ull) {
            const int c = (int)(e >> 3) / SAMP_THREADS;
            const int i = c * 8 + (int)(e & 7);
            if (i < 32) rm0 |= 1u << i;
            else if (i < 64) rm1 |= 1u << (i - 32);
            else if (i < 96) rm2 |= 1u << (i - 64);
            else rm3 |= 1u << (i - 96);
            best = local_best(rm0, rm1, rm2, rm3);
        }
    }
}

template <bool WOR>
static int launch_rows(const void* logits, int64_t ld_logits, const void* rnd, int64_t ld_rand,
                       const int32_t* row_ids, int n_rows, int vocab, int k, float temperature, int64_t* out,
                       const int32_t* branch, const int32_t* out_off, hipStream_t st) {
    dim3 g(n_rows), b(SAMP_THREADS);
#define SQ_LAUNCH(EPT)                                                                                         \
    hipLaunchKernelGGL((sample_rows_kernel<EPT, WOR>), g, b, 0, st, (const half_t*)logits, ld_logits,            \
                       (const half_t*)rnd, ld_rand, row_ids, vocab, k, temperature, out, branch, out_off)
    if (vocab <= 8 * SAMP_THREADS) SQ_LAUNCH(8);
    else if (vocab <= 32 * SAMP_THREADS) SQ_LAUNCH(32);
    else if (vocab <= 128 * SAMP_THREADS) SQ_LAUNCH(128);
    else return SQ_EUNSUPPORTED;
#undef SQ_LAUNCH
    return sq_check_launch();
}

static int check_rows(const void* logits, int64_t ld, int n_rows, int vocab, int k, const int64_t* out,
                      const int32_t* branch, const int32_t* out_off) {
    if (!logits || !out || n_rows < 0 || vocab <= 0 || k <= 0 || ld < vocab) return SQ_EINVAL;
    if ((branch == nullptr) != (out_off == nullptr)) return SQ_EINVAL;
    if (k > SQ_MAX_TOPK || k > vocab) return SQ_EUNSUPPORTED;
    if ((vocab & 7) || (ld & 7) || ((uintptr_t)logits & 15)) return SQ_EUNSUPPORTED;
    return SQ_OK;
}

extern "C" int sq_sample_wor_f16(const void* logits, int64_t ld_logits, const void* rnd, int64_t ld_rand,
                                 const int32_t* d_row_ids, int n_rows, int vocab, int k, float temperature,
                                 int64_t* out, const int32_t* d_branch, const int32_t* d_out_off, void* stream) {
    int rc = check_rows(logits, ld_logits, n_rows, vocab, k, out, d_branch, d_out_off);
    if (rc != SQ_OK) return rc;
    if (!rnd || ld_rand < vocab || (ld_rand & 7) || ((uintptr_t)rnd & 15)) return SQ_EINVAL;
    if (!(temperature > 0.f)) return SQ_EINVAL;
    if (n_rows == 0) return SQ_OK;
    return launch_rows<true>(logits, ld_logits, rnd, ld_rand, d_row_ids, n_rows, vocab, k, temperature, out, d_branch,
                             d_out_off, (hipStream_t)stream);
}

extern "C" int sq_topk_f16(const void* logits, int64_t ld_logits, const int32_t* d_row_ids, int n_rows, int vocab,
                           int k, int64_t* out, const int32_t* d_branch, const int32_t* d_out_off, void* stream) {
    int rc = check_rows(logits, ld_logits, n_rows, vocab, k, out, d_branch, d_out_off);
    if (rc != SQ_OK) return rc;
    if (n_rows == 0) return SQ_OK;
    return launch_rows<false>(logits, ld_logits, nullptr, 0, d_row_ids, n_rows, vocab, k, 1.0f, out, d_branch, d_out_off,
                              (hipStream_t)stream);
}
