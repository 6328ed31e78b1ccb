// a6/a7/a8: verification.  Two launches on the caller's stream:
//   1. verify_nodes: one 1024-thread workgroup per tree node, all nodes in parallel.  Node t reads
//      its target-logits row once and (internal nodes) its draft-logits row once -- the
//      algorithmic byte count (n + n_internal) * V * 2 of SURVEY.md §8(d) -- and evaluates
//      accept_step (Tree/SpecTree.py:136-157) for its own children speculatively: first
//      accepted child, or the residual distribution and the bonus token drawn from it.
//   2. verify_walk: one wave follows node -> accepted child from the root (longest accepted
//      path), applies the reference's side effects along the walked path only, and fills the
//      result record.  No host synchronisation anywhere.
// The fp16 rounding points are those of the reference's fp16 tensor expressions (see
// oracle/ops_np.py); the residual normaliser and the inverse CDF are exact integer sums on the
// 2^-24 grid, so they are independent of reduction order.
#include "common.h"

// vocab is a multiple of 8 (checked by every entry point): a 16-byte chunk is inside the row or outside it as a whole,
// so bounds are tested per chunk (e0 < vocab), never per element
#define VER_THREADS 1024
#define VER_WAVES (VER_THREADS / 64)

struct VerifyWs {
    int32_t* child;   // [n] tree id of the first accepted child, -1 if none
    int32_t* bonus;   // [n] token drawn from the node's final distribution (valid when child == -1)
    int32_t* flag;    // [n] 1 = residual is NaN
    int32_t* nrej;    // [n] number of rejected children (= index of the accepted one)
    int32_t* path;    // [n] tree ids of the accepted path (written by the walker)
    int64_t* tgt;     // [n] greedy: target argmax
};
__host__ __device__ static inline VerifyWs ws_layout(void* ws, int n) {
    VerifyWs w;
    int32_t* p = (int32_t*)ws;
    w.child = p; w.bonus = p + n; w.flag = p + 2 * n; w.nrej = p + 3 * n; w.path = p + 4 * n;
    w.tgt = (int64_t*)(p + 5 * n + (n & 1));
    return w;
}
extern "C" size_t sq_verify_workspace_bytes(int n_tree) {
    if (n_tree <= 0) return 0;
    return (size_t)(5 * n_tree + (n_tree & 1)) * 4 + (size_t)n_tree * 8 + 64;
}

__device__ __forceinline__ int v_elem(int c, int t, int j) { return (c * VER_THREADS + t) * 8 + j; }

// exact round-to-nearest-even of total * 2^-24 to fp16 (total is an exact integer sum)
__device__ __forceinline__ half_t grid_sum_to_f16(unsigned long long total) {
    if (total == 0ull) return (half_t)0.0f;
    const int hb = 63 - __clzll((long long)total);
    if (hb <= 10) return (half_t)((float)(uint32_t)total * 5.9604644775390625e-08f);  // exact
    const int shift = hb - 10;
    unsigned long long q = total >> shift;
    const unsigned long long rem = total & ((1ull << shift) - 1ull);
    const unsigned long long halfway = 1ull << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1ull))) q += 1ull;
    return (half_t)ldexpf((float)(uint32_t)q, shift - 24);
}

// Correctly rounded a / b for a wave-uniform divisor: the refined reciprocal is computed once, every quotient costs one
// multiply and two fused multiply-adds (same result as div_rn for finite a).
// (packed fp32 forms of the quotient -- v_pk_mul_f32 / v_pk_fma_f32 on pairs -- measured slower here: 129 vs 110 us on the
// all-rejected tree)
struct RcpDiv {
    float b, r;
    __device__ __forceinline__ explicit RcpDiv(float b_) : b(b_) {
        const float r0 = __builtin_amdgcn_rcpf(b_);
        r = __builtin_fmaf(__builtin_fmaf(-b_, r0, 1.0f), r0, r0);
    }
    __device__ __forceinline__ float operator()(float a) const {     // a finite
        const float q0 = a * r;
        float q = __builtin_fmaf(__builtin_fmaf(-q0, b, a), r, q0);
        // The quotient must exist as an fp32 value: callers round it to fp16 next, and the reference's result is that
        // DOUBLE rounding (fp32 division, then the cast).  Without the barrier the compiler folds the cast into the
        // last fma (v_fma_mixlo_f16: one rounding of the exact fma result), which differs near fp16 ties.
        asm volatile("" : "+v"(q));
        return q;
    }
};

// softmax(x / T) with the reference's rounding: returns h(exp(y - max) / sum) per element (y = h(x / T)).  The
// exponentials stay in registers between the sum pass and the normalisation (the array is dead afterwards).
__device__ __forceinline__ half8 neg_inf8() {
    const half_t n = (half_t)(-INFINITY);
    return half8{n, n, n, n, n, n, n, n};
}

// fp16 register arrays are held as explicit pairs (one VGPR per two elements); element i of a pair array:
#define H2(arr, i) arr[(i) >> 1][(i) & 1]

template <int EPT>
__device__ __forceinline__ void row_softmax_f16(const half_t* __restrict__ x, int vocab, float temperature, int t,
                                                half2v (&p)[EPT / 2], float* s_f) {
    float e[EPT];
    const RcpDiv div_t(temperature);
    float lmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < EPT / 8; ++c) {
        const int e0 = v_elem(c, t, 0);
        half8 v = neg_inf8();                                   // chunks past the row end read as -inf (branch-free below)
        if (e0 < vocab) v = *(const half8*)(x + e0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = (float)v[j];
            const float qv = div_t(xv);                         // evaluated unconditionally: no branch per element
            const float yy = (float)(half_t)((__builtin_fabsf(xv) < INFINITY) ? qv : xv);   // +-inf / NaN pass through (T > 0)
            e[c * 8 + j] = yy;
            lmax = fmaxf(lmax, yy);
        }
    }
    const float mx = block_max_f32<VER_WAVES>(lmax, s_f);
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) { e[i] = exp_fast(e[i] - mx); lsum += e[i]; }
    const float z = block_sum_f32<VER_WAVES>(lsum, s_f);
    const RcpDiv div_z(z);
#pragma unroll
    for (int i = 0; i < EPT; ++i) H2(p, i) = (half_t)div_z(e[i]);
}

// Exact inverse CDF of a (possibly unnormalised) fp16 distribution held EPT elements per thread: the token whose
// cumulative mass interval, in element-index order (chunk, thread, j) and in exact integer arithmetic on the 2^-24
// grid, contains u24 / 2^24 of the total.  Returns -1 for an all-zero distribution.  Block-uniform result.
template <int EPT>
__device__ __forceinline__ int block_inverse_cdf(half2v (&p)[EPT / 2], int vocab, int t, uint32_t u24) {
    constexpr int CH = EPT / 8;
    __shared__ unsigned long long s_ct[VER_WAVES][EPT / 8];
    __shared__ unsigned long long s_scan[VER_WAVES];
    __shared__ int s_pick;
    uint32_t csum[CH];                   // per-thread sum of one 8-element chunk: <= 8 * 2^24
    unsigned long long total = 0ull, base = 0ull;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        uint32_t sv = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (v_elem(c, t, 0) < vocab) sv += (uint32_t)((float)H2(p, c * 8 + j) * 16777216.0f);
        csum[c] = sv;
        const unsigned long long wsum = wave_sum_u32_wide_dpp(sv);
        if ((t & 63) == 0) s_ct[t >> 6][c] = wsum;
    }
    __syncthreads();
    // two passes over the 16 x CH wave sums in LDS (broadcast reads) instead of a register array of chunk totals: the
    // 64-bit totals were what spilled at the 128-VGPR cap of a 1024-thread workgroup
    auto chunk_total = [&](int c) {
        unsigned long long acc = 0ull;
#pragma unroll
        for (int w2 = 0; w2 < VER_WAVES; ++w2) acc += s_ct[w2][c];
        return acc;
    };
#pragma unroll
    for (int c = 0; c < CH; ++c) total += chunk_total(c);
    if (total == 0ull) { __syncthreads(); return -1; }
    const unsigned long long thr = (__umul64hi((unsigned long long)u24, total) << 40) |
                                   (((unsigned long long)u24 * total) >> 24);
    int cstar = CH - 1;
    unsigned long long run = 0ull;
    bool found = false;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const unsigned long long ct = chunk_total(c);
        if (!found && run + ct > thr) { cstar = c; base = run; found = true; }
        run += ct;
    }
    unsigned long long mine = 0ull;
#pragma unroll
    for (int c = 0; c < CH; ++c) if (c == cstar) mine = (unsigned long long)csum[c];
    // inclusive scan of `mine` over the block's threads
    unsigned long long inc = mine;
    const int lane = t & 63, w = t >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, o, 64);
        uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), o, 64);
        if (lane >= o) inc += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63) s_scan[w] = inc;
    if (t == 0) s_pick = -1;
    __syncthreads();
    unsigned long long woff = 0ull;
    for (int i = 0; i < w; ++i) woff += s_scan[i];
    const unsigned long long excl = base + woff + inc - mine;
    if (mine > 0ull && excl <= thr && thr < excl + mine) {
        unsigned long long acc = excl;
        int pick = -1;
        // the one thread that holds the quantile converts its chunk again: without the barrier the compiler keeps all EPT
        // grid values of the first pass alive for this loop (32 registers -> spills at the 128-VGPR cap)
#pragma unroll
        for (int k2 = 0; k2 < EPT / 2; ++k2) asm volatile("" : "+v"(p[k2]));
        int tt = t;                                              // likewise the element ids: recomputed, not kept
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c == cstar) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc += (unsigned long long)(uint32_t)((float)H2(p, c * 8 + j) * 16777216.0f);
                    if (pick < 0 && acc > thr) pick = v_elem(c, tt, j);
                }
            }
        }
        s_pick = pick;
    }
    __syncthreads();
    const int res = s_pick;
    __syncthreads();                       // s_pick / s_ct are reused by the next call
    return res;
}

// one block-wide reduction of three quantities at once: exact integer sum, float max, float sum
struct Red3 { unsigned long long isum; float fmax; float fsum; };
__device__ __forceinline__ Red3 block_red3(uint32_t isum32, float fmx, float fsum, unsigned long long* s_u,
                                           float* s_m, float* s_s) {
    unsigned long long isum = wave_sum_u32_wide_dpp(isum32);     // per-thread partial <= EPT * 2^24 < 2^32
    fmx = wave_max_f32_dpp(fmx);
    fsum = wave_sum_f32_dpp(fsum);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { s_u[w] = isum; s_m[w] = fmx; s_s[w] = fsum; }
    __syncthreads();
    Red3 r;
    r.isum = 0ull; r.fmax = -INFINITY; r.fsum = 0.f;
#pragma unroll
    for (int i = 0; i < VER_WAVES; ++i) {        // fixed order => deterministic
        r.isum += s_u[i]; r.fmax = fmaxf(r.fmax, s_m[i]); r.fsum += s_s[i];
    }
    __syncthreads();
    return r;
}

// Device-resident step state read by the device-driven speculation step (include/sequoia_hip.h, SQ_STEP_*).
__device__ __forceinline__ int step_gt(const int32_t* d_step, int gt) { return d_step ? d_step[SQ_STEP_GT] : gt; }

// REPLACE = SpecInfer's rule (Tree/SpecInferTree.py:141-164): children were drawn WITH replacement, so a rejected
// token stays in q (no masking, no renormalisation of q) and the test is p >= r q.
//
// Per thread: the target distribution p (fp16; after a rejection the UNNORMALISED residual d = relu(p - q), its
// normaliser `sf` is applied on the fly by the next pass), the draft row y = h(x / T) (fp16, rejected tokens -inf) and
// its exponentials e = exp(y - mx) (fp32, computed once per maximum).  One rejection = one pass over the registers
// (p <- h(d / sf); q <- h(e / z); d <- relu(h(p - q)); exact integer sum of d; fresh sum / maximum of the draft without the
// rejected token) and one block reduction.
// RULE 2 = the acceptance-rate probe (SpecTreeTest.accept_step, Tree/SpecTree.py:396-417): r is fp32, torch promotes
// r * q[token] and the comparison to fp32, and the test is p >= r q; rejections mask q like RULE 0.
template <int EPT, int RULE>
__global__ void __launch_bounds__(VER_THREADS)
verify_nodes_kernel(const half_t* __restrict__ target_logits, const half_t* __restrict__ draft_logits,
                    const int64_t* __restrict__ tokens, const void* __restrict__ r_raw,
                    const int32_t* __restrict__ child_off, const int32_t* __restrict__ child_ids, int n_tree,
                    int vocab, int gt_arg, float temperature, uint32_t u24_arg, void* ws_raw,
                    const int32_t* __restrict__ d_step, const uint32_t* __restrict__ d_bonus, int n_bonus) {
    constexpr int CH = EPT / 8;
    constexpr bool REPLACE = RULE == 1;
    __shared__ float s_f[VER_WAVES];
    __shared__ float s_m[VER_WAVES];
    __shared__ float s_s[VER_WAVES];
    __shared__ unsigned long long s_u[VER_WAVES];
    __shared__ float s_tok[2];            // e[tok], p[tok] of the child under test
    __shared__ floatx4 s_e4[(EPT / 4) * VER_THREADS];     // the draft row's exponentials (128 KB at V <= 32768)
    const int t = threadIdx.x;
    const int node = blockIdx.x;
    const VerifyWs ws = ws_layout(ws_raw, n_tree);
    const int gt = step_gt(d_step, gt_arg);
    const uint32_t u24 = (d_step && d_bonus) ? (d_bonus[(uint32_t)d_step[SQ_STEP_INDEX] % (uint32_t)n_bonus] & 0xffffffu) : u24_arg;

    half2v p[EPT / 2];
    row_softmax_f16<EPT>(target_logits + (size_t)node * vocab, vocab, temperature, t, p, s_f);

    const int c0 = child_off[node], nc = child_off[node + 1] - c0;
    int accepted = -1, nrej = 0, nan_flag = 0;
    float sf = 1.0f;                      // normaliser of p: 1 = p holds probabilities, else p holds relu(p - q) and sf its sum
    bool scaled = false;
    if (nc > 0) {
        // draft side: e = exp(y - mx) with y = h(x / T), kept in LDS ([EPT / 4][thread] float4 columns: thread-private,
        // conflict-free 16-byte accesses) -- 32 fewer live registers than a register array, and the rejected token's entry
        // is addressed directly instead of through a 32-way select.  A rejected token gets e = 0 and its bit in `masked`
        // (the row is re-read from L2 only when the maximum itself is rejected and the exponentials are rebased).
        uint32_t masked = 0u;
        const half_t* xd = draft_logits + (size_t)node * vocab;
        const RcpDiv div_t(temperature);
        half8 xrow[CH];                    // the draft row's chunks of this thread, (re)loaded by fetch_row()
        auto fetch_row = [&]() {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int e0 = v_elem(c, t, 0);
                xrow[c] = neg_inf8();
                if (e0 < vocab) xrow[c] = *(const half8*)(xd + e0);
            }
        };
        float mx, z;
        auto rebase = [&]() {             // e <- exp(y - max y), z <- sum e; y = h(x / T) replaces x in xrow (fp16: no fp32 copy of the row)
            float lmax = -INFINITY;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xv = (float)xrow[c][j];
                    const float qv = div_t(xv);
                    half_t yy = (half_t)((__builtin_fabsf(xv) < INFINITY) ? qv : xv);   // +-inf / NaN pass through (T > 0)
                    yy = ((masked >> (c * 8 + j)) & 1u) ? (half_t)(-INFINITY) : yy;
                    xrow[c][j] = yy;
                    lmax = fmaxf(lmax, (float)yy);
                }
            }
            mx = block_max_f32<VER_WAVES>(lmax, s_f);
            float lsum = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < EPT / 4; ++k4) {
                floatx4 ev;
#pragma unroll
                for (int h4 = 0; h4 < 4; ++h4) {
                    const int i = 4 * k4 + h4;
                    ev[h4] = exp_fast((float)xrow[i >> 3][i & 7] - mx);
                    lsum += ev[h4];
                }
                s_e4[k4 * VER_THREADS + t] = ev;
            }
            z = block_sum_f32<VER_WAVES>(lsum, s_f);
        };
        fetch_row();
        rebase();

        for (int jc = 0; jc < nc; ++jc) {
            const int child = child_ids[c0 + jc];
            const int slot = child + gt - 1;
            const int tok = (int)tokens[slot];
            const float rr = (RULE == 2) ? ((const float*)r_raw)[slot] : (float)((const half_t*)r_raw)[slot];
            // element tok lives in thread (tok/8) % THREADS at register (tok/8/THREADS)*8 + tok%8:
            // a scalar index compare instead of 32 per-thread element ids held in registers
            const bool tok_ok = tok >= 0 && tok < vocab;
            const bool mine = tok_ok && t == ((tok >> 3) & (VER_THREADS - 1));
            const int tok_local = tok_ok ? ((tok >> 3) / VER_THREADS) * 8 + (tok & 7) : 0;
            float* e_tok_ptr = (float*)&s_e4[(tok_local >> 2) * VER_THREADS + t] + (tok_local & 3);
            const RcpDiv div_sf(sf), div_z(z);
            // broadcast e[tok], p[tok] from the owning thread
            if (mine) s_tok[0] = *e_tok_ptr;
            float p_sel = 0.f;                                   // register tok_local of the owner: a select chain, one quotient
#pragma unroll
            for (int i = 0; i < EPT; ++i) p_sel = (i == tok_local) ? (float)H2(p, i) : p_sel;
            const float p_div = div_sf(p_sel);
            if (mine) s_tok[1] = scaled ? (float)(half_t)p_div : p_sel;
            __syncthreads();
            const float e_tok = s_tok[0];
            const half_t q_tok = (half_t)div_z(e_tok);
            const half_t p_tok = (half_t)s_tok[1];
            bool ok;
            if (RULE == 2) {
                ok = (float)p_tok >= rr * (float)q_tok;                      // fp32 product and comparison (:412)
            } else {
                const half_t rq = (half_t)(rr * (float)q_tok);
                ok = REPLACE ? (p_tok >= rq) : (p_tok > rq);                  // Tree/SpecTree.py:152 (strict)
            }
            ok = ok && tok_ok;
            if (ok) { accepted = child; break; }
            // the rejected token sat at the maximum (exp(0) is exactly 1): the exponentials will be rebased after this
            // rejection.  (SQ_VERIFY_PREFETCH_ROW issues the row's re-read from L2 here, under the residual pass: the 16
            // registers it holds across the pass push the kernel over its 128-VGPR cap -- 17 spilled -- so it is off.)
            const bool will_rebase = !REPLACE && e_tok == 1.0f;
#ifdef SQ_VERIFY_PREFETCH_ROW
            if (will_rebase) fetch_row();
#endif
            // reject: p <- relu(p - q) / sum(relu(p - q));  draft_logits[tok] <- -65504 (=> q[tok] = 0)
            uint32_t lint = 0u;
            float nsum = 0.f;
            const half2v zero2 = {(half_t)0.0f, (half_t)0.0f};
#pragma unroll
            for (int k4 = 0; k4 < EPT / 4; ++k4) {
                // at most two LDS columns in flight: hoisting all EPT / 4 reads costs 32 registers that the prefetched
                // draft row needs (4 waves per SIMD hide the LDS latency)
                if ((k4 & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                const floatx4 ev = s_e4[k4 * VER_THREADS + t];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int k2 = 2 * k4 + h2;
                    half2v pi = p[k2];
                    // p / sf unconditionally: before the first rejection sf = 1 and the correctly rounded quotient is p itself
                    // (a test of `scaled` here compiled to one exec-mask branch per register pair)
                    pi[0] = (half_t)div_sf((float)pi[0]); pi[1] = (half_t)div_sf((float)pi[1]);
                    half2v q;
                    q[0] = (half_t)div_z(ev[2 * h2]); q[1] = (half_t)div_z(ev[2 * h2 + 1]);
                    // relu_(p - q) in packed fp16 (one rounding, like the reference's fp16 tensor op); a NaN difference
                    // becomes 0 (max returns the number), as `d > 0 ? d : 0` did
                    const half2v di = __builtin_elementwise_max(pi - q, zero2);
                    p[k2] = di;                                  // p now holds the unnormalised residual
                    lint += (uint32_t)((float)di[0] * 16777216.0f) + (uint32_t)((float)di[1] * 16777216.0f);
#pragma unroll
                    for (int h1 = 0; h1 < 2; ++h1) {             // fresh sum of the draft without the rejected token
                        const int i = 2 * k2 + h1;
                        nsum += (!REPLACE && mine && i == tok_local) ? 0.f : ev[2 * h2 + h1];
                    }
                }
            }
            if (!REPLACE && mine) { *e_tok_ptr = 0.f; masked |= 1u << tok_local; }
            const Red3 red = block_red3(lint, 0.f, nsum, s_u, s_m, s_s);   // also orders the s_tok reads above
            if (red.isum == 0ull) nan_flag = 1;                  // 0/0 -> NaN residual (utils.py:7)
            sf = (float)grid_sum_to_f16(red.isum);
            scaled = true;
            nrej = jc + 1;
            if (nan_flag) break;          // every later comparison with NaN is false: all rejected
            if (REPLACE) continue;        // q is unchanged
            z = red.fsum;
#ifdef SQ_VERIFY_PREFETCH_ROW
            if (will_rebase) rebase();
#else
            if (will_rebase) { fetch_row(); rebase(); }
#endif
        }
        if (nan_flag) nrej = nc;
    }

    int bonus = -1;
    if (accepted < 0 && !nan_flag) {
        if (scaled) {                     // materialise the normalised residual for the draw
            const RcpDiv div_sf(sf);
#pragma unroll
            for (int i = 0; i < EPT; ++i) H2(p, i) = (half_t)div_sf((float)H2(p, i));
        }
        bonus = block_inverse_cdf<EPT>(p, vocab, t, u24);
        if (bonus < 0) nan_flag = 1;      // empty distribution: treated like the NaN residual
    }
    if (t == 0) {
        ws.child[node] = accepted;
        ws.bonus[node] = bonus;
        ws.flag[node] = nan_flag;
        ws.nrej[node] = nrej;
    }
}

// One wave.  Walk root -> accepted children; side effects of Tree/SpecTree.py:156,222,224.
// Everything the walk reads (per-node verdicts, the tree's tokens, the greedy targets, the children lists) is first loaded
// into LDS by all 64 lanes at once -- the walk itself (lane 0) then touches no global memory, where every step used to be a
// dependent round trip -- and the side effects (the -65504 writes of the rejected children, the token compaction, the
// result record and its ring copy) are issued by the lanes in parallel.
#define WALK_MAX_CHILD_IDS (2 * SQ_MAX_TREE)
__global__ void __launch_bounds__(64) verify_walk_kernel(half_t* draft_logits, int64_t* tokens, int token_capacity,
                                   const int32_t* __restrict__ child_off, const int32_t* __restrict__ child_ids,
                                   int n_tree, int vocab, int gt_arg, void* ws_raw, int32_t* result, int mode,
                                   const int64_t* __restrict__ tgt_tokens, int32_t* d_step, int32_t* d_ring) {
    // mode 0: stochastic (Sequoia), 1: token equality against tgt (greedy argmax or caller-supplied samples),
    //      2: stochastic without the draft-logit side effects (SpecInfer)
    const VerifyWs ws = ws_layout(ws_raw, n_tree);
    const int lane = threadIdx.x;
    const int gt = step_gt(d_step, gt_arg);
    __shared__ int32_t s_rec[SQ_RESULT_INTS];
    if (d_step && d_step[SQ_STEP_ACTIVE] == 0) {
        // A step that was enqueued behind a terminal one (device-driven loop, steps in flight): it commits nothing.
        // tokens[0, gt) -- the finished text -- and the compacted KV rows stay as the terminal step left them (its gt
        // is the terminal step's accept length, so everything this step's samplers and forwards wrote lies beyond the
        // text); the record says "skipped" and the step block does not move.
        int v = 0;
        if (lane == SQ_RES_ACCEPT_LEN || lane == SQ_RES_GT) v = gt;
        if (lane == SQ_RES_BONUS) v = -1;
        if (lane == SQ_RES_TERMINAL) v = 1;
        if (lane == SQ_RES_REASON) v = SQ_REASON_SKIPPED;
        if (lane == 7) v = d_step[SQ_STEP_INDEX];
        result[lane] = v;
        if (lane == 0) d_step[SQ_STEP_NEXT_GT] = gt;
        if (d_ring) {
            int32_t* slot = d_ring + ((uint32_t)d_step[SQ_STEP_INDEX] % SQ_RESULT_RING) * SQ_RESULT_INTS;
            if (lane != 7) slot[lane] = v;
            __threadfence_system();
            __syncthreads();
            if (lane == 0) __hip_atomic_store(slot + 7, d_step[SQ_STEP_INDEX], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int greedy = (mode & 3) == 1;
    const int gather_first = greedy || (mode & 4);
    mode &= 3;
    const int64_t* tgt = tgt_tokens ? tgt_tokens : ws.tgt;

    // ---- stage the walk's inputs in LDS --------------------------------------------------------------------------
    __shared__ int32_t s_child[SQ_MAX_TREE], s_nrej[SQ_MAX_TREE], s_flag[SQ_MAX_TREE], s_bonus[SQ_MAX_TREE];
    __shared__ int32_t s_off[SQ_MAX_TREE + 1], s_ids[WALK_MAX_CHILD_IDS];
    __shared__ int64_t s_tok[SQ_MAX_TREE], s_tgt[SQ_MAX_TREE];
    __shared__ int32_t s_path[SQ_MAX_TREE], s_walked[SQ_MAX_TREE];       // accepted nodes; every node the walk visited
    __shared__ int32_t s_out[8];                                           // n_acc, terminal, reason, bonus, last node, n_walked
    const int n_ids = child_off[n_tree];
    const bool ids_in_lds = n_ids <= WALK_MAX_CHILD_IDS;
    for (int t = lane; t < n_tree; t += 64) {
        if (!greedy) { s_child[t] = ws.child[t]; s_nrej[t] = ws.nrej[t]; s_flag[t] = ws.flag[t]; s_bonus[t] = ws.bonus[t]; }
        else s_tgt[t] = tgt[t];
        s_tok[t] = tokens[t + gt - 1];
        s_off[t] = child_off[t];
    }
    if (lane == 0) s_off[n_tree] = n_ids;
    if (ids_in_lds)
        for (int i = lane; i < n_ids; i += 64) s_ids[i] = child_ids[i];
    __syncthreads();
    auto cid = [&](int i) { return ids_in_lds ? s_ids[i] : child_ids[i]; };

    // ---- the walk (lane 0, LDS only) -------------------------------------------------------------------------------
    if (lane == 0) {
        int node = 0, n_acc = 0, terminal = 0, reason = 0, n_walked = 0;
        for (int guard = 0; guard < n_tree; ++guard) {
            int child = -1;
            const int c0 = s_off[node], nc = s_off[node + 1] - c0;
            s_walked[n_walked++] = node;
            if (greedy) {
                const int64_t want = s_tgt[node];
                for (int j = 0; j < nc; ++j) {
                    const int c = cid(c0 + j);
                    if (s_tok[c] == want) { child = c; break; }
                }
            } else {
                child = s_child[node];
            }
            if (child < 0) break;
            node = child;
            s_path[n_acc++] = node;
            const int64_t tk = s_tok[node];
            if (tk == 0 || tk == 2) { terminal = 1; reason = 1; break; }   // Tree/SpecTree.py:208
        }
        int bonus = -1;
        if (!terminal) {
            if (greedy) bonus = (int)s_tgt[node];
            else if (s_flag[node]) { terminal = 1; reason = 2; }            // isnan(residual), :219
            else bonus = s_bonus[node];
        }
        const int a = gt + n_acc;
        if (!terminal && token_capacity > 0 && a >= token_capacity) {
            // no slot for the bonus token (a fully accepted chain that ends at max_length): the reference raises
            // IndexError at `self.tokens[accept_length] = ...` (Tree/SpecTree.py:222); here the step becomes terminal
            terminal = 1; reason = 3; bonus = -1;
        }
        s_out[0] = n_acc; s_out[1] = terminal; s_out[2] = reason; s_out[3] = bonus; s_out[4] = node; s_out[5] = n_walked;
    }
    __syncthreads();
    const int n_acc = s_out[0], terminal = s_out[1], reason = s_out[2], bonus = s_out[3], last = s_out[4], n_walked = s_out[5];
    const int a = gt + n_acc;

    // ---- side effects, lanes in parallel ---------------------------------------------------------------------------------
    // draft_logits[node][token of every rejected child] = finfo(fp16).min for every walked node (:156)
    if (mode == 0 && !greedy) {
        for (int wi = 0; wi < n_walked; ++wi) {
            const int node = s_walked[wi];
            const int c0 = s_off[node], nc = s_off[node + 1] - c0;
            const int nr = min(s_nrej[node], nc);
            for (int j = lane; j < nr; j += 64) {
                const int64_t tok = s_tok[cid(c0 + j)];
                if (tok >= 0 && tok < vocab) draft_logits[(size_t)node * vocab + tok] = (half_t)(-65504.0f);
            }
        }
    }
    // tokens[:a] = tokens[accept_list] and the bonus token at slot a.  Order of the two writes follows the reference:
    // SpecTree / SpecInferTree store the bonus token at slot a BEFORE the gather (Tree/SpecTree.py:222-224), so an accepted
    // node that happens to sit at slot a (tree node n_acc + 1 on the accepted path, e.g. a fully accepted 8x8 tree) is
    // committed with the bonus token's id; GreedyTree / GreedySTree gather first (Tree/GreedyTree.py:204-206).  Reproduced
    // for token parity; callers that want the lossless order pass SQ_VERIFY_GATHER_FIRST with the bonus uniform.  The
    // gather reads the staged copies (slots ascending, dst <= src: the sequential in-place move reads original values too).
    if (!gather_first && !terminal && lane == 0 && n_acc + 1 < n_tree) s_tok[n_acc + 1] = (int64_t)bonus;   // slot a == node n_acc + 1
    __syncthreads();
    for (int j = lane; j < n_acc; j += 64) tokens[gt + j] = s_tok[s_path[j]];
    if (!terminal && lane == 0) tokens[a] = bonus;
    for (int j = lane; j < n_acc; j += 64) {
        ws.path[j] = s_path[j];
        result[SQ_RESULT_INTS + j] = s_path[j] + gt - 1;                    // full list (chains deeper than the header)
    }
    {
        int v = 0;
        if (lane == SQ_RES_ACCEPT_LEN) v = a;
        if (lane == SQ_RES_N_TREE) v = n_acc;
        if (lane == SQ_RES_BONUS) v = bonus;
        if (lane == SQ_RES_TERMINAL) v = terminal;
        if (lane == SQ_RES_REASON) v = reason;
        if (lane == SQ_RES_GT) v = gt;
        if (lane == SQ_RES_LAST_NODE) v = last;
        if (lane == 7) v = d_step ? d_step[SQ_STEP_INDEX] : 0;
        if (lane >= SQ_RES_SLOTS && lane - SQ_RES_SLOTS < n_acc) v = s_path[lane - SQ_RES_SLOTS] + gt - 1;
        s_rec[lane] = v;
        result[lane] = v;
    }
    if (d_step) {
        const int index = d_step[SQ_STEP_INDEX];
        __syncthreads();
        if (lane == 0) {
            // device-driven step: the next step starts at new_gt = a + 1 (the bonus token is committed at slot a).
            // A terminal step leaves gt = a: steps already in flight behind it then work beyond the finished text
            // (tokens[0, a) and the KV rows compacted to [gt, a) are not touched again) and skip their commit (above).
            int next = terminal ? a : a + 1;
            if (terminal && token_capacity > 0 && next + n_tree - 1 > token_capacity) next = gt;     // keep a follower in bounds
            d_step[SQ_STEP_NEXT_GT] = next;
            if (terminal) d_step[SQ_STEP_ACTIVE] = 0;
        }
        if (d_ring) {
            // copy of the header for the host, one slot per step.  The ring may live in pinned host memory (the host then
            // polls it instead of waiting on an event): everything but the step-index word first, a system-scope fence,
            // then the index word -- a reader that sees the index sees the record.
            int32_t* slot = d_ring + ((uint32_t)index % SQ_RESULT_RING) * SQ_RESULT_INTS;
            if (lane != 7) slot[lane] = s_rec[lane];
            __threadfence_system();
            __syncthreads();
            if (lane == 0) __hip_atomic_store(slot + 7, index, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct StepArgs {                       // optional device-driven extras of the verify entry points
    int token_capacity;
    int32_t* d_step;
    const uint32_t* d_bonus;
    int n_bonus;
    int32_t* d_ring;
};

static int verify_stochastic_impl(const void* target_logits, void* draft_logits, int64_t* tokens, const void* r,
                                  const int32_t* d_child_off, const int32_t* d_child_ids, int n_tree, int vocab, int gt,
                                  float temperature, uint32_t bonus_u24, void* workspace, int32_t* d_result, void* stream,
                                  int rule, const StepArgs& sa) {
    const bool replace = rule == 1;
    if (!target_logits || !draft_logits || !tokens || !r || !d_child_off || !workspace || !d_result) return SQ_EINVAL;
    if (n_tree <= 0 || n_tree > SQ_MAX_TREE || vocab <= 0 || (!sa.d_step && gt < 1) || !(temperature > 0.f)) return SQ_EINVAL;
    if (n_tree > 1 && !d_child_ids) return SQ_EINVAL;
    if (sa.d_bonus && sa.n_bonus <= 0) return SQ_EINVAL;
    if ((vocab & 7) || ((uintptr_t)target_logits & 15) || ((uintptr_t)draft_logits & 15)) return SQ_EUNSUPPORTED;
    const int gather_first = (bonus_u24 & SQ_VERIFY_GATHER_FIRST) ? 4 : 0;
    bonus_u24 &= ~SQ_VERIFY_GATHER_FIRST;
    if (bonus_u24 >= (1u << 24)) return SQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(n_tree), b(VER_THREADS);
#define SQ_LAUNCH(EPT, REP)                                                                                     \
    hipLaunchKernelGGL((verify_nodes_kernel<EPT, REP>), g, b, 0, st, (const half_t*)target_logits,               \
                       (const half_t*)draft_logits, (const int64_t*)tokens, (const void*)r, d_child_off,          \
                       d_child_ids, n_tree, vocab, gt, temperature, bonus_u24, workspace, (const int32_t*)sa.d_step, \
                       sa.d_bonus, sa.n_bonus)
    if (vocab <= 8 * VER_THREADS) { if (rule == 1) SQ_LAUNCH(8, 1); else if (rule == 2) SQ_LAUNCH(8, 2); else SQ_LAUNCH(8, 0); }
    else if (vocab <= 32 * VER_THREADS) { if (rule == 1) SQ_LAUNCH(32, 1); else if (rule == 2) SQ_LAUNCH(32, 2); else SQ_LAUNCH(32, 0); }
    else return SQ_EUNSUPPORTED;
#undef SQ_LAUNCH
    int rc = sq_check_launch();
    if (rc != SQ_OK) return rc;
    hipLaunchKernelGGL(verify_walk_kernel, dim3(1), dim3(64), 0, st, (half_t*)draft_logits, tokens, sa.token_capacity,
                       d_child_off, d_child_ids, n_tree, vocab, gt, workspace, d_result, (replace ? 2 : 0) | gather_first,
                       (const int64_t*)nullptr, sa.d_step, sa.d_ring);
    return sq_check_launch();
}

extern "C" int sq_verify_stochastic_f16(const void* target_logits, void* draft_logits, int64_t* tokens, int token_capacity,
                                        const void* r, const int32_t* d_child_off, const int32_t* d_child_ids, int n_tree,
                                        int vocab, int gt, float temperature, uint32_t bonus_u24, void* workspace,
                                        int32_t* d_result, int32_t* d_step, const uint32_t* d_bonus_u24, int n_bonus,
                                        int32_t* d_result_ring, void* stream) {
    const StepArgs sa{token_capacity, d_step, d_bonus_u24, n_bonus, d_result_ring};
    return verify_stochastic_impl(target_logits, draft_logits, tokens, r, d_child_off, d_child_ids, n_tree, vocab, gt,
                                  temperature, bonus_u24, workspace, d_result, stream, 0, sa);
}

extern "C" int sq_verify_specinfer_f16(const void* target_logits, const void* draft_logits, int64_t* tokens,
                                       int token_capacity, const void* r, const int32_t* d_child_off,
                                       const int32_t* d_child_ids, int n_tree, int vocab, int gt, float temperature,
                                       uint32_t bonus_u24, void* workspace, int32_t* d_result, void* stream) {
    const StepArgs sa{token_capacity, nullptr, nullptr, 0, nullptr};
    return verify_stochastic_impl(target_logits, (void*)draft_logits, tokens, r, d_child_off, d_child_ids, n_tree, vocab,
                                  gt, temperature, bonus_u24, workspace, d_result, stream, 1, sa);
}

// SpecTreeTest.verify (Tree/SpecTree.py:396-481), the acceptance-rate probe of tests/test_accept.py: Sequoia's walk with r in
// fp32 and the test p >= r q evaluated in fp32.  The walker gathers first (the probe concatenates the accepted tokens and the
// bonus token, :472-474): pass SQ_VERIFY_GATHER_FIRST.
extern "C" int sq_verify_probe_f16(const void* target_logits, void* draft_logits, int64_t* tokens, int token_capacity,
                                   const float* r32, const int32_t* d_child_off, const int32_t* d_child_ids, int n_tree,
                                   int vocab, int gt, float temperature, uint32_t bonus_u24, void* workspace,
                                   int32_t* d_result, void* stream) {
    const StepArgs sa{token_capacity, nullptr, nullptr, 0, nullptr};
    return verify_stochastic_impl(target_logits, draft_logits, tokens, r32, d_child_off, d_child_ids, n_tree, vocab, gt,
                                  temperature, bonus_u24, workspace, d_result, stream, 2, sa);
}

// ---- i.i.d. draws from softmax(logits / T) (SpecInfer's draft expansion, Tree/SpecInferTree.py:104-109) ------------
// One workgroup per row; draw j of row i is the exact inverse CDF at u24[i][j] / 2^24 (explicit uniforms replace
// torch's device multinomial stream).  Output placement like sq_sample_wor_f16: branch / out_off gather.
template <int EPT>
__global__ void __launch_bounds__(VER_THREADS)
sample_iid_kernel(const half_t* __restrict__ logits, int64_t ld, const int32_t* __restrict__ row_ids, int vocab, int k,
                  float temperature, const uint32_t* __restrict__ u24, int64_t* __restrict__ out,
                  const int32_t* __restrict__ branch, const int32_t* __restrict__ out_off) {
    __shared__ float s_f[VER_WAVES];
    const int t = threadIdx.x, row = blockIdx.x;
    const int src = row_ids ? row_ids[row] : row;
    half2v p[EPT / 2];
    row_softmax_f16<EPT>(logits + (size_t)src * ld, vocab, temperature, t, p, s_f);
    const int take = branch ? branch[row] : k;
    const int64_t base = out_off ? (int64_t)out_off[row] : (int64_t)row * k;
    for (int j = 0; j < take; ++j) {
        const int tok = block_inverse_cdf<EPT>(p, vocab, t, u24[(size_t)row * k + j] & 0xffffffu);
        if (t == 0) out[base + j] = tok;
    }
}

extern "C" int sq_sample_iid_f16(const void* logits, int64_t ld, const int32_t* d_row_ids, int n_rows, int vocab, int k,
                                 float temperature, const uint32_t* d_u24, int64_t* d_out, const int32_t* d_branch,
                                 const int32_t* d_out_off, void* stream) {
    if (!logits || !d_u24 || !d_out || n_rows < 0 || vocab <= 0 || k <= 0 || ld < vocab || !(temperature > 0.f)) return SQ_EINVAL;
    if ((vocab & 7) || (ld & 7) || ((uintptr_t)logits & 15)) return SQ_EUNSUPPORTED;
    if (n_rows == 0) return SQ_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(n_rows), b(VER_THREADS);
    if (vocab <= 8 * VER_THREADS)
        hipLaunchKernelGGL((sample_iid_kernel<8>), g, b, 0, st, (const half_t*)logits, ld, d_row_ids, vocab, k, temperature,
                           d_u24, d_out, d_branch, d_out_off);
    else if (vocab <= 32 * VER_THREADS)
        hipLaunchKernelGGL((sample_iid_kernel<32>), g, b, 0, st, (const half_t*)logits, ld, d_row_ids, vocab, k, temperature,
                           d_u24, d_out, d_branch, d_out_off);
    else
        return SQ_EUNSUPPORTED;
    return sq_check_launch();
}

// ---- nucleus (top-p) filter on the target logits, in place (utils.get_sampling_logits, utils.py:65-77) ----
// One workgroup per row.  A token is removed iff the probability mass ranked strictly before it
// (descending logit, ties by token id), summed exactly on the 2^-24 grid and rounded to fp16 like
// the reference's fp16 cumsum, exceeds fp16(top_p).  No sort: the boundary key level is found by a
// 16-step bisection over the ordered fp16 key space (each step = one masked exact block sum), ties
// inside the boundary level are resolved with an exact prefix scan in token order.
template <int EPT>
__global__ void __launch_bounds__(VER_THREADS)
top_p_filter_kernel(half_t* __restrict__ logits, int64_t ld, int vocab, float top_p, float temperature) {
    constexpr int CH = EPT / 8;
    __shared__ float s_f[VER_WAVES];
    __shared__ unsigned long long s_u[VER_WAVES];
    __shared__ unsigned long long s_ct[VER_WAVES][EPT / 8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    half_t* x = logits + (size_t)blockIdx.x * ld;
    half2v p[EPT / 2];
    row_softmax_f16<EPT>(x, vocab, temperature, t, p, s_f);
    uint32_t okey[EPT];          // ordered key of the raw logit (sort key of the reference)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e0 = v_elem(c, t, 0);
        half8 v;
        if (e0 < vocab) v = *(const half8*)(x + e0);
#pragma unroll
        for (int j = 0; j < 8; ++j) okey[c * 8 + j] = (e0 < vocab) ? f16_to_ordered(v[j]) : 0u;
    }
    const half_t th16 = (half_t)top_p;
    auto mass_above = [&](uint32_t kappa) {          // exact sum of w_i over okey_i > kappa
        uint32_t part = 0u;
#pragma unroll
        for (int i = 0; i < EPT; ++i)
            if (okey[i] > kappa && v_elem(i >> 3, t, 0) < vocab) part += (uint32_t)((float)H2(p, i) * 16777216.0f);
        unsigned long long ws = wave_sum_u32_wide_dpp(part);
        if (lane == 0) s_u[wave] = ws;
        __syncthreads();
        unsigned long long tot = 0ull;
#pragma unroll
        for (int i = 0; i < VER_WAVES; ++i) tot += s_u[i];
        __syncthreads();
        return tot;
    };
    // smallest kappa whose level is still reachable: h(A(kappa)) <= th16  (monotone in kappa)
    uint32_t lo = 0u, hi = 65535u;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const bool reach = !(grid_sum_to_f16(mass_above(mid)) > th16);
        if (reach) hi = mid; else lo = mid + 1u;
    }
    const uint32_t kmin = lo;
    const unsigned long long a_min = mass_above(kmin);
    // exclusive prefix, in token order (chunk, thread, j), of the tie masses at level kmin
    uint32_t tsum[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        uint32_t sv = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (okey[c * 8 + j] == kmin && v_elem(c, t, 0) < vocab) sv += (uint32_t)((float)H2(p, c * 8 + j) * 16777216.0f);
        tsum[c] = sv;
        const unsigned long long wsum = wave_sum_u32_wide_dpp(sv);
        if (lane == 0) s_ct[wave][c] = wsum;
    }
    __syncthreads();
    unsigned long long run = a_min;                 // mass before the first tie of chunk c, wave 0
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        unsigned long long before = run;
        for (int w2 = 0; w2 < VER_WAVES; ++w2) {
            if (w2 < wave) before += s_ct[w2][c];
            run += s_ct[w2][c];
        }
        // intra-wave exclusive prefix of tsum[c]
        unsigned long long inc = tsum[c];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t l32 = (uint32_t)__shfl_up((int)(uint32_t)inc, o, 64);
            uint32_t h32 = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), o, 64);
            if (lane >= o) inc += ((unsigned long long)h32 << 32) | l32;
        }
        unsigned long long acc = before + inc - tsum[c];
        const int e0 = v_elem(c, t, 0);
        if (e0 < vocab) {
            half8 v = *(const half8*)(x + e0);
            bool dirty = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = okey[c * 8 + j];
                bool remove = k < kmin;
                if (k == kmin) {
                    remove = grid_sum_to_f16(acc) > th16;          // mass ranked strictly before this tie
                    acc += (unsigned long long)(uint32_t)((float)H2(p, c * 8 + j) * 16777216.0f);
                }
                if (remove && (e0 < vocab)) { v[j] = (half_t)(-INFINITY); dirty = true; }
            }
            if (dirty) *(half8*)(x + e0) = v;
        }
    }
}

extern "C" int sq_top_p_filter_f16(void* logits, int64_t ld, int n_rows, int vocab, float top_p, float temperature,
                                   void* stream) {
    if (!logits || n_rows < 0 || vocab <= 0 || ld < vocab || !(temperature > 0.f) || !(top_p > 0.f)) return SQ_EINVAL;
    if ((vocab & 7) || (ld & 7) || ((uintptr_t)logits & 15)) return SQ_EUNSUPPORTED;
    if (n_rows == 0 || top_p >= 1.0f) return SQ_OK;          // identity at top_p = 1 (utils.py:68)
    hipStream_t st = (hipStream_t)stream;
    dim3 g(n_rows), b(VER_THREADS);
    if (vocab <= 8 * VER_THREADS)
        hipLaunchKernelGGL((top_p_filter_kernel<8>), g, b, 0, st, (half_t*)logits, ld, vocab, top_p, temperature);
    else if (vocab <= 32 * VER_THREADS)
        hipLaunchKernelGGL((top_p_filter_kernel<32>), g, b, 0, st, (half_t*)logits, ld, vocab, top_p, temperature);
    else
        return SQ_EUNSUPPORTED;
    return sq_check_launch();
}

// greedy: argmax per node (ties -> lowest id), then the same walker with token equality
template <int EPT>
__global__ void __launch_bounds__(VER_THREADS)
argmax_rows_kernel(const half_t* __restrict__ logits, int vocab, int n_tree, void* ws_raw) {
    __shared__ unsigned long long s_b[VER_WAVES];
    const int t = threadIdx.x;
    const half_t* x = logits + (size_t)blockIdx.x * vocab;
    unsigned long long best = 0ull;
#pragma unroll
    for (int c = 0; c < EPT / 8; ++c) {
        const int e0 = v_elem(c, t, 0);
        if (e0 < vocab) {
            const half8 v = *(const half8*)(x + e0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // NaN must not win an argmax used as a token id: order it lowest
                const half_t val = v[j];
                const uint32_t ord = (val != val) ? 0u : f16_to_ordered(val);
                const unsigned long long comp = ((unsigned long long)(ord + 1u) << 32) | (uint32_t)(0xffffffffu - (uint32_t)(e0 + j));
                best = comp > best ? comp : best;
            }
        }
    }
    const unsigned long long win = block_max_u64<VER_WAVES>(best, s_b);
    if (t == 0) ws_layout(ws_raw, n_tree).tgt[blockIdx.x] = (int64_t)(0xffffffffu - (uint32_t)(win & 0xffffffffu));
}

extern "C" int sq_verify_greedy_f16(const void* target_logits, int64_t* tokens, int token_capacity,
                                    const int32_t* d_child_off, const int32_t* d_child_ids, int n_tree, int vocab, int gt,
                                    void* workspace, int32_t* d_result, int32_t* d_step, int32_t* d_result_ring,
                                    void* stream) {
    if (!target_logits || !tokens || !d_child_off || !workspace || !d_result) return SQ_EINVAL;
    if (n_tree <= 0 || n_tree > SQ_MAX_TREE || vocab <= 0 || (!d_step && gt < 1)) return SQ_EINVAL;
    if (n_tree > 1 && !d_child_ids) return SQ_EINVAL;
    if ((vocab & 7) || ((uintptr_t)target_logits & 15)) return SQ_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(n_tree), b(VER_THREADS);
    if (vocab <= 8 * VER_THREADS)
        hipLaunchKernelGGL((argmax_rows_kernel<8>), g, b, 0, st, (const half_t*)target_logits, vocab, n_tree, workspace);
    else if (vocab <= 32 * VER_THREADS)
        hipLaunchKernelGGL((argmax_rows_kernel<32>), g, b, 0, st, (const half_t*)target_logits, vocab, n_tree, workspace);
    else if (vocab <= 128 * VER_THREADS)
        hipLaunchKernelGGL((argmax_rows_kernel<128>), g, b, 0, st, (const half_t*)target_logits, vocab, n_tree, workspace);
    else
        return SQ_EUNSUPPORTED;
    int rc = sq_check_launch();
    if (rc != SQ_OK) return rc;
    hipLaunchKernelGGL(verify_walk_kernel, dim3(1), dim3(64), 0, st, (half_t*)nullptr, tokens, token_capacity, d_child_off,
                       d_child_ids, n_tree, vocab, gt, workspace, d_result, 1, (const int64_t*)nullptr, d_step, d_result_ring);
    return sq_check_launch();
}

// token-equality walk against caller-supplied target tokens (GreedySTree: one token SAMPLED per node from the
// target distribution instead of the argmax, Tree/GreedySTree.py:188-190,196-214)
extern "C" int sq_verify_tokens_f16(const int64_t* d_target_tokens, int64_t* tokens, int token_capacity,
                                    const int32_t* d_child_off, const int32_t* d_child_ids, int n_tree, int gt,
                                    void* workspace, int32_t* d_result, void* stream) {
    if (!d_target_tokens || !tokens || !d_child_off || !workspace || !d_result) return SQ_EINVAL;
    if (n_tree <= 0 || n_tree > SQ_MAX_TREE || gt < 1) return SQ_EINVAL;
    if (n_tree > 1 && !d_child_ids) return SQ_EINVAL;
    hipLaunchKernelGGL(verify_walk_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (half_t*)nullptr, tokens, token_capacity,
                       d_child_off, d_child_ids, n_tree, 0, gt, workspace, d_result, 1, d_target_tokens, (int32_t*)nullptr,
                       (int32_t*)nullptr);
    return sq_check_launch();
}
