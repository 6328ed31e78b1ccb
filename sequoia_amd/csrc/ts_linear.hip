// Tall-skinny linear layers of the tree forwards: out[M, N] = A[M, K] . W[N, K]^T for M <= 144 rows
// (M = the nodes of one speculation tree / tree level), fp16 in, fp32 accumulate, fp16 out.
//
// The regime: every weight is used by <= 128 rows, so the layer is an HBM stream of W (7B: 13 GB per
// verify) with a little MFMA work on the side.  The kernel is built as a weight stream first:
//   * both operands live in HBM in MFMA-fragment order ("fragment-major"): for v_mfma_f32_16x16x32_f16 a lane
//     (g = lane / 16, r = lane % 16) supplies 8 consecutive k of row r, k = 32 kb + 8 g + (0..7); the image
//         X_f[tile = row / 16][kb = k / 32][lane][8]            (weights:      tile-major, 1 KB per (tile, kb))
//         A_f[kb = k / 32][tile = row / 16][lane][8]            (activations:  k-major)
//     makes every wave-wide operand load one contiguous 1 KB read.  Row-major operands cost this kernel
//     2-2.5x (measured: 16 rows x 64 B per instruction, rows a power-of-two stride apart: half-used lines and
//     L2 channel conflicts).  Weights are repacked once at load (sq_repack_linear_weight_f16); activations are
//     written fragment-major by their producers (RMSNorm, attention, the SwiGLU epilogue below);
//   * a workgroup owns a contiguous range of 16-column units of the output (balanced partition of N / 16
//     units over the launch's `tiles` workgroups, <= NT MFMA column tiles each) and ALL M activation rows;
//     its 4 waves split the workgroup's K range, so every weight byte is loaded by exactly one wave,
//     straight into the MFMA B-operand registers (no LDS), non-temporal, and each wave streams NT contiguous
//     runs of HBM;
//   * HBM latency is covered by depth, not occupancy: each wave keeps a ring of D k-steps of operand
//     loads in flight (D x (NT + MT) 16-byte loads per lane; 4 waves x D x NT KB of weights per CU);
//   * the 4 K-partials of a workgroup meet in LDS once ([row][col] fp32 images, one barrier) and are
//     summed in wave order by the threads that write the output, 16 bytes per lane;
//   * small-N layers (o_proj, down_proj: too few column tiles to fill 256 CUs) also split K over `splits`
//     workgroups; each writes its fp32 partial matrix to slab[split][M][N] and the consumer
//     (sq_add_rmsnorm_slabs_f16: the residual add + RMSNorm that follows both layers) sums them in split
//     order -- no atomics, no fences, results independent of scheduling;
//   * epilogues (splits == 1): plain, + residual (fp16 add after the fp16 rounding of the product), or
//     SwiGLU (W = [gate rows | up rows]; a unit = 16 gate + the matching 16 up rows; writes
//     h(h(silu(h(g))) * h(u)), the rounding points of LlamaMLP_FI, Engine/Llama_modules.py:271); the output
//     is row-major or fragment-major (when it feeds the next tall-skinny layer).
#include "common.h"
#include <stdlib.h>

// Compile-time experiment switches (-DTS_DBG=bits, tools/ts_dbg_build.sh): 1 = activations from one k-step (L1 hits),
// 2 = weights likewise, 4 = no MFMA (INVALID as a timing: with the MFMAs gone the compiler deletes a third of the loop's loads
// and every s_waitcnt -- round 6 found the loop of that build without a single wait), 64 = no MFMA but every operand
// fragment still consumed by an empty asm statement (loads and waits stay: the valid "loads only" build),
// 8 = no loads in the loop, 16 = no prologue loads, 32 = no merge / stores.
// They must not be run-time branches: a conditional around the loads makes the
// compiler drain vmcnt at the join and the kernel loses a third of its speed (measured).
#ifndef TS_DBG
#define TS_DBG 0
#endif
#define TS_WAVES 4
#define TS_THREADS (TS_WAVES * 64)
#define TS_MAXM 144              // 9 row tiles: the 129-node 64x2 tree of the 70B configuration

struct TsParams {
    const half_t* a;      // fragment-major activations [K/32][mtp][64][8]
    const half_t* w;      // fragment-major weights [N/16][K/32][64][8]  (SILU: N = 2 n_out, gate tiles then up tiles)
    const half_t* res;    // [M][ldo] residual or null (row-major output only)
    half_t* out;          // [M][ldo] row-major, or fragment-major [n_out/32][mtp][64][8]   (splits == 1)
    float* slab;          // [splits][M][n_out] fp32                                         (splits > 1)
    int m, mtp, n_out, k, ldo, splits, tiles, units, out_frag;
};


// row tiles per merge pass: 4 wave images of [16 MH][16 NT + 4] fp32 must fit 150 KB of LDS
// (experiment switches, tools/ts_dbg_build.sh: TS_MERGE_KB caps the merge area -- 70 lets TWO workgroups share a compute unit --,
// TS_FORCE_D fixes the ring depth)
#ifndef TS_MERGE_KB
#define TS_MERGE_KB 150
#endif
constexpr int ts_merge_tiles(int mt, int nt) {
    const int per_tile = TS_WAVES * 16 * (nt * 16 + 4) * 4;
    const int fit = (TS_MERGE_KB * 1024) / per_tile;
    return fit >= mt ? mt : fit;
}

// TAIL: the activation block has 16 MT + 1 rows -- a tree of 2^k-ary levels plus its root: the 129-node 64x2 / 16x8 / 128x1
// trees of the reference's growmaps.  Rounding 129 rows up to 9 MFMA row tiles costs an eighth more activation ingest and
// MFMA work and, above all, accumulator registers (9 x 8 tiles do not fit: <= 6 column tiles per workgroup).  Here the 16 MT
// full tiles go through the MFMAs and the ONE extra row rides
// beside them on the vector ALU: its 8 k-values of the lane's k-group (a 16-byte broadcast load from tile MT of the
// fragment-major image, row 0) are multiplied into the weight fragment the lane holds anyway -- 4 v_dot2c_f32_f16 per column
// tile and k-step, on the otherwise idle VALU -- and the 4 k-groups' partial dots meet by two lane shuffles at the end.
template <int MT, int NT, int D, bool SILU, bool TAIL>
__device__ __forceinline__ void ts_linear_body(const TsParams& P) {
    extern __shared__ float ts_lds[];                       // 4 waves x [MH*16][LDW] fp32 (MH = row tiles per merge pass)
    constexpr int LDW = NT * 16 + 4;                         // row stride: 16-byte aligned, 4 rows apart = 16 banks apart
    constexpr int TPU = SILU ? 2 : 1;                        // MFMA column tiles per 16-column output unit
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: K ranges and loop bounds stay scalar
    const int r16 = lane & 15, g = lane >> 4;
    // Block -> (column tile, K split).  Block b runs on XCD b % 8 (observed placement, used for speed only), and each XCD has its
    // own L2: with 2 / 4 / 8 K-splits the split is b % splits, so every XCD works on ONE K range and its L2 holds only that
    // 1 / splits slice of the activation image (7B down_proj: 0.7 MB instead of the whole 2.8 MB image per XCD -- PMC showed the
    // image re-fetched once per XCD: o / down 1.20-1.22x, the 70B down_proj 1.34x of the algorithmic bytes).  Other split counts
    // keep the tile-minor order.  The work per (tile, split) and its result are the same either way.
    const bool split_minor = P.splits == 2 || P.splits == 4 || P.splits == 8;
    const int tile = split_minor ? (int)(blockIdx.x / P.splits) : (int)(blockIdx.x % P.tiles);
    const int split = split_minor ? (int)(blockIdx.x % P.splits) : (int)(blockIdx.x / P.tiles);
    const int u0 = (int)((long)tile * P.units / P.tiles), u1 = (int)((long)(tile + 1) * P.units / P.tiles);
    const int nu = u1 - u0;                                  // 16-column units of this workgroup (<= NT / TPU)

    // ---- this wave's K range -----------------------------------------------------------------------------
    const int ksteps = P.k >> 5;
    const int parts = P.splits * TS_WAVES;
    const int per = (ksteps + parts - 1) / parts;
    const int ks0 = min(ksteps, (split * TS_WAVES + wave) * per), ks1 = min(ksteps, ks0 + per);

    // byte offsets of this lane's operand fragments at k-step 0; column tiles beyond the workgroup's range and
    // row tiles beyond the activation's alias the last valid one (L1 hits; their products are never stored)
    uint32_t woff[NT], aoff[MT];
    const uint32_t w_tile_bytes = (uint32_t)ksteps * 1024u;   // one 16-row weight tile, all of K
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tt = min(t, nu * TPU - 1);
        const int wtile = u0 + tt / TPU + ((SILU && (tt & 1)) ? P.units : 0);
        woff[t] = (uint32_t)wtile * w_tile_bytes + (uint32_t)lane * 16u;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aoff[mt] = (uint32_t)min(mt, P.mtp - 1) * 1024u + (uint32_t)lane * 16u;
    const uint32_t atoff = (uint32_t)MT * 1024u + (uint32_t)g * 256u;      // TAIL: row 16 MT = tile MT, r16 = 0, k-group g
    const uint32_t a_step = (TS_DBG & 1) ? 0u : (uint32_t)P.mtp * 1024u;
    const uint32_t w_shift = (TS_DBG & 2) ? 31u : 10u;
    const char* wbase = (const char*)P.w;
    const char* abase = (const char*)P.a;

    floatx4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = floatx4{0.f, 0.f, 0.f, 0.f};
    float tacc[TAIL ? NT : 1];                               // TAIL: this lane's partial dots of the extra row, per column tile
#pragma unroll
    for (int t = 0; t < (TAIL ? NT : 1); ++t) tacc[t] = 0.f;

    if (ks0 < ks1) {
        half8 wr[D][NT], ar[D][MT], at[TAIL ? D : 1];
        // Step i of this wave's range is k-step ks0 + (i + rot) mod n: workgroups start at different places of K, so
        // at any moment they read different lines of the shared activation image instead of all walking it in
        // lockstep (measured: 1-3 % on the layer projections, 8-12 % on lm_head's 500 workgroups).  The rotation is a
        // function of the tile index only: results stay deterministic.
        const int nst = ks1 - ks0;
        const int rot = (int)(((unsigned)tile * 2654435761u >> 8) % (unsigned)nst);
#define TS_KS(I) ({ int i_ = (I) + rot; i_ = i_ >= nst ? i_ - nst : i_; i_ = i_ >= nst ? i_ - nst : i_; ks0 + i_; })
#define TS_LOAD(d, KS)                                                                                 \
    {                                                                                                  \
        const uint32_t kw_ = ((uint32_t)(KS) << w_shift) & 0x7fffffffu, ka_ = (uint32_t)(KS) * a_step;                      \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                 \
            wr[d][t] = __builtin_nontemporal_load((const half8*)(wbase + (woff[t] + kw_)));            \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) ar[d][mt] = *(const half8*)(abase + (aoff[mt] + ka_)); \
        if (TAIL) at[d] = *(const half8*)(abase + (atoff + ka_));                                      \
    }
#define TS_MMA(d)                                                                                      \
    {                                                                                                  \
        if (TS_DBG & 64) {                                                                             \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) asm volatile("" :: "v"(ar[d][mt]));      \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) asm volatile("" :: "v"(wr[d][t]));          \
        } else                                                                                         \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                              \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                             \
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[d][mt], wr[d][t], acc[mt][t], 0, 0, 0); \
        if (TAIL) {                                                                                    \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                             \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
                    tacc[t] = __builtin_amdgcn_fdot2(half2v{at[d][2 * j], at[d][2 * j + 1]},           \
                                                     half2v{wr[d][t][2 * j], wr[d][t][2 * j + 1]}, tacc[t], false); \
        }                                                                                              \
    }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (!(TS_DBG & 16)) { TS_LOAD(d, TS_KS(min(d, nst - 1))); }
            else {
                _Pragma("unroll") for (int t = 0; t < NT; ++t) wr[d][t] = half8{1, 1, 1, 1, 1, 1, 1, 1};
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) ar[d][mt] = half8{1, 1, 1, 1, 1, 1, 1, 1};
                if (TAIL) at[d] = half8{1, 1, 1, 1, 1, 1, 1, 1};
            }
        }
        const int nfull = nst / D, rem = nst - nfull * D;
        int i = 0;
        for (int it = 0; it < nfull; ++it, i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (!(TS_DBG & 4)) TS_MMA(d);
                if (!(TS_DBG & 8)) TS_LOAD(d, TS_KS(min(i + D + d, nst - 1)));   // refill the stage just consumed (clamped at the end)
                __builtin_amdgcn_sched_barrier(0);           // keep the stages in ring order (no cross-stage MFMA interleave)
            }
        }
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (d < rem) TS_MMA(d);                          // stages 0 .. rem-1 hold the last steps
#undef TS_KS
#undef TS_MMA
#undef TS_LOAD
    }

    // ---- the 4 wave partials meet in LDS: image[wave][row][col], MFMA C layout row = 16 mt + 4 g + i; MH row tiles per
    //      pass so that the four images fit the CU's LDS (one pass except for the widest tiles) --------------------
    constexpr int MH = ts_merge_tiles(MT, NT);
    if (TS_DBG & 32) {                                         // experiment: no merge, no stores (keeps the accumulators live)
        float sacc = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) sacc += acc[mt][t][0] + acc[mt][t][1] + acc[mt][t][2] + acc[mt][t][3];
        if (sacc == 12345.678f) P.out[tid] = (half_t)sacc;
        return;
    }
    const int groups = nu * 2;                                 // output items: (row, 8-column group); a unit holds 2
    // one output item: the 4 wave images [img_rows][LDW] at ts_lds, summed in wave order, through the epilogue
    auto emit = [&](int row, int img_row, int img_rows, int grp) {
        const int unit = grp >> 1, half = grp & 1;
        const int col = (unit * TPU) * 16 + half * 8;          // LDS column of the main (gate) values
        float v[8], u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = 0.f; u[j] = 0.f; }
#pragma unroll
        for (int wv = 0; wv < TS_WAVES; ++wv) {
            const float* src = ts_lds + ((size_t)wv * img_rows + img_row) * LDW + col;
            const floatx4 x = *(const floatx4*)src, y = *(const floatx4*)(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += x[j]; v[4 + j] += y[j]; }
            if (SILU) {
                const floatx4 p = *(const floatx4*)(src + 16), q = *(const floatx4*)(src + 20);
#pragma unroll
                for (int j = 0; j < 4; ++j) { u[j] += p[j]; u[4 + j] += q[j]; }
            }
        }
        const int ocol = (u0 + unit) * 16 + half * 8;
        if (P.splits > 1) {
            float* dst = P.slab + ((size_t)split * P.m + row) * P.n_out + ocol;
            *(floatx4*)dst = floatx4{v[0], v[1], v[2], v[3]};
            *(floatx4*)(dst + 4) = floatx4{v[4], v[5], v[6], v[7]};
            return;
        }
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t h = (half_t)v[j];
            if (SILU) {
                const float gf = (float)h;
                const half_t sg = (half_t)(gf / (1.0f + expf(-gf)));
                h = (half_t)((float)sg * (float)(half_t)u[j]);
            }
            o[j] = h;
        }
        if (P.out_frag) {      // element (row, ocol + j) -> [ocol / 32][row / 16][(ocol / 8 % 4) * 16 + row % 16][j]
            const size_t foff = (((size_t)(ocol >> 5) * P.mtp + (row >> 4)) * 64 + ((ocol >> 3) & 3) * 16 + (row & 15)) * 8;
            *(half8*)(P.out + foff) = o;
            return;
        }
        const size_t off = (size_t)row * P.ldo + ocol;
        if (!SILU && P.res) {
            const half8 r = *(const half8*)(P.res + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)o[j] + (float)r[j]);
        }
        *(half8*)(P.out + off) = o;
    };
    for (int m0 = 0; m0 < MT; m0 += MH) {
        if (m0 > 0) __syncthreads();                           // the previous pass has been read out
        float* mine = ts_lds + (size_t)wave * (MH * 16) * LDW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (mt >= m0 && mt < m0 + MH)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mine[((mt - m0) * 16 + g * 4 + i) * LDW + t * 16 + r16] = acc[mt][t][i];
        __syncthreads();
        const int row_lo = m0 * 16, row_hi = min(min(P.m, MT * 16), (m0 + MH) * 16);
        const int items = max(0, row_hi - row_lo) * groups;
        for (int it = tid; it < items; it += TS_THREADS) emit(row_lo + it / groups, it / groups, MH * 16, it % groups);
    }
    if (TAIL) {
        // the extra row: the 4 k-groups of a wave meet by lane shuffles, the 4 waves through one-row LDS images
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float x = tacc[t];
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            if (g == 0) ts_lds[(size_t)wave * LDW + t * 16 + r16] = x;
        }
        __syncthreads();
        if (MT * 16 < P.m)
            for (int it = tid; it < groups; it += TS_THREADS) emit(MT * 16, 0, 1, it);
    }
}

template <int MT, int NT, int D, bool SILU>
__global__ void __launch_bounds__(TS_THREADS) ts_linear_kernel(const TsParams P) { ts_linear_body<MT, NT, D, SILU, false>(P); }

// 16 MT + 1 activation rows (the kernel name keeps the MFMA row-tile count: <8, ...> runs 129 rows)
template <int MT, int NT, int D, bool SILU>
__global__ void __launch_bounds__(TS_THREADS) ts_linear_tail_kernel(const TsParams P) { ts_linear_body<MT, NT, D, SILU, true>(P); }


extern "C" size_t sq_linear_ts_workspace_bytes(int m, int n_out, int splits) {
    return splits > 1 ? (size_t)splits * m * n_out * sizeof(float) : 0;
}

template <int MT, int NT, bool SILU, bool TAIL>
static void ts_go(const TsParams& P, hipStream_t st) {
    // ring depth: the vmcnt counter tracks 63 loads, so D (NT + MT) must stay below.  Measured on MI355X: deeper
    // rings (up to 8), 8-wave workgroups, and both operands through LDS-DMA (global_load_lds_dwordx4 into a
    // wave-private LDS ring, hand-counted waits: 27.6 vs 29.6 us on the 128-row qkv, 64 vs 62 us on gate_up) are no
    // faster -- the stream is bound by the CU's memory ingest (~14 B/clk/CU for weights + activations together),
    // not by bytes in flight or by the VGPR return path.
#ifdef TS_FORCE_D
    constexpr int D = TS_FORCE_D;
#else
    constexpr int D = (MT * NT > 48) ? 2 : ((MT * NT > 24) ? 3 : 4);     // 8 x 8 accumulator tiles: 256 registers, ring of 2
#endif
    const size_t lds = (size_t)TS_WAVES * ts_merge_tiles(MT, NT) * 16 * (NT * 16 + 4) * sizeof(float);
    void (*kern)(const TsParams);
    if constexpr (TAIL) kern = ts_linear_tail_kernel<MT, NT, D, SILU>;
    else kern = ts_linear_kernel<MT, NT, D, SILU>;
    static bool attr_done[16] = {};          // the attribute is per (function, device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(P.tiles * P.splits), dim3(TS_THREADS), lds, st, P);
}

template <int MT, bool TAIL = false>
static int ts_dispatch(const TsParams& P, int silu, int nt, hipStream_t st) {
    if (silu) {
        if (nt <= 2) ts_go<MT, 2, true, TAIL>(P, st);
        else if (nt <= 4) ts_go<MT, 4, true, TAIL>(P, st);
        else if (nt <= 6) ts_go<MT, 6, true, TAIL>(P, st);
        else if constexpr (MT <= 8) {                   // 4 gate+up units per workgroup (13B gate_up: 864 units -> 216 workgroups,
            if (nt <= 8) ts_go<MT, 8, true, TAIL>(P, st);     // one resident wave of them instead of 288 = 256 + 32)
            else return SQ_EUNSUPPORTED;
        } else return SQ_EUNSUPPORTED;
        return SQ_OK;
    }
    if (nt <= 2) ts_go<MT, 2, false, TAIL>(P, st);
    else if (nt <= 3) ts_go<MT, 3, false, TAIL>(P, st);
    else if (nt <= 4) ts_go<MT, 4, false, TAIL>(P, st);
    else if (nt <= 6) ts_go<MT, 6, false, TAIL>(P, st);
    else if constexpr (MT <= 8 && !(TAIL && MT == 8)) { // 9 x 8 accumulator tiles + a ring of 2 do not fit 512 registers
        if (nt <= 8) ts_go<MT, 8, false, TAIL>(P, st);        // (52 spilled VGPRs): 129-144 rows take <= 6 column tiles; the plain
        //                                                       8 x 8 + extra-row build spills 5: 129 rows take <= 6 there too
        else return SQ_EUNSUPPORTED;
    } else return SQ_EUNSUPPORTED;
    return SQ_OK;
}

// 129 rows (the 64x2 / 16x8 / 128x1 ... trees: 128 nodes + the root) run the 8-tile kernel with the extra row on the vector ALU
// (ts_linear_body<.., TAIL>): what it buys is the 8-column-tile SwiGLU workgroup the 9-tile build has no registers for
// (70B gate_up at 129 rows: 249 -> 224 us); SEQUOIA_TS_TAIL=0 rounds 129 up to 9 row tiles as before
static bool ts_tail_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SEQUOIA_TS_TAIL"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1;
}

extern "C" int sq_linear_ts_f16(const void* a_frag, const void* w_frag, const void* res, void* out, int ldo, int out_frag,
                                int m, int n_out, int k, int silu, int tiles, int splits, void* slab, size_t slab_bytes,
                                void* stream) {
    if (!a_frag || !w_frag || m <= 0 || n_out <= 0 || k <= 0 || splits < 1 || tiles < 1) return SQ_EINVAL;
    if (splits == 1 && (!out || (!out_frag && ldo < n_out))) return SQ_EINVAL;
    if (m > TS_MAXM || (k & 31) || (n_out & 15) || (ldo & 7) || ((uintptr_t)a_frag & 15) || ((uintptr_t)w_frag & 15) ||
        ((uintptr_t)out & 15) || (res && ((uintptr_t)res & 15)) || (size_t)(silu ? 2 : 1) * n_out * k * 2 >= (1ull << 32))
        return SQ_EUNSUPPORTED;
    if ((silu && res) || (out_frag && (res || (n_out & 31)))) return SQ_EUNSUPPORTED;
    TsParams P;
    P.a = (const half_t*)a_frag; P.w = (const half_t*)w_frag; P.res = (const half_t*)res; P.out = (half_t*)out;
    P.slab = (float*)slab;
    P.m = m; P.mtp = (m + 15) / 16; P.n_out = n_out; P.k = k; P.ldo = ldo; P.splits = splits; P.out_frag = out_frag;
    P.units = n_out / 16;
    P.tiles = tiles > P.units ? P.units : tiles;
    if (splits > 1) {
        if (silu || res || out_frag) return SQ_EUNSUPPORTED;   // the slab consumer applies the epilogue
        if (!slab || slab_bytes < (size_t)splits * m * n_out * sizeof(float) || ((uintptr_t)slab & 15)) return SQ_EINVAL;
        if ((k >> 5) < splits * TS_WAVES) return SQ_EUNSUPPORTED;
    }
    const int max_units = (P.units + P.tiles - 1) / P.tiles;
    const int nt = max_units * (silu ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    int rc = SQ_OK;
    const int mt = P.mtp;                                    // row tiles; 5 and 7 run on the 6 / 8 builds (tiles alias)
    // (65 rows -- 4 tiles + 1 -- measured equal on the 6-tile build, whose aliased tiles hit L1: not routed here)
    if (m == 129 && ts_tail_enabled()) rc = ts_dispatch<8, true>(P, silu, nt, st);
    else if (mt <= 1) rc = ts_dispatch<1>(P, silu, nt, st);
    else if (mt == 2) rc = ts_dispatch<2>(P, silu, nt, st);
    else if (mt == 3) rc = ts_dispatch<3>(P, silu, nt, st);
    else if (mt == 4) rc = ts_dispatch<4>(P, silu, nt, st);
    else if (mt <= 6) rc = ts_dispatch<6>(P, silu, nt, st);
    else if (mt <= 8) rc = ts_dispatch<8>(P, silu, nt, st);
    else rc = ts_dispatch<9>(P, silu, nt, st);
    if (rc != SQ_OK) return rc;
    return sq_check_launch();
}

// ---- fragment-major repacks ----------------------------------------------------------------------------------
// weights: w [n][k] row-major -> w_f[n / 16][k / 32][lane = (k / 8 % 4) * 16 + n % 16][8]   (once, at load)
__global__ void __launch_bounds__(256) repack_weight_kernel(const half_t* __restrict__ w, half_t* __restrict__ wf, int n, int k) {
    const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x;        // one 16-byte chunk of the OUTPUT image
    const size_t total = (size_t)n * k / 8;
    if (chunk >= total) return;
    const int ksteps = k >> 5;
    const int lane = (int)(chunk & 63);
    const size_t blk = chunk >> 6;
    const int kb = (int)(blk % ksteps), tile = (int)(blk / ksteps);
    const int row = tile * 16 + (lane & 15), col = kb * 32 + (lane >> 4) * 8;
    *(half8*)(wf + chunk * 8) = *(const half8*)(w + (size_t)row * k + col);
}

extern "C" int sq_repack_linear_weight_f16(const void* w, void* w_frag, int n, int k, void* stream) {
    if (!w || !w_frag || n <= 0 || k <= 0) return SQ_EINVAL;
    if ((n & 15) || (k & 31) || w == w_frag) return SQ_EUNSUPPORTED;
    const size_t chunks = (size_t)n * k / 8;
    hipLaunchKernelGGL(repack_weight_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)w, (half_t*)w_frag, n, k);
    return sq_check_launch();
}

// activations: x [m][ldx] row-major -> x_f[k / 32][ceil(m / 16)][lane = (k / 8 % 4) * 16 + m % 16][8]; rows beyond m
// are zero.  (The production producers write this image directly; this entry serves callers that hold row-major data.)
__global__ void __launch_bounds__(256) repack_rows_kernel(const half_t* __restrict__ x, half_t* __restrict__ xf, int m, int k, int ldx) {
    const int mtp = (m + 15) >> 4;
    const size_t chunk = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)mtp * 16 * k / 8;
    if (chunk >= total) return;
    const int lane = (int)(chunk & 63);
    const size_t blk = chunk >> 6;
    const int mt = (int)(blk % mtp), kb = (int)(blk / mtp);
    const int row = mt * 16 + (lane & 15), col = kb * 32 + (lane >> 4) * 8;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < m) v = *(const half8*)(x + (size_t)row * ldx + col);
    *(half8*)(xf + chunk * 8) = v;
}

extern "C" int sq_repack_rows_frag_f16(const void* x, int ldx, void* x_frag, int m, int k, void* stream) {
    if (!x || !x_frag || m <= 0 || k <= 0 || ldx < k) return SQ_EINVAL;
    if ((k & 31) || (ldx & 7) || x == x_frag) return SQ_EUNSUPPORTED;
    const size_t chunks = (size_t)((m + 15) / 16) * 16 * k / 8;
    hipLaunchKernelGGL(repack_rows_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)x_frag, m, k, ldx);
    return sq_check_launch();
}
