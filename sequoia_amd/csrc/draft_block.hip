// f1: the attention half of a decoder layer of a SMALL draft model as ONE launch (hidden 512 / 768 / 1024, heads of 64:
// the 68m / 160m drafts), for forwards whose rows never attend to each other -- the draft forward over one tree level
// (siblings and cousins: no node of a level is another's ancestor) and every one-row forward.
//
// What it replaces, per layer:  sq_linear_ts_f16 (qkv) -> sq_rope_kv_write_slabs_f16 -> sq_tree_attention_f16 ->
// sq_linear_ts_f16 (o_proj), four dependent launches that each sit on the ~4.6-5.6 us floor of a dependent graph node
// (profiles/r05_bench_kernel_stats_loop_only.md: 4.7 + 5.0 + 5.6 + 4.8 us for < 5 MB of weights).  Here one workgroup owns
// one (head, 16-row tile):
//   phase 0  every load the block needs that does not depend on this forward's arithmetic is issued up front: the head's
//            q | k | v weight slices (12 column tiles x K, K split over the 8 waves: 36 x 16 B per lane at K = 768), the
//            tile's rows of the normalised activation image, the rows' position ids -> RoPE table rows;
//   phase 1  q | k | v = A . W^T for 16 rows x 192 columns on the MFMAs (v_mfma_f32_16x16x32_f16), the 8 K-partials meet
//            in LDS and are summed in wave order; the loads of the NEXT phases are issued before that meeting: this wave's
//            first chunk of cached K / V rows and its o_proj weight fragments (none depends on q | k | v);
//   phase 2  fp16 rounding of the projection, RoPE with the reference's fp16 rounding after every op
//            (Engine/offload_engine.py:63-66), new K / V rows into their cache slots (Engine/Llama_KV.py:72-89) and into LDS;
//   phase 3  tree attention of the 16 queries over the cached keys [0, q_slot0) -- 32-key chunks dealt over the 8 waves,
//            online softmax, both contractions on the MFMAs, the structure of csrc/tree_attention.hip -- plus each query's
//            OWN key (the only key of this forward a row may see) folded in on the vector ALU;
//   phase 4  the head's slice of o_proj: out_h[16][hidden] = O_h[16][64] . Wo[:, 64 h .. 64 h + 64]^T, written as the fp32
//            partial slab[h][row][:].  The head partials are summed in head order by the residual add + RMSNorm that follows
//            (sq_add_rmsnorm_slabs_f16 with splits = n_heads): o_proj = h(sum_h partial_h), the same rounding point as the
//            split-K projection it replaces; only the fp32 summation order differs.
// KV_ONLY: the last layer of a forward whose logits nobody reads (draft forward over the last tree level): phases 0-2 for
// the k | v slices only.
// Reference lines replaced: LlamaAttention_FI.forward (Engine/Llama_modules.py:87-140): q/k/v_proj, rotary embedding,
// kv_cache.update_kv_cache, the masked softmax attention, o_proj.
#include "common.h"
#include "attn_map.h"

#define DB_WAVES 8
#define DB_THREADS (DB_WAVES * 64)
#define DB_D 64                      // head dimension
#define DB_BM 16                     // rows per workgroup
#define DB_BK 32                     // keys per attention chunk
#define DB_QS (DB_D + 8)             // halves per row of the q / k / v / o LDS tiles (16-byte aligned rows, de-phased banks)
#define DB_MAX_WORDS (SQ_MAX_TREE / 64)
// Compile-time experiment switch (tools/block_dbg_build.sh): the full kernel returns after 1 = phase 2, 2 = the attention
// prologue [removed], 3 = the key loop [removed], 4 = the attention.  Not a run-time branch: the timeline of the phases is read from the differences.
#ifndef DB_STOP
#define DB_STOP 0
#endif
#define DB_STOP_AT(n, live) if (DB_STOP == (n)) { if ((live) == 12345.678f) P.slab[tid] = (live); return; }

typedef __fp16 db_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct DbParams {
    const half_t* a;          // normalised activations, fragment-major [hidden / 32][mtp][64][8]
    const half_t* wqkv;       // fragment-major [(3 H D) / 16][hidden / 32][64][8]   (q rows | k rows | v rows)
    const half_t* wo;         // fragment-major [hidden / 16][(H D) / 32][64][8]
    float* slab;              // [H][q_len][hidden] fp32: per-head partial products of o_proj
    half_t* k_layer;          // [H][M][D]
    half_t* v_layer;
    const half_t* cos_tab;    // [max_pos][D]
    const half_t* sin_tab;
    const int64_t* position_ids;
    const int64_t* storage_ids;
    const uint64_t* bitmask;  // [n_tree][words]
    const int32_t* ctx;       // optional device override of {q_slot0, gt, kv_len}
    int words, n_tree, q_slot0, gt;
    int q_len, mtp, n_heads, hidden, m;
    float scale_log2e;
};

__device__ __forceinline__ void db_lane_swap16(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void db_lane_swap32(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// max / sum over the four 16-lane groups holding the same query (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float db_group4_max(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    db_lane_swap16(a, b);
    v = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
    a = __builtin_bit_cast(unsigned, v); b = a;
    db_lane_swap32(a, b);
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float db_group4_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    db_lane_swap16(a, b);
    v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    a = __builtin_bit_cast(unsigned, v); b = a;
    db_lane_swap32(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// LDS map (bytes).  The K-partials of phase 1 ([wave][16][LDW] fp32) are dead after phase 2; the attention's wave-private
// K / V tiles and, after them, its merge tiles reuse that area.
template <int NT>
struct DbLds {
    static constexpr int LDW = NT * 16 + 4;                                   // floats per partial row
    static constexpr int PART = DB_WAVES * DB_BM * LDW * 4;
    static constexpr int ATT = 2 * DB_BK * DB_QS * 2 * DB_WAVES > DB_WAVES * DB_BM * (DB_D + 4) * 4
                                   ? 2 * DB_BK * DB_QS * 2 * DB_WAVES : DB_WAVES * DB_BM * (DB_D + 4) * 4;     // == DbAtt<DB_D>::MAIN
    static constexpr int MAIN = PART > ATT ? PART : ATT;
    static constexpr int QKVO = MAIN;                                          // q_s, k_s, v_s, o_s: [16][DB_QS] halves each
    static constexpr int ML = QKVO + 4 * DB_BM * DB_QS * 2;                    // [wave][16][2] floats
    static constexpr int BM = ML + DB_WAVES * DB_BM * 8;                       // [16][DB_MAX_WORDS] u64
    static constexpr int TOTAL = BM + DB_BM * DB_MAX_WORDS * 8;
};


// ---- the attention of one (head, 16-query tile) over the CACHED keys [0, n_keys) plus each query's own key -------------------
// Shared by the fused block below (D = 64) and by level_attention_kernel (D = 64 / 128).  Inputs in LDS: q_s / k_s / v_s =
// the tile's rotated queries, rotated keys and values ([16][D + 8] fp16); lds_bm = the queries' ancestor-bitmask rows
// ([16][DB_MAX_WORDS]); kr_cur / vr_cur = this wave's first chunk of cached K / V rows, already requested by the caller.
// Output: o_s ([16][D + 8] fp16, may alias q_s) = softmax(QK^T / sqrt(D)) V rounded to fp16, visible to every wave on return.
// 32-key chunks are dealt over the 8 waves (wave-private LDS tiles, next chunk in flight in registers), both contractions run
// on v_mfma_f32_16x16x32_f16 (S^T = K Q^T, O^T += V^T P^T with V^T through ds_read_b64_tr_b16), the online-softmax state is
// per wave and merged through LDS -- the structure of csrc/tree_attention.hip; the query's OWN key (the only key of this forward
// it may see) is one more online-softmax step on the vector ALU of the wave with the fewest chunks.
template <int D>
struct DbAtt {
    static constexpr int QS = D + 8;                                           // halves per LDS row
    static constexpr int CPR = D / 8;                                          // 16-byte chunks per K / V row
    static constexpr int VITER = DB_BK * CPR / 64;                             // 16-byte K (and V) loads per lane and chunk
    static constexpr int NTO = D / 16;                                         // output column tiles
    static constexpr int DSTEPS = D / 32;                                      // MFMA k-steps of S^T
    static constexpr int KV_TILE = 2 * DB_BK * QS;                             // halves per wave: V tile then K tile
    static constexpr int ATT_V = DB_WAVES * KV_TILE * 2;
    static constexpr int OSTR = D + 4;
    static constexpr int ATT_O = DB_WAVES * DB_BM * OSTR * 4;
    static constexpr int MAIN = ATT_V > ATT_O ? ATT_V : ATT_O;                 // bytes of the K / V tiles, reused by the merge
};

template <int D>
__device__ __forceinline__ void db_issue_kv(int chunk, int n_keys, int lane, const half_t* kbase, const half_t* vbase,
                                            u32x4 (&kr)[DbAtt<D>::VITER], u32x4 (&vr)[DbAtt<D>::VITER]) {
    constexpr int CPR = DbAtt<D>::CPR;
    const int key0 = chunk * DB_BK;
#pragma unroll
    for (int it = 0; it < DbAtt<D>::VITER; ++it) {
        const int idx = it * 64 + lane;
        const int r = idx / CPR, c = idx % CPR;
        int row = key0 + r; if (row >= n_keys) row = n_keys - 1;
        kr[it] = *(const u32x4*)(kbase + (size_t)row * D + c * 8);
        vr[it] = *(const u32x4*)(vbase + (size_t)row * D + c * 8);
    }
}

template <int D>
__device__ __forceinline__ void db_attend(const half_t* q_s, const half_t* k_s, const half_t* v_s, half_t* o_s,
                                          unsigned char* lds_main, float* lds_ml, const uint64_t* lds_bm,
                                          u32x4 (&kr_cur)[DbAtt<D>::VITER], u32x4 (&vr_cur)[DbAtt<D>::VITER],
                                          const half_t* kbase, const half_t* vbase, int n_keys, int q0, int q_len, int q_slot0,
                                          int gt, int n_tree, int words, float scale_log2e, int tid, int wave) {
    using A = DbAtt<D>;
    constexpr int QS = A::QS, CPR = A::CPR, VITER = A::VITER, NTO = A::NTO, DSTEPS = A::DSTEPS;
    const int lane = tid & 63, qc = lane & 15, g = lane >> 4;
    const int n_chunks = (n_keys + DB_BK - 1) / DB_BK;
    half_t* lds_v = (half_t*)lds_main;
    float* lds_o = (float*)lds_main;
    const int qi_c = min(q0 + qc, q_len - 1);
    const int slot = q_slot0 + qi_c;
    const int tnode = slot - (gt - 1);
    const bool causal_row = slot < gt;
    const bool tree_ok = tnode < n_tree;

    // Q fragments (B operand): lane (n = qc, g) holds Q[qc][32 s + 8 g .. +8]; this lane's share of q . k of its own row
    half8 qf[DSTEPS];
#pragma unroll
    for (int s = 0; s < DSTEPS; ++s) qf[s] = *(const half8*)(q_s + qc * QS + s * 32 + g * 8);
    float s_diag = 0.f;
#pragma unroll
    for (int c = 0; c < D / 32; ++c) {                            // the lane's D / 4 elements of the row, 8 at a time
        const half8 kk = *(const half8*)(k_s + qc * QS + g * (D / 4) + c * 8), qq = *(const half8*)(q_s + qc * QS + g * (D / 4) + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s_diag += (float)qq[e] * (float)kk[e];
    }
    s_diag = db_group4_sum(s_diag);
    __syncthreads();                                              // every wave holds its fragments: the K / V tile area is free

    float m_run = -INFINITY, l_run = 0.f;
    floatx4 o_acc[NTO];
#pragma unroll
    for (int i = 0; i < NTO; ++i) o_acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    half_t* my_v = lds_v + wave * A::KV_TILE;
    half_t* my_k = my_v + DB_BK * QS;
    u32x4 kr_nxt[VITER], vr_nxt[VITER];
    for (int ch = wave; ch < n_chunks; ch += DB_WAVES) {
        const int key0 = ch * DB_BK;
        const bool has_next = ch + DB_WAVES < n_chunks;
        if (has_next) db_issue_kv<D>(ch + DB_WAVES, n_keys, lane, kbase, vbase, kr_nxt, vr_nxt);
#pragma unroll
        for (int it = 0; it < VITER; ++it) {
            const int idx = it * 64 + lane;
            *(u32x4*)(my_k + (idx / CPR) * QS + (idx % CPR) * 8) = kr_cur[it];
            *(u32x4*)(my_v + (idx / CPR) * QS + (idx % CPR) * 8) = vr_cur[it];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // S^T = K Q^T: lane (q = qc, g) receives keys key0 + 16 t + 4 g + r
        floatx4 s_acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx4 a = {0.f, 0.f, 0.f, 0.f};
            half8 kf[DSTEPS];
#pragma unroll
            for (int s = 0; s < DSTEPS; ++s) kf[s] = *(const half8*)(my_k + (t * 16 + qc) * QS + s * 32 + g * 8);
#pragma unroll
            for (int s = 0; s < DSTEPS; ++s) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[s], qf[s], a, 0, 0, 0);
            s_acc[t] = a;
        }
        float sv[8];
        float cmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + t * 16 + g * 4 + r;
                float x = s_acc[t][r] * scale_log2e;
                const int j = key - (gt - 1);
                const uint64_t wv = lds_bm[qc * DB_MAX_WORDS + (((unsigned)j >> 6) < (unsigned)words ? (j >> 6) : 0)];
                const bool bit = (wv >> (j & 63)) & 1ull;
                const bool vis_tree = (key < gt) | (tree_ok & ((unsigned)j < (unsigned)n_tree) & bit);
                const bool vis = (key < n_keys) & (causal_row | vis_tree);     // (a committed-text row sees every key in front of it)
                x = vis ? x : -INFINITY;
                sv[t * 4 + r] = x;
                cmax = fmaxf(cmax, x);
            }
        cmax = db_group4_max(cmax);
        const float m_new = fmaxf(m_run, cmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        half8 pf;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = __builtin_amdgcn_exp2f(sv[j] - m_use);
            psum += p;
            pf[j] = (half_t)p;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            const half_t* a0 = my_v + (0 * 16 + g * 4 + (qc >> 2)) * QS + nt * 16 + (qc & 3) * 4;
            const half_t* a1 = my_v + (1 * 16 + g * 4 + (qc >> 2)) * QS + nt * 16 + (qc & 3) * 4;
            half8 vf;
            const db_fp16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) db_fp16x4*)a0);
            const db_fp16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) db_fp16x4*)a1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { vf[j] = (half_t)b0[j]; vf[4 + j] = (half_t)b1[j]; }
            floatx4 o = o_acc[nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= alpha;
            o_acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        if (has_next) {
#pragma unroll
            for (int it = 0; it < VITER; ++it) { kr_cur[it] = kr_nxt[it]; vr_cur[it] = vr_nxt[it]; }
        }
    }
    // each query's own key (slot q_slot0 + row: written by the caller, not read back): one more online-softmax step on the wave
    // that would take the next chunk.  o_acc[nt][r] = O[q = qc][d = 16 nt + 4 g + r]; P is rounded to fp16 like the MFMA operand.
    if (wave == n_chunks % DB_WAVES) {
        const bool vis = causal_row | tree_ok;
        const float x = vis ? s_diag * scale_log2e : -INFINITY;
        const float m_new = fmaxf(m_run, x);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        const float p = __builtin_amdgcn_exp2f(x - m_use);
        const float ph = (float)(half_t)p;
        l_run = l_run * alpha + (g == 0 ? p : 0.f);          // (l_run is summed over the 4 lane groups below)
        m_run = m_new;
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            const half4 vv = *(const half4*)(v_s + qc * QS + nt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) o_acc[nt][r] = o_acc[nt][r] * alpha + ph * (float)vv[r];
        }
    }

    // merge the waves
    l_run = db_group4_sum(l_run);
    __syncthreads();                                      // everyone is done with the K / V tiles
    if (g == 0) { lds_ml[(wave * DB_BM + qc) * 2] = m_run; lds_ml[(wave * DB_BM + qc) * 2 + 1] = l_run; }
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt)
        *(floatx4*)(lds_o + (wave * DB_BM + qc) * A::OSTR + nt * 16 + g * 4) = o_acc[nt];
    __syncthreads();
    {
        constexpr int EPT = DB_BM * D / DB_THREADS;           // 2 (D = 64) or 4 (D = 128) output elements per thread
        const int row = tid / (D / EPT), col = (tid % (D / EPT)) * EPT;
        float mw[DB_WAVES], lw[DB_WAVES], mmax = -INFINITY;
#pragma unroll
        for (int w = 0; w < DB_WAVES; ++w) {
            mw[w] = lds_ml[(w * DB_BM + row) * 2];
            lw[w] = lds_ml[(w * DB_BM + row) * 2 + 1];
            mmax = fmaxf(mmax, mw[w]);
        }
        float denom = 0.f, wgt[DB_WAVES];
#pragma unroll
        for (int w = 0; w < DB_WAVES; ++w) {
            wgt[w] = (mw[w] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw[w] - mmax);
            denom += lw[w] * wgt[w];
        }
        const float inv = denom > 0.f ? 1.0f / denom : 0.f;
        half_t ov[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < DB_WAVES; ++w) a += lds_o[(w * DB_BM + row) * A::OSTR + col + e] * wgt[w];
            ov[e] = (half_t)(a * inv);
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) o_s[row * QS + col + e] = ov[e];
    }
    __syncthreads();
}

// KSW: k-steps (of 32) per wave: hidden = 256 KSW.
template <int KSW, bool KV_ONLY>
__global__ void __launch_bounds__(DB_THREADS) draft_block_kernel(const DbParams P) {
    constexpr int NPART = KV_ONLY ? 2 : 3;
    constexpr int NT = NPART * 4;                        // MFMA column tiles of this head's q | k | v slice
    using L = DbLds<NT>;
    constexpr int LDW = L::LDW;
    extern __shared__ __attribute__((aligned(16))) unsigned char db_lds[];
    float* part = (float*)db_lds;
    half_t* q_s = (half_t*)(db_lds + L::QKVO);
    half_t* k_s = q_s + DB_BM * DB_QS;
    half_t* v_s = k_s + DB_BM * DB_QS;
    half_t* o_s = v_s + DB_BM * DB_QS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    // block -> (head, row tile): 16 ids per tile, head = id % 16, so every tile of a head runs on XCD head % 8 (block b is
    // placed on XCD b % 8 -- a speed hint: the head's weight slices and K / V rows are fetched into one L2)
    const int head = blockIdx.x & 15, q_tile = blockIdx.x >> 4;
    if (head >= P.n_heads) return;
    const int q0 = q_tile * DB_BM;
    const int ksteps = P.hidden >> 5;                    // == 8 KSW
    const int H4 = P.n_heads * 4;                        // column tiles per q / k / v part

    // ---- phase 0: position / slot of this thread's row first (the RoPE table rows hang on it), then the weight slices and
    //      the activation rows, then the table rows -- issued while the weights are in flight, no branch around any load
    //      (a conditional load makes the compiler drain vmcnt at the join) ---------------------------------------------------
    // phase-2 roles: wave 0 rotates q (KV_ONLY: k), wave 1 rotates k, the two waves behind them move v.  A rotating thread
    // owns chunk rc (8 halves) of the first half of row rrow and the matching chunk of the second half (rotate_half pairs
    // e with e + 32); every thread loads the table rows of ITS (rrow, rc) whether it rotates or not.
    constexpr int ROPE_WAVES = KV_ONLY ? 1 : 2;
    const int rrow = tid & 15, rc = (tid >> 4) & 3;
    const int ri = min(q0 + rrow, P.q_len - 1);
    const int64_t pos = P.position_ids[ri];
    const int64_t rslot = P.storage_ids[ri];
    int q_slot0 = P.q_slot0, gt = P.gt;
    if (P.ctx) { q_slot0 = P.ctx[0]; gt = P.ctx[1]; }
    if (gt < 1) gt = 1;
    if (q_slot0 < 0) q_slot0 = 0;
    if (q_slot0 > P.m) q_slot0 = P.m;
    // ancestor-bitmask word (tid % 8) of query row (tid / 8) -- thread tid < 128 stages it into LDS in phase 3; loaded here so
    // that no load issued later has to land before it (loads return in order)
    uint64_t bm_word = 0ull;
    if constexpr (!KV_ONLY) {
        const int br = (tid >> 3) & 15, bw = tid & 7;
        const int tn = q_slot0 + min(q0 + br, P.q_len - 1) - (gt - 1);
        const bool ok = tid < DB_BM * DB_MAX_WORDS && bw < P.words && tn >= 1 && tn < P.n_tree && P.bitmask;
        if (ok) bm_word = P.bitmask[(size_t)tn * P.words + bw];
    }
    half8 wr[NT][KSW], ar[KSW];
    {
        const char* wbase = (const char*)P.wqkv;
        const char* abase = (const char*)P.a;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int partn = (KV_ONLY ? 1 : 0) + (t >> 2);
            const uint32_t wtile = (uint32_t)(partn * H4 + head * 4 + (t & 3));
#pragma unroll
            for (int s = 0; s < KSW; ++s) {
                const uint32_t ks = (uint32_t)(wave * KSW + s);
                wr[t][s] = *(const half8*)(wbase + (((size_t)wtile * ksteps + ks) * 64 + lane) * 16);
            }
        }
#pragma unroll
        for (int s = 0; s < KSW; ++s) {
            const uint32_t ks = (uint32_t)(wave * KSW + s);
            ar[s] = *(const half8*)(abase + (((size_t)ks * P.mtp + q_tile) * 64 + lane) * 16);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const half8 c1 = *(const half8*)(P.cos_tab + (size_t)pos * DB_D + rc * 8);
    const half8 c2 = *(const half8*)(P.cos_tab + (size_t)pos * DB_D + 32 + rc * 8);
    const half8 s1 = *(const half8*)(P.sin_tab + (size_t)pos * DB_D + rc * 8);
    const half8 s2 = *(const half8*)(P.sin_tab + (size_t)pos * DB_D + 32 + rc * 8);
    // the threads that move V (phase 2): row vrow (== rrow), chunk vc of 8
    const int vt = tid - ROPE_WAVES * 64;
    const int vrow = vt & 15, vc = (vt >> 4) & 7;
    const bool v_thread = vt >= 0 && vt < 128;
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 1: q | k | v partial products of this wave's K range ---------------------------------------------------
    // Loads of the later phases that do not depend on q | k | v -- this wave's first chunk of cached K / V rows, its o_proj
    // weight fragments -- are issued as early as the register file allows: the K / V rows in front of the MFMAs (K = 512 /
    // 768; behind them at K = 1024, where the weight fragments alone take 192 registers), the o_proj fragments behind them
    // (the q | k | v fragments are dead by then); both have landed by the time phase 2 is through.
    constexpr int CPR = DB_D / 8;                        // 16-byte chunks per K / V row
    constexpr int VITER = DB_BK * CPR / 64;              // 16-byte K (and V) loads per lane and chunk
    constexpr int OT = 2 * KSW;                          // o_proj column tiles per wave: hidden / 16 / 8
    constexpr bool KV_EARLY = KSW <= 3;
    u32x4 kr_cur[VITER], vr_cur[VITER];
    half8 wo_r[KV_ONLY ? 1 : OT][2];
    const int n_keys = q_slot0;                          // cached keys: every slot in front of this forward's rows
    const int n_chunks = (n_keys + DB_BK - 1) / DB_BK;
    const half_t* kbase = P.k_layer + (size_t)head * P.m * DB_D;
    const half_t* vbase = P.v_layer + (size_t)head * P.m * DB_D;
    auto issue_kv = [&](int chunk, u32x4 (&kr)[VITER], u32x4 (&vr)[VITER]) { db_issue_kv<DB_D>(chunk, n_keys, lane, kbase, vbase, kr, vr); };
    if constexpr (!KV_ONLY && KV_EARLY) {
        if (wave < n_chunks) issue_kv(wave, kr_cur, vr_cur);
        __builtin_amdgcn_sched_barrier(0);
    }
    floatx4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KSW; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[s], wr[t][s], acc[t], 0, 0, 0);
    if constexpr (!KV_ONLY) {
        if constexpr (!KV_EARLY) { if (wave < n_chunks) issue_kv(wave, kr_cur, vr_cur); }
        const char* obase = (const char*)P.wo;
        const int oks = P.n_heads * 2;                   // k-steps of o_proj's K = H D
#pragma unroll
        for (int j = 0; j < OT; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                wo_r[j][s] = *(const half8*)(obase + (((size_t)(wave * OT + j) * oks + head * 2 + s) * 64 + lane) * 16);
    }

    // the 8 K-partials meet in LDS: part[wave][row = 4 g + i][col = 16 t + r16]
    {
        float* mine = part + (size_t)wave * DB_BM * LDW;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) mine[(g * 4 + i) * LDW + t * 16 + r16] = acc[t][i];
    }
    __syncthreads();

    // ---- phase 2: sum in wave order, fp16 rounding, RoPE, K / V rows into the cache and LDS ---------------------------
    auto sum8 = [&](int row, int col, half8& out) {
        floatx4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < DB_WAVES; ++w) {
            const float* src = part + ((size_t)w * DB_BM + row) * LDW + col;
            lo += *(const floatx4*)src;
            hi += *(const floatx4*)(src + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[j] = (half_t)lo[j]; out[4 + j] = (half_t)hi[j]; }
    };
    if (wave < ROPE_WAVES) {
        const bool is_k = KV_ONLY || wave == 1;
        const int colbase = KV_ONLY ? 0 : wave * DB_D;                 // q at 0, k at 64 (KV_ONLY: k at 0)
        half8 x1, x2, o1, o2;
        sum8(rrow, colbase + rc * 8, x1);
        sum8(rrow, colbase + 32 + rc * 8, x2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // first half: x1*cos + (-x2)*sin ; second half: x2*cos + x1*sin -- every product and sum rounded to fp16
            const half_t a1 = (half_t)((float)x1[e] * (float)c1[e]);
            const half_t b1 = (half_t)((float)(-x2[e]) * (float)s1[e]);
            o1[e] = (half_t)((float)a1 + (float)b1);
            const half_t a2 = (half_t)((float)x2[e] * (float)c2[e]);
            const half_t b2 = (half_t)((float)x1[e] * (float)s2[e]);
            o2[e] = (half_t)((float)a2 + (float)b2);
        }
        half_t* dst_s = (is_k ? k_s : q_s) + rrow * DB_QS;
        *(half8*)(dst_s + rc * 8) = o1;
        *(half8*)(dst_s + 32 + rc * 8) = o2;
        if (is_k && q0 + rrow < P.q_len && rslot >= 0 && rslot < P.m) {
            half_t* dst = P.k_layer + ((size_t)head * P.m + rslot) * DB_D;
            *(half8*)(dst + rc * 8) = o1;
            *(half8*)(dst + 32 + rc * 8) = o2;
        }
    } else if (v_thread) {
        half8 x;
        sum8(vrow, (KV_ONLY ? DB_D : 2 * DB_D) + vc * 8, x);
        *(half8*)(v_s + vrow * DB_QS + vc * 8) = x;
        if (q0 + vrow < P.q_len && rslot >= 0 && rslot < P.m)
            *(half8*)(P.v_layer + ((size_t)head * P.m + rslot) * DB_D + vc * 8) = x;
    }
    if constexpr (KV_ONLY) return;
    __syncthreads();
    DB_STOP_AT(1, (float)q_s[tid])

    if constexpr (!KV_ONLY) {
        // ---- phase 3: attention of the tile's 16 queries over the cached keys + each query's own key ---------------------
        uint64_t* lds_bm = (uint64_t*)(db_lds + L::BM);
        float* lds_ml = (float*)(db_lds + L::ML);
        if (tid < DB_BM * DB_MAX_WORDS) lds_bm[tid] = bm_word;        // the ancestor-bitmask rows of the 16 queries
        // (db_attend's first barrier orders these stores, and the reads of the K-partials above, before the K / V tiles)
        db_attend<DB_D>(q_s, k_s, v_s, o_s, db_lds, lds_ml, lds_bm, kr_cur, vr_cur, kbase, vbase, n_keys, q0, P.q_len, q_slot0, gt,
                        P.n_tree, P.words, P.scale_log2e, tid, wave);

        DB_STOP_AT(4, (float)o_s[tid])
        // ---- phase 4: this head's slice of o_proj ---------------------------------------------------------------------
        half8 of[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) of[s] = *(const half8*)(o_s + r16 * DB_QS + s * 32 + g * 8);
        floatx4 c[OT];
#pragma unroll
        for (int j = 0; j < OT; ++j) {
            c[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(of[s], wo_r[j][s], c[j], 0, 0, 0);
        }
        // the [16][hidden] tile meets in LDS (the merge tiles are dead: every wave passed the barrier above after reading
        // them) and leaves as whole rows, 16 bytes per lane
        float* tile = (float*)db_lds;
        const int tstr = P.hidden + 4;
#pragma unroll
        for (int j = 0; j < OT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[(g * 4 + i) * tstr + (wave * OT + j) * 16 + r16] = c[j][i];
        __syncthreads();
        const int cpr = P.hidden >> 2;                               // float4 chunks per row
        for (int it = tid; it < DB_BM * cpr; it += DB_THREADS) {
            const int row = it / cpr, cc = it - row * cpr;
            if (q0 + row < P.q_len)
                *(floatx4*)(P.slab + ((size_t)head * P.q_len + q0 + row) * P.hidden + cc * 4) = *(const floatx4*)(tile + row * tstr + cc * 4);
        }
    }
}

template <int KSW, bool KV_ONLY>
static void db_go(const DbParams& P, int n_tiles, hipStream_t st) {
    using L = DbLds<KV_ONLY ? 8 : 12>;
    auto kern = draft_block_kernel<KSW, KV_ONLY>;
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL);
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_tiles * 16), dim3(DB_THREADS), L::TOTAL, st, P);
}

extern "C" int sq_draft_attn_block_f16(const void* a_frag, const void* wqkv_frag, const void* wo_frag, float* slab,
                                       size_t slab_bytes, void* k_layer, void* v_layer, const void* cos_tab,
                                       const void* sin_tab, const int64_t* d_position_ids, const int64_t* d_storage_ids,
                                       int q_len, int n_heads, int d, int hidden, int m, float scale, int q_slot0, int gt,
                                       int n_tree, const uint64_t* d_bitmask, int words, const int32_t* d_ctx, int kv_only,
                                       void* stream) {
    if (!a_frag || !wqkv_frag || !k_layer || !v_layer || !cos_tab || !sin_tab || !d_position_ids || !d_storage_ids)
        return SQ_EINVAL;
    if (q_len < 0 || n_heads <= 0 || hidden <= 0 || m <= 0 || n_tree < 1 || n_tree > SQ_MAX_TREE) return SQ_EINVAL;
    if (d != DB_D || n_heads > 16 || (hidden != 512 && hidden != 768 && hidden != 1024)) return SQ_EUNSUPPORTED;
    if (n_tree > 1 && (!d_bitmask || words < SQ_MASK_WORDS(n_tree) || words > DB_MAX_WORDS)) return SQ_EINVAL;
    if (n_tree == 1) { words = 1; d_bitmask = nullptr; }        // a one-node tree has no row a query could look up
    if (!kv_only) {
        if (!wo_frag || !slab || ((uintptr_t)slab & 15)) return SQ_EINVAL;
        if (slab_bytes < (size_t)n_heads * q_len * hidden * sizeof(float)) return SQ_EINVAL;
    }
    if (!d_ctx && (q_slot0 < 0 || q_slot0 > m || gt < 1)) return SQ_EINVAL;
    if (q_len == 0) return SQ_OK;
    DbParams P;
    P.a = (const half_t*)a_frag; P.wqkv = (const half_t*)wqkv_frag; P.wo = (const half_t*)wo_frag; P.slab = slab;
    P.k_layer = (half_t*)k_layer; P.v_layer = (half_t*)v_layer; P.cos_tab = (const half_t*)cos_tab;
    P.sin_tab = (const half_t*)sin_tab; P.position_ids = d_position_ids; P.storage_ids = d_storage_ids;
    P.bitmask = d_bitmask; P.ctx = d_ctx; P.words = words; P.n_tree = n_tree; P.q_slot0 = q_slot0; P.gt = gt < 1 ? 1 : gt;
    P.q_len = q_len; P.mtp = (q_len + 15) / 16; P.n_heads = n_heads; P.hidden = hidden; P.m = m;
    P.scale_log2e = scale * 1.4426950408889634f;
    const int n_tiles = P.mtp;
    hipStream_t st = (hipStream_t)stream;
    const int ksw = hidden / 256;
#define SQ_DB(K_)                                                                              \
    { if (kv_only) db_go<K_, true>(P, n_tiles, st); else db_go<K_, false>(P, n_tiles, st); }
    if (ksw == 2) SQ_DB(2) else if (ksw == 3) SQ_DB(3) else SQ_DB(4)
#undef SQ_DB
    return sq_check_launch();
}

// ---- RoPE + KV write + tree attention of a forward whose rows never see each other, ANY draft (heads of 64 or 128) -----------
// What it replaces: sq_rope_kv_write(_slabs)_f16 -> sq_tree_attention_f16, two dependent launches per layer (4.7 + 6.9 us on
// the 1.3B draft of configuration D, 120 layer passes per speculation step; 5.0 + 5.6 us on the 68m draft when the block above
// does not apply).  One workgroup per (query head, 16-row tile): the tile's q | k | v rows of that head are read from the
// projection's output -- fp16 rows, or the fp32 split-K partials of sq_linear_ts_f16, summed in split order and rounded to fp16
// like sq_rope_kv_write_slabs_f16 does --, rotated with the reference's fp16 roundings, the K / V rows go to their cache slots
// (by the first query head of a KV group) and, together with the rotated queries, into LDS; db_attend runs the attention over
// the cached keys + the row's own key; the output leaves as fp16 rows or as the fragment-major image of the o_proj operand.
// Rounding points are those of the two launches; the attention's fp32 summation order differs (the own key is folded last).
struct LaParams {
    const half_t* qkv;        // [q_len][stride] fp16 rows, or null
    const float* slab;        // [splits][q_len][stride] fp32 partials, or null
    int splits, stride;
    half_t* out;              // [q_len][H D] rows, or the fragment-major image (out_frag_mtp > 0)
    int out_frag_mtp;
    half_t* k_layer;          // [H_kv][M][D]
    half_t* v_layer;
    const half_t* cos_tab;
    const half_t* sin_tab;
    const int64_t* position_ids;
    const int64_t* storage_ids;
    const uint64_t* bitmask;
    const int32_t* ctx;
    int words, n_tree, q_slot0, gt;
    int q_len, n_heads, h_kv, m, xcd_span;
    float scale_log2e;
};

template <int D>
struct LaLds {
    using A = DbAtt<D>;
    static constexpr int QKV = A::MAIN;                                        // q_s (later o_s), k_s, v_s: [16][D + 8] halves each
    static constexpr int ML = QKV + 3 * DB_BM * A::QS * 2;
    static constexpr int BM = ML + DB_WAVES * DB_BM * 8;
    static constexpr int TOTAL = BM + DB_BM * DB_MAX_WORDS * 8;
};

template <int D, bool SLAB>
__global__ void __launch_bounds__(DB_THREADS) level_attention_kernel(const LaParams P) {
    using A = DbAtt<D>;
    using L = LaLds<D>;
    constexpr int QS = A::QS, VITER = A::VITER;
    constexpr int RW = D / 64;                           // waves per rotating role: a thread owns one 8-chunk of the first half + its mate
    extern __shared__ __attribute__((aligned(16))) unsigned char db_lds[];
    half_t* q_s = (half_t*)(db_lds + L::QKV);
    half_t* k_s = q_s + DB_BM * QS;
    half_t* v_s = k_s + DB_BM * QS;
    uint64_t* lds_bm = (uint64_t*)(db_lds + L::BM);
    float* lds_ml = (float*)(db_lds + L::ML);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (P.q_len + DB_BM - 1) / DB_BM;
    const int grp = P.n_heads / P.h_kv;
    int head, q_tile;
    if (!att_decode_block(blockIdx.x, P.n_heads, P.h_kv, n_tiles, P.xcd_span, &head, &q_tile)) return;
    const int q0 = q_tile * DB_BM;
    const int kvh = head / grp;
    int q_slot0 = P.q_slot0, gt = P.gt;
    if (P.ctx) { q_slot0 = P.ctx[0]; gt = P.ctx[1]; }
    if (gt < 1) gt = 1;
    if (q_slot0 < 0) q_slot0 = 0;
    if (q_slot0 > P.m) q_slot0 = P.m;
    const int n_keys = q_slot0;
    const half_t* kbase = P.k_layer + (size_t)kvh * P.m * D;
    const half_t* vbase = P.v_layer + (size_t)kvh * P.m * D;

    // the queries' ancestor-bitmask rows, this wave's first chunk of cached K / V rows: nothing here depends on q | k | v
    if (tid < DB_BM * DB_MAX_WORDS) {
        const int br = tid >> 3, bw = tid & 7;
        const int tn = q_slot0 + min(q0 + br, P.q_len - 1) - (gt - 1);
        uint64_t w = 0ull;
        if (bw < P.words && tn >= 1 && tn < P.n_tree && P.bitmask) w = P.bitmask[(size_t)tn * P.words + bw];
        lds_bm[tid] = w;
    }
    u32x4 kr_cur[VITER], vr_cur[VITER];
    if (wave * DB_BK < n_keys) db_issue_kv<D>(wave, n_keys, lane, kbase, vbase, kr_cur, vr_cur);

    // roles: waves [0, RW) rotate q, [RW, 2 RW) rotate k, [2 RW, 4 RW) move v (D = 64: waves 4-7 have no rows to move)
    const size_t split_stride = (size_t)P.q_len * P.stride;
    if (wave < 2 * RW) {
        const bool is_k = wave >= RW;
        const int lt = tid - (is_k ? RW * 64 : 0);
        const int row = lt & 15, c = lt >> 4;                        // chunk c of the first half, 0 .. D / 16 - 1
        const int ri = min(q0 + row, P.q_len - 1);
        const int64_t pos = P.position_ids[ri];
        const int64_t slot = P.storage_ids[ri];
        const size_t src = (size_t)ri * P.stride + (size_t)(is_k ? P.n_heads + kvh : head) * D;
        half8 x1, x2;
        if (SLAB) {
            const float* const sp[2] = {P.slab + src + c * 8, P.slab + src + D / 2 + c * 8};
            floatx4 lo[2], hi[2];
            slab_sum8<2>(sp, P.splits, split_stride, lo, hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x1[j] = (half_t)lo[0][j]; x1[4 + j] = (half_t)hi[0][j];
                x2[j] = (half_t)lo[1][j]; x2[4 + j] = (half_t)hi[1][j];
            }
        } else {
            x1 = *(const half8*)(P.qkv + src + c * 8);
            x2 = *(const half8*)(P.qkv + src + D / 2 + c * 8);
        }
        const half8 c1 = *(const half8*)(P.cos_tab + (size_t)pos * D + c * 8);
        const half8 c2 = *(const half8*)(P.cos_tab + (size_t)pos * D + D / 2 + c * 8);
        const half8 s1 = *(const half8*)(P.sin_tab + (size_t)pos * D + c * 8);
        const half8 s2 = *(const half8*)(P.sin_tab + (size_t)pos * D + D / 2 + c * 8);
        half8 o1, o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // first half: x1*cos + (-x2)*sin ; second half: x2*cos + x1*sin -- every product and sum rounded to fp16
            const half_t a1 = (half_t)((float)x1[e] * (float)c1[e]);
            const half_t b1 = (half_t)((float)(-x2[e]) * (float)s1[e]);
            o1[e] = (half_t)((float)a1 + (float)b1);
            const half_t a2 = (half_t)((float)x2[e] * (float)c2[e]);
            const half_t b2 = (half_t)((float)x1[e] * (float)s2[e]);
            o2[e] = (half_t)((float)a2 + (float)b2);
        }
        half_t* dst_s = (is_k ? k_s : q_s) + row * QS;
        *(half8*)(dst_s + c * 8) = o1;
        *(half8*)(dst_s + D / 2 + c * 8) = o2;
        if (is_k && head == kvh * grp && q0 + row < P.q_len && slot >= 0 && slot < P.m) {
            half_t* dst = P.k_layer + ((size_t)kvh * P.m + slot) * D;
            *(half8*)(dst + c * 8) = o1;
            *(half8*)(dst + D / 2 + c * 8) = o2;
        }
    } else if (wave < 4 * RW) {
        const int lt = tid - 2 * RW * 64;
        const int row = lt & 15, c = lt >> 4;                        // chunk c of the row, 0 .. D / 8 - 1
        const int ri = min(q0 + row, P.q_len - 1);
        const int64_t slot = P.storage_ids[ri];
        const size_t src = (size_t)ri * P.stride + (size_t)(P.n_heads + P.h_kv + kvh) * D + c * 8;
        half8 x;
        if (SLAB) {
            const float* const sp[1] = {P.slab + src};
            floatx4 lo[1], hi[1];
            slab_sum8<1>(sp, P.splits, split_stride, lo, hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = (half_t)lo[0][j]; x[4 + j] = (half_t)hi[0][j]; }
        } else {
            x = *(const half8*)(P.qkv + src);
        }
        *(half8*)(v_s + row * QS + c * 8) = x;
        if (head == kvh * grp && q0 + row < P.q_len && slot >= 0 && slot < P.m)
            *(half8*)(P.v_layer + ((size_t)kvh * P.m + slot) * D + c * 8) = x;
    }
    __syncthreads();

    half_t* o_s = q_s;                                   // (every wave holds its query fragments before the merge writes here)
    db_attend<D>(q_s, k_s, v_s, o_s, db_lds, lds_ml, lds_bm, kr_cur, vr_cur, kbase, vbase, n_keys, q0, P.q_len, q_slot0, gt,
                 P.n_tree, P.words, P.scale_log2e, tid, wave);

    // the tile's rows of this head: 16-byte chunks, row-major or fragment-major
    if (tid < DB_BM * (D / 8)) {
        const int row = tid / (D / 8), c = tid % (D / 8);
        if (q0 + row < P.q_len) {
            const half8 o = *(const half8*)(o_s + row * QS + c * 8);
            const int ocol = head * D + c * 8;
            half_t* dst = P.out_frag_mtp ? P.out + frag_chunk_offset(q0 + row, ocol >> 3, P.out_frag_mtp)
                                         : P.out + (size_t)(q0 + row) * (P.n_heads * D) + ocol;
            *(half8*)dst = o;
        }
    }
}

template <int D, bool SLAB>
static void la_go(const LaParams& P, int blocks, hipStream_t st) {
    using L = LaLds<D>;
    auto kern = level_attention_kernel<D, SLAB>;
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL);
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(DB_THREADS), L::TOTAL, st, P);
}

extern "C" int sq_level_attention_f16(const void* qkv, const float* qkv_slab, int splits, int qkv_stride, void* out, int out_frag,
                                      void* k_layer, void* v_layer, const void* cos_tab, const void* sin_tab,
                                      const int64_t* d_position_ids, const int64_t* d_storage_ids, int q_len, int n_heads,
                                      int h_kv, int d, int m, float scale, int q_slot0, int gt, int n_tree,
                                      const uint64_t* d_bitmask, int words, const int32_t* d_ctx, void* stream) {
    if ((!qkv && !qkv_slab) || !out || !k_layer || !v_layer || !cos_tab || !sin_tab || !d_position_ids || !d_storage_ids)
        return SQ_EINVAL;
    if (q_len < 0 || n_heads <= 0 || h_kv <= 0 || n_heads % h_kv || m <= 0 || n_tree < 1 || n_tree > SQ_MAX_TREE) return SQ_EINVAL;
    if (qkv_stride < (n_heads + 2 * h_kv) * d || (qkv_stride & 7)) return SQ_EINVAL;
    if (qkv_slab && (splits < 1 || ((uintptr_t)qkv_slab & 15))) return SQ_EINVAL;
    if (d != 64 && d != 128) return SQ_EUNSUPPORTED;
    if (out_frag && ((n_heads * d) & 31)) return SQ_EUNSUPPORTED;
    if (n_tree > 1 && (!d_bitmask || words < SQ_MASK_WORDS(n_tree) || words > DB_MAX_WORDS)) return SQ_EINVAL;
    if (n_tree == 1) { words = 1; d_bitmask = nullptr; }
    if (!d_ctx && (q_slot0 < 0 || q_slot0 > m || gt < 1)) return SQ_EINVAL;
    if (q_len == 0) return SQ_OK;
    LaParams P;
    P.qkv = (const half_t*)qkv; P.slab = qkv_slab; P.splits = splits; P.stride = qkv_stride; P.out = (half_t*)out;
    P.out_frag_mtp = out_frag ? (q_len + 15) / 16 : 0;
    P.k_layer = (half_t*)k_layer; P.v_layer = (half_t*)v_layer; P.cos_tab = (const half_t*)cos_tab;
    P.sin_tab = (const half_t*)sin_tab; P.position_ids = d_position_ids; P.storage_ids = d_storage_ids;
    P.bitmask = d_bitmask; P.ctx = d_ctx; P.words = words; P.n_tree = n_tree; P.q_slot0 = q_slot0; P.gt = gt < 1 ? 1 : gt;
    P.q_len = q_len; P.n_heads = n_heads; P.h_kv = h_kv; P.m = m;
    P.scale_log2e = scale * 1.4426950408889634f;
    const int n_tiles = (q_len + DB_BM - 1) / DB_BM;
    P.xcd_span = h_kv < 8 ? att_xcd_span(n_heads, h_kv, n_tiles) : 0;
    const int blocks = att_grid_blocks(n_heads, h_kv, n_tiles, P.xcd_span);
    hipStream_t st = (hipStream_t)stream;
    if (d == 128) { if (qkv_slab) la_go<128, true>(P, blocks, st); else la_go<128, false>(P, blocks, st); }
    else          { if (qkv_slab) la_go<64, true>(P, blocks, st);  else la_go<64, false>(P, blocks, st); }
    return sq_check_launch();
}
