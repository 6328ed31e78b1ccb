// a3/a4: tree-batched attention for one layer, MFMA (v_mfma_f32_16x16x32_f16) for both
// contractions, fp32 online softmax, tree-causal mask evaluated from the ancestor bitmask
// (no dense mask in HBM) or, for drop-in callers, from the reference's dense additive mask.
//
// Decomposition (latency-bound problem: per layer only ~6 MB of K/V/Q/O, 0.5 GFLOP):
//   grid  = one workgroup per (16-query tile, head) -> 256 workgroups for the 7B verify (q = 128,
//           H = 32), one per CU.  The linear block id is decoded XCD-aware: the dispatcher places
//           block b on XCD b % 8, so all query tiles of a head are given ids with the same b % 8 and
//           the head's K/V is fetched into ONE XCD's L2 instead of all eight.
//   block = 8 waves; wave w walks the 32-key chunks w, w+8, ... of the KV range (split-KV inside
//           the workgroup, <= 2 chunks per wave up to 512 keys, next chunk's K/V prefetched into
//           registers while the current one is computed) with its own online-softmax state; the
//           partial results are merged through LDS at the end.
// Per 32-key chunk and wave:
//   S^T = K Q^T   (A = K rows straight from HBM as 16-byte fragment loads, B = Q, held in VGPRs)
//         -> lane (q = lane&15, g = lane>>4) holds keys {16t + 4g + r}: exactly the A-operand
//            layout the P V MFMA wants (k-slot 8g + j <-> key 16(j>>2) + 4g + (j&3)), so P never
//            leaves registers.
//   O^T += V^T P^T (A = V^T through a wave-private LDS tile read back transposed with
//                  ds_read_b64_tr_b16, B = P^T = the registers S^T left behind): the accumulator
//                  column is again the lane's own query, so the online-softmax rescale is lane-local.
#include "common.h"
#include "attn_map.h"
#include <stdlib.h>

#define ATT_WAVES 8
#define ATT_THREADS (ATT_WAVES * 64)
#define ATT_BK 32      // keys per chunk

typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct AttnParams {
    const half_t* q;        // [H][q_len][D]
    const half_t* k;        // [H_kv][M][D]
    const half_t* v;        // [H_kv][M][D]
    half_t* out;            // [q_len][H*D], or its fragment-major image (out_frag_mtp = ceil(q_len / 16) > 0)
    int out_frag_mtp;
    int q_len, n_heads, h_kv, m, kv_len;
    float scale_log2e;      // scale * log2(e)
    int mask_mode;          // 0 dense additive, 1 implicit tree
    const half_t* dense;    // [q_len][mask_stride]
    int mask_stride;
    int q_slot0, gt, n_tree;
    const uint64_t* bitmask;
    int words;
    const int32_t* ctx;     // optional device override of {q_slot0, gt, kv_len} (hipGraph replays)
    int xcd_span;           // h_kv < 8: XCDs one KV head's workgroups are dealt over (host: att_xcd_span)
};

// max / sum over the four 16-lane groups holding the same query (lanes l, l^16, l^32, l^48) with the
// gfx950 VALU lane swaps (v_permlane16_swap / v_permlane32_swap) instead of LDS-routed ds_bpermute
// (inline asm: with ROCm 7.2's clang the __builtin_amdgcn_permlane*_swap builtins return the first
// result twice; the s_nop's are the VALU-write -> permlane-read wait states hipcc would insert.)
__device__ __forceinline__ void lane_swap16(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap32(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float group4_max(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    lane_swap16(a, b);                   // a = rows [0,0,2,2], b = rows [1,1,3,3]
    v = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
    a = __builtin_bit_cast(unsigned, v); b = a;
    lane_swap32(a, b);                   // a = lower half twice, b = upper half twice
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float group4_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    lane_swap16(a, b);
    v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    a = __builtin_bit_cast(unsigned, v); b = a;
    lane_swap32(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// MASK: 0 = dense additive mask, 1 = tree mask with the bitmask row in two registers (n <= 128),
// 2 = tree mask with the bitmask rows staged in LDS (n <= 512).
// (A variant that also applied RoPE and wrote the new K/V rows -- one launch instead of sq_rope_kv_write_f16 + this
// kernel -- was bit-identical but slower on every measured shape: 13.6 vs 8.4 + 2.5 us on the 7B verify layer, 256 VGPRs
// with spills; removed in round 2.)
template <int D, int MASK>
__global__ void __launch_bounds__(ATT_THREADS) tree_attention_kernel(AttnParams P) {
    if (P.ctx) {   // uniform: step-dependent scalars live in device memory so the launch is graph-replayable
        P.q_slot0 = P.ctx[0]; P.gt = P.ctx[1]; P.kv_len = P.ctx[2];
        if (P.kv_len > P.m) P.kv_len = P.m;
        if (P.kv_len < 1) P.kv_len = 1;
    }
    constexpr int DSTEPS = D / 32;       // MFMA k-steps for S^T
    constexpr int NT = D / 16;           // 16-wide output column tiles
    constexpr int VSTRIDE = D + 8;       // halves; +16 B keeps ds_write_b128 aligned and de-phases banks
    constexpr int V_TILE = 2 * ATT_BK * VSTRIDE;                // halves per wave: V tile then K tile
    constexpr int OSTR = D + 4;                                 // padded row of the merge tile (floats): 16-byte stores from 8 lanes hit 8 distinct bank quads
    constexpr int O_TILE = ATT_BM * OSTR;                       // floats per wave
    // LDS: V staging (per wave) is reused for the cross-wave merge of O.
    constexpr int LDS_BYTES_V = ATT_WAVES * V_TILE * 2;
    constexpr int LDS_BYTES_O = ATT_WAVES * O_TILE * 4;
    constexpr int LDS_MAIN = LDS_BYTES_V > LDS_BYTES_O ? LDS_BYTES_V : LDS_BYTES_O;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_MAIN + ATT_WAVES * ATT_BM * 8 + ATT_BM * 8 * SQ_MAX_TREE / 64];
    half_t* lds_v = (half_t*)lds_raw;
    float* lds_o = (float*)lds_raw;
    float* lds_ml = (float*)(lds_raw + LDS_MAIN);                          // [wave][16][2] (m, l)
    uint64_t* lds_bm = (uint64_t*)(lds_raw + LDS_MAIN + ATT_WAVES * ATT_BM * 8);  // [16][words]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int qc = lane & 15, g = lane >> 4;
    // XCD-aware decode (att_decode_block): every query tile of a head -- of a KV head, under GQA -- shares its XCD's L2
    const int n_tiles = (P.q_len + ATT_BM - 1) / ATT_BM;
    const int grp = P.n_heads / P.h_kv;
    int head, q_tile;
    if (!att_decode_block(blockIdx.x, P.n_heads, P.h_kv, n_tiles, P.xcd_span, &head, &q_tile)) return;     // uniform per block
    const int q0 = q_tile * ATT_BM;
    const int kvh = head / grp;
    const half_t* kbase = P.k + (size_t)kvh * P.m * D;
    const half_t* vbase = P.v + (size_t)kvh * P.m * D;

    // this lane's query row (softmax side): q0 + qc
    const int qi = q0 + qc;
    const bool q_valid = qi < P.q_len;
    const int qi_c = q_valid ? qi : P.q_len - 1;
    const int slot = P.q_slot0 + qi_c;
    const int tnode = slot - (P.gt - 1);

    // this lane's ancestor-bitmask row: trees up to 128 nodes (2 words) live in two registers, larger
    // trees are staged in LDS
    uint64_t bm0 = 0ull, bm1 = 0ull;
    if constexpr (MASK == 1) {
        if (tnode >= 1 && tnode < P.n_tree) {
            bm0 = P.bitmask[(size_t)tnode * P.words];
            if (P.words == 2) bm1 = P.bitmask[(size_t)tnode * P.words + 1];
        }
    }
    // branch-free visibility test: per-lane constants hoisted out of the key loop
    const bool causal_row = slot < P.gt;                 // committed-text query: plain causal row
    const bool tree_ok = tnode < P.n_tree;
    if constexpr (MASK == 2) {
        // stage the ancestor-bitmask rows of the 16 queries
        for (int i = tid; i < ATT_BM * P.words; i += ATT_THREADS) {
            const int r = i / P.words, w = i - r * P.words;
            int qq = q0 + r; if (qq >= P.q_len) qq = P.q_len - 1;
            const int tn = P.q_slot0 + qq - (P.gt - 1);
            lds_bm[i] = (tn >= 1 && tn < P.n_tree) ? P.bitmask[(size_t)tn * P.words + w] : 0ull;
        }
    }

    // Q fragments (B operand): lane (n = qc, g) holds Q[qi][32*s + 8*g .. +8]
    half8 qf[DSTEPS];
    {
        const half_t* qrow = P.q + ((size_t)head * P.q_len + qi_c) * D;
#pragma unroll
        for (int s = 0; s < DSTEPS; ++s) qf[s] = *(const half8*)(qrow + s * 32 + g * 8);
    }
    __syncthreads();

    float m_run = -INFINITY, l_run = 0.f;
    floatx4 o_acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) o_acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};

    half_t* my_v = lds_v + wave * V_TILE;
    const int n_chunks = (P.kv_len + ATT_BK - 1) / ATT_BK;
    constexpr int CPR = D / 8;                      // 16-byte chunks per K/V row
    constexpr int VITER = ATT_BK * CPR / 64;        // 16-byte V loads per lane and chunk

    // K fragments (A operand: lane (key = qc, g) holds K[key][32 s + 8 g .. +8]) and the V rows of a
    // chunk, straight from HBM/L2 into registers.  Two register sets: the next chunk is in flight
    // while the current one is computed.
    u32x4 kr_cur[VITER], kr_nxt[VITER];
    u32x4 vr_cur[VITER], vr_nxt[VITER];
    half_t* my_k = my_v + ATT_BK * VSTRIDE;
    // Both K and V rows are fetched as whole 16-byte-per-lane row segments (a wave instruction
    // covers 4 (D=128) or 8 (D=64) complete rows = full cache lines) and staged through the wave's
    // private LDS tiles; fragment-shaped global loads (16 rows x 64 B per instruction) cost ~2x.
    auto issue_loads = [&](int chunk, u32x4 (&kr)[VITER], u32x4 (&vr)[VITER]) {
        const int key0 = chunk * ATT_BK;
#pragma unroll
        for (int it = 0; it < VITER; ++it) {
            const int idx = it * 64 + lane;
            const int r = idx / CPR, c = idx % CPR;
            int row = key0 + r; if (row >= P.kv_len) row = P.kv_len - 1;
            kr[it] = *(const u32x4*)(kbase + (size_t)row * D + c * 8);
            vr[it] = *(const u32x4*)(vbase + (size_t)row * D + c * 8);
        }
    };

    int ch = wave;
    if (ch < n_chunks) issue_loads(ch, kr_cur, vr_cur);
    for (; ch < n_chunks; ch += ATT_WAVES) {
        const int key0 = ch * ATT_BK;
        const bool has_next = ch + ATT_WAVES < n_chunks;
        if (has_next) issue_loads(ch + ATT_WAVES, kr_nxt, vr_nxt);
        // ---- K, V chunk -> wave-private LDS tiles (row-major, padded) ---------------------------------
#pragma unroll
        for (int it = 0; it < VITER; ++it) {
            const int idx = it * 64 + lane;
            *(u32x4*)(my_k + (idx / CPR) * VSTRIDE + (idx % CPR) * 8) = kr_cur[it];
            *(u32x4*)(my_v + (idx / CPR) * VSTRIDE + (idx % CPR) * 8) = vr_cur[it];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- S^T = K Q^T: A fragment = K[key = 16t + qc][32s + 8g .. +8] via ds_read_b128 (row stride
        //      D+8 halves = 68 dwords -> the 16 lanes of a group hit 16 distinct 16-byte bank slots) ------
        floatx4 s_acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            half8 kf[DSTEPS];
#pragma unroll
            for (int s = 0; s < DSTEPS; ++s) kf[s] = *(const half8*)(my_k + (t * 16 + qc) * VSTRIDE + s * 32 + g * 8);
#pragma unroll
            for (int s = 0; s < DSTEPS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[s], qf[s], acc, 0, 0, 0);
            s_acc[t] = acc;
        }
        // ---- mask + online softmax (this lane: query qc, keys key0 + 16t + 4g + r) -----------------
        float sv[8];
        float cmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + t * 16 + g * 4 + r;
                float x = s_acc[t][r] * P.scale_log2e;
                const bool in_len = key < P.kv_len;
                if constexpr (MASK == 0) {
                    const int kc = in_len ? key : 0;
                    const float mk = (float)P.dense[(size_t)qi_c * P.mask_stride + kc] * 1.4426950408889634f;
                    x = in_len ? x + mk : -INFINITY;
                } else {
                    // tree rule without branches: every lane evaluates both the causal and the tree
                    // predicate and selects (divergent if/else chains cost ~40 cycles per branch here)
                    const int j = key - (P.gt - 1);
                    uint64_t wv;
                    if constexpr (MASK == 1) wv = (j & 64) ? bm1 : bm0;
                    else wv = lds_bm[qc * P.words + (((unsigned)j >> 6) < (unsigned)P.words ? (j >> 6) : 0)];
                    const bool bit = (wv >> (j & 63)) & 1ull;
                    const bool vis_tree = (key < P.gt) | (tree_ok & ((unsigned)j < (unsigned)P.n_tree) & bit);
                    const bool vis = in_len & (causal_row ? (key <= slot) : vis_tree);
                    x = vis ? x : -INFINITY;
                }
                sv[t * 4 + r] = x;
                cmax = fmaxf(cmax, x);
            }
        }
        cmax = group4_max(cmax);
        const float m_new = fmaxf(m_run, cmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;      // fully masked so far
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // exp2(-inf) = 0 on the first chunk
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
        // ---- O^T += V^T P^T: A = V^T fragment (rows = d), B = P^T (n = this lane's query), so the
        //      accumulator column is the lane's own query and the rescale by alpha is lane-local ------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // B fragment: k-slot 8g + j  <->  key row 16(j>>2) + 4g + (j&3), column 16nt + qc.
            // ds_read_b64_tr_b16: lane i of a 16-lane group supplies 4 contiguous halves of row
            // (i>>2), columns 4(i&3).., and receives column i of the group's 4x16 block.
            const half_t* a0 = my_v + (0 * 16 + g * 4 + (qc >> 2)) * VSTRIDE + nt * 16 + (qc & 3) * 4;
            const half_t* a1 = my_v + (1 * 16 + g * 4 + (qc >> 2)) * VSTRIDE + nt * 16 + (qc & 3) * 4;
            half8 vf;
            const fp16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)a0);
            const fp16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)a1);
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

    // ---- merge the waves ---------------------------------------------------------------------
    l_run = group4_sum(l_run);
    __syncthreads();                       // everyone is done with the V staging area
    if (g == 0) { lds_ml[(wave * ATT_BM + qc) * 2] = m_run; lds_ml[(wave * ATT_BM + qc) * 2 + 1] = l_run; }
    // o_acc[nt][r] = O[q = qc][d = 16 nt + 4 g + r]: four consecutive d per lane -> one 16-byte LDS store
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        *(floatx4*)(lds_o + (wave * ATT_BM + qc) * OSTR + nt * 16 + g * 4) = o_acc[nt];
    __syncthreads();
    {
        constexpr int EPT_O = ATT_BM * D / ATT_THREADS;       // 4 (D=128) or 2 (D=64)
        const int row = tid / (D / EPT_O);
        const int col = (tid % (D / EPT_O)) * EPT_O;
        float mw[ATT_WAVES], lw[ATT_WAVES], mmax = -INFINITY;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) {
            mw[w] = lds_ml[(w * ATT_BM + row) * 2];
            lw[w] = lds_ml[(w * ATT_BM + row) * 2 + 1];
            mmax = fmaxf(mmax, mw[w]);
        }
        float denom = 0.f, wgt[ATT_WAVES];
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) {
            wgt[w] = (mw[w] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw[w] - mmax);
            denom += lw[w] * wgt[w];
        }
        const float inv = denom > 0.f ? 1.0f / denom : 0.f;
        if (q0 + row < P.q_len) {
            const int ocol = head * D + col;
            half_t* dst = P.out_frag_mtp ? P.out + frag_chunk_offset(q0 + row, ocol >> 3, P.out_frag_mtp) + (ocol & 7)
                                         : P.out + (size_t)(q0 + row) * (P.n_heads * D) + ocol;
            float res[EPT_O];
#pragma unroll
            for (int e = 0; e < EPT_O; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < ATT_WAVES; ++w) acc += lds_o[(w * ATT_BM + row) * OSTR + col + e] * wgt[w];
                res[e] = acc * inv;
            }
            if constexpr (EPT_O == 8) {
                half8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = (half_t)res[e];
                *(half8*)dst = o8;
            } else if constexpr (EPT_O == 4) {
                half4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (half_t)res[e];
                *(half4*)dst = o4;
            } else {
                static_assert(EPT_O == 2, "unexpected tile/thread ratio");
                half2v o2;
                o2[0] = (half_t)res[0]; o2[1] = (half_t)res[1];
                *(half2v*)dst = o2;
            }
        }
    }
}

// Host-side view of the launch's block map (no device work): the grid size of sq_tree_attention_f16 for these head counts,
// and, for 0 <= block < *n_blocks, the (head, 16-query tile) the block computes (-1, -1: a block without work) and the XCD
// it is expected on (block % 8).
extern "C" int sq_tree_attention_block_decode(int block, int q_len, int n_heads, int h_kv, int* head, int* tile, int* n_blocks) {
    if (q_len <= 0 || n_heads <= 0 || h_kv <= 0 || n_heads % h_kv || !n_blocks) return SQ_EINVAL;
    const int n_tiles = (q_len + ATT_BM - 1) / ATT_BM;
    const int span = h_kv < 8 ? att_xcd_span(n_heads, h_kv, n_tiles) : 0;
    *n_blocks = att_grid_blocks(n_heads, h_kv, n_tiles, span);
    if (block < 0 || block >= *n_blocks || !head || !tile) return block < 0 || !head || !tile ? SQ_OK : SQ_EINVAL;
    if (!att_decode_block(block, n_heads, h_kv, n_tiles, span, head, tile)) { *head = -1; *tile = -1; }
    return SQ_OK;
}

extern "C" int sq_tree_attention_f16(const void* q, const void* k_layer, const void* v_layer, void* out, int q_len,
                                     int n_heads, int h_kv, int d, int m, int kv_len, float scale, int mask_mode,
                                     const void* dense_mask, int mask_stride, int q_slot0, int gt, int n_tree,
                                     const uint64_t* d_bitmask, int words, const int32_t* d_ctx, void* stream) {
    if (!q || !k_layer || !v_layer || !out || q_len < 0 || n_heads <= 0 || h_kv <= 0 || m <= 0) return SQ_EINVAL;
    if (d_ctx) { if (kv_len <= 0) kv_len = 1; if (gt < 1) gt = 1; }
    if (kv_len <= 0 || kv_len > m || n_heads % h_kv) return SQ_EINVAL;
    if (d != 64 && d != 128) return SQ_EUNSUPPORTED;
    const int out_frag = mask_mode & SQ_ATT_OUT_FRAG;
    mask_mode &= ~SQ_ATT_OUT_FRAG;
    if (out_frag && ((n_heads * d) & 31)) return SQ_EUNSUPPORTED;
    AttnParams P;
    P.out_frag_mtp = out_frag ? (q_len + 15) / 16 : 0;
    P.q = (const half_t*)q; P.k = (const half_t*)k_layer; P.v = (const half_t*)v_layer; P.out = (half_t*)out;
    P.q_len = q_len; P.n_heads = n_heads; P.h_kv = h_kv; P.m = m; P.kv_len = kv_len;
    P.scale_log2e = scale * 1.4426950408889634f;
    P.mask_mode = mask_mode;
    P.dense = (const half_t*)dense_mask; P.mask_stride = mask_stride;
    P.q_slot0 = q_slot0; P.gt = gt; P.n_tree = n_tree; P.bitmask = d_bitmask; P.words = words; P.ctx = d_ctx;
    if (mask_mode == 0) {
        if (!dense_mask || mask_stride < kv_len) return SQ_EINVAL;
    } else if (mask_mode == 1) {
        if (gt < 1 || n_tree < 1 || n_tree > SQ_MAX_TREE) return SQ_EINVAL;
        if (n_tree > 1 && (!d_bitmask || words < SQ_MASK_WORDS(n_tree) || words > SQ_MAX_TREE / 64)) return SQ_EINVAL;
        if (n_tree == 1) { P.words = words > 0 ? words : 1; }
    } else {
        return SQ_EINVAL;
    }
    if (q_len == 0) return SQ_OK;
    const int n_tiles = (q_len + ATT_BM - 1) / ATT_BM;
    P.xcd_span = h_kv < 8 ? att_xcd_span(n_heads, h_kv, n_tiles) : 0;
    dim3 grid(att_grid_blocks(n_heads, h_kv, n_tiles, P.xcd_span)), block(ATT_THREADS);
    hipStream_t st = (hipStream_t)stream;
    const int mk = mask_mode == 0 ? 0 : (P.words <= 2 ? 1 : 2);
#define SQ_ATT(DD, MM) hipLaunchKernelGGL((tree_attention_kernel<DD, MM>), grid, block, 0, st, P)
    if (d == 128) { if (mk == 0) SQ_ATT(128, 0); else if (mk == 1) SQ_ATT(128, 1); else SQ_ATT(128, 2); }
    else          { if (mk == 0) SQ_ATT(64, 0);  else if (mk == 1) SQ_ATT(64, 1);  else SQ_ATT(64, 2); }
#undef SQ_ATT
    return sq_check_launch();
}
