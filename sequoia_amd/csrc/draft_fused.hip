// f1: fewer launches per level of a SMALL draft model (hidden <= 1024: the 68m / 160m drafts; <= 48 rows per forward).
//
// A draft level is launch-bound, not byte-bound (28 MB of layer weights, ~24 dependent launches at the
// launch floor).  The row-wise RMSNorm in front of qkv / gate_up / lm_head costs a launch each only because it needs whole
// rows; at this size the whole normalised activation block fits one workgroup's LDS (48 rows x 768 x 2 B = 72 KB), so
// every workgroup of the projection normalises the block ITSELF (a redundant 52 KB read from L2) and the norm launch
// disappears -- together with the residual-add launch in front of it once o_proj / down_proj write the residual stream
// through the tall-skinny kernel's "+ residual" epilogue (ts_linear.hip, splits == 1).  Per level of the 68m draft:
// 4 norm launches + the final norm + (optionally) the embedding launch are gone.
// MEASURED (profiles/r03_draft_fused_not_adopted.md): not faster than the unfused 19-launch sequence (96 vs 88 us at 34 rows)
// -- every unfused kernel already sits at the ~4.7 us floor of a dependent graph node and the fused body is the sum of the
// two dependency chains it replaces.  Opt-in (SEQUOIA_DRAFT_FUSED=1), kept with its tests as the record of the experiment.
//
//   dn_linear_kernel<MT, NT, EPI>:  out = epilogue( RMSNorm(x) * g  @  W^T )
//     prologue  x (row-major residual stream, or embed[ids] for the first layer) -> registers, all 4 waves: wave w takes the
//               k-steps w, w + 4, ...; lane (r = lane % 16, cg = lane / 16) holds chunk 4 i + cg of row 16 mt + r for every
//               row tile; sum of squares over the row's 4 lanes, then over the 4 waves through LDS (fixed order, fp32),
//               h(x * rstd) -> h(g * .) (the rounding points of Engine/Llama_modules.py:282-288), written fragment-major into
//               LDS (one conflict-free 1 KB store per wave instruction).  Straight-line per K (a switch over K / 128): a
//               predicate around the loads makes the compiler drain vmcnt at every join (measured: 14 vs 8 us per launch).
//               The wave's first weight loads are issued before the prologue;
//     main      the 4 waves split K; weights stream from the fragment-major image of ts_linear (same repack, same
//               HBM / L2 access pattern) straight into MFMA B registers, A fragments come from LDS;
//     epilogue  the 4 K-partials meet in LDS (the activation image is dead by then) and are summed in wave order:
//               PLAIN   fp16 rows [m][ldo]           (qkv -> sq_rope_kv_write_f16; lm_head -> the tree's draft_logits rows)
//               SWIGLU  h(h(silu(h(g))) * h(u)) as the fragment-major image of the down projection's input (:271)
// Replaces, for small drafts: LlamaRMSNorm before q/k/v_proj, gate/up_proj and lm_head (Engine/Llama_modules.py:282-288,
// 341-346; Engine/Llama_model.py:165-216,280-283) + embed_tokens (:151).
#include "common.h"

#define DN_WAVES 4
#define DN_THREADS (DN_WAVES * 64)
#define DN_MAX_K 1024                  // 32 chunks of 8 per lane
#define DN_MAX_MT 3                    // 48 rows

struct DnParams {
    const half_t* x;        // [m][k] residual stream (ids == null)
    const int64_t* ids;     // first layer: token ids [m]; row r of the stream is embed[ids[r]]
    const half_t* embed;    // [vocab][k]
    half_t* x_out;          // first layer: the residual stream, written by workgroup 0
    const half_t* g;        // norm weight [k]
    const half_t* w;        // fragment-major weights [n_tiles][k/32][64][8]  (SWIGLU: gate tiles, then up tiles)
    half_t* out;
    int vocab, m, mtp, n_out, k, ldo, units, tiles;
    float eps;
};

// Prologue of dn_linear_kernel: KPW k-steps per wave (k = 128 KPW), MT row tiles.  Lane (r = lane % 16, cg = lane / 16)
// of wave w holds, for every row tile, the chunks 4 i + cg of the k-steps i = w, w + 4, ... of row 16 mt + r.
template <int MT, int KPW>
__device__ __forceinline__ void dn_prologue(const DnParams& P, half_t* a_lds, float* ss_lds, int wave, int lane) {
    const int r16 = lane & 15, g4 = lane >> 4;
    half8 v[MT][KPW];
    const half_t* src[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = min(mt * 16 + r16, P.m - 1);             // rows beyond m re-read the last row; zeroed below
        if (P.ids) {
            int64_t id = P.ids[row];
            id = id < 0 ? 0 : (id >= P.vocab ? P.vocab - 1 : id);
            src[mt] = P.embed + (size_t)id * P.k;
        } else {
            src[mt] = P.x + (size_t)row * P.k;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < KPW; ++j) v[mt][j] = *(const half8*)(src[mt] + (((j * DN_WAVES + wave) * 4 + g4) * 8));
    half8 gw[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) gw[j] = *(const half8*)(P.g + (((j * DN_WAVES + wave) * 4 + g4) * 8));
    // sum of squares: this lane's chunks, the row's 4 lanes, then the 4 waves through LDS (fixed order)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < KPW; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += (float)v[mt][j][e] * (float)v[mt][j][e];
        const float s1 = ss + __shfl_xor(ss, 16, 64);
        const float s2 = s1 + __shfl_xor(s1, 32, 64);
        if (g4 == 0) ss_lds[wave * (DN_MAX_MT * 16) + mt * 16 + r16] = s2;
    }
    if (P.ids && P.x_out && blockIdx.x == 0) {                   // first layer: workgroup 0 also writes the residual stream
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + r16;
            if (row < P.m) {
#pragma unroll
                for (int j = 0; j < KPW; ++j)
                    *(half8*)(P.x_out + (size_t)row * P.k + (((j * DN_WAVES + wave) * 4 + g4) * 8)) = v[mt][j];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float* q = ss_lds + mt * 16 + r16;
        const float tot = ((q[0] + q[DN_MAX_MT * 16]) + q[2 * DN_MAX_MT * 16]) + q[3 * DN_MAX_MT * 16];
        const float inv = rsqrtf(tot / (float)P.k + P.eps);
        const bool live = mt * 16 + r16 < P.m;
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t n = (half_t)((float)v[mt][j][e] * inv);
                o[e] = (half_t)((float)gw[j][e] * (float)n);
            }
            if (!live) o = half8{0, 0, 0, 0, 0, 0, 0, 0};
            *(half8*)(a_lds + (((size_t)(j * DN_WAVES + wave) * MT + mt) * 64 + lane) * 8) = o;
        }
    }
}

template <int MT, int NT, bool SWIGLU>
__global__ void __launch_bounds__(DN_THREADS) dn_linear_kernel(const DnParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dn_lds[];
    half_t* a_lds = (half_t*)dn_lds;                         // [k/32][MT][64][8]
    float* m_lds = (float*)dn_lds;                           // merge: [wave][MT*16][LDW]   (after the main loop)
    constexpr int LDW = NT * 16 + 4;
    constexpr int TPU = SWIGLU ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g4 = lane >> 4;
    const int ksteps = P.k >> 5;

    // ---- this workgroup's column units, this wave's K range; the first weight loads go out BEFORE the prologue (they do not depend on it) ------------------------------------------------------------
    const int tile = blockIdx.x;
    const int u0 = (int)((long)tile * P.units / P.tiles), u1 = (int)((long)(tile + 1) * P.units / P.tiles);
    const int nu = u1 - u0;
    const int per = (ksteps + DN_WAVES - 1) / DN_WAVES;
    const int ks0 = min(ksteps, wave * per), ks1 = min(ksteps, ks0 + per);
    uint32_t woff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tt = min(t, nu * TPU - 1);
        const int wtile = u0 + tt / TPU + ((SWIGLU && (tt & 1)) ? P.units : 0);
        woff[t] = (uint32_t)wtile * (uint32_t)ksteps * 1024u + (uint32_t)lane * 16u;
    }
    const char* wbase = (const char*)P.w;
    floatx4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = floatx4{0.f, 0.f, 0.f, 0.f};

    constexpr int D = NT <= 2 ? 6 : (NT <= 4 ? 4 : 2);       // k-steps of weight loads in flight per wave
    half8 wr[D][NT];
    const int nst = ks1 - ks0;
#define DN_LOAD(d, I)                                                                                          \
    {                                                                                                          \
        const uint32_t kw_ = (uint32_t)(ks0 + max(0, min((I), nst - 1))) << 10;                                \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                         \
            wr[d][t] = __builtin_nontemporal_load((const half8*)(wbase + (woff[t] + kw_)));                    \
    }
#define DN_MMA(d, I)                                                                                           \
    {                                                                                                          \
        const half_t* ap_ = a_lds + ((size_t)(ks0 + (I)) * MT * 64 + lane) * 8;                                \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                    \
            const half8 af_ = *(const half8*)(ap_ + mt * 512);                                                 \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                     \
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_, wr[d][t], acc[mt][t], 0, 0, 0);       \
        }                                                                                                      \
    }
#pragma unroll
    for (int d = 0; d < D; ++d) DN_LOAD(d, d);               // (an empty K range re-reads k-step ks0 - harmless, never used)

    // ---- prologue: normalise the activation block into LDS; all 4 waves, wave w takes k-steps w, w + 4, ... ----------------
    // (straight-line per K: a predicate around the loads would make the compiler drain vmcnt at every join)
    float* ss_lds = (float*)(dn_lds + 160 * 1024 - DN_WAVES * DN_MAX_MT * 16 * sizeof(float));     // [wave][MT * 16]
    switch (ksteps >> 2) {
        case 2: dn_prologue<MT, 2>(P, a_lds, ss_lds, wave, lane); break;
        case 4: dn_prologue<MT, 4>(P, a_lds, ss_lds, wave, lane); break;
        case 6: dn_prologue<MT, 6>(P, a_lds, ss_lds, wave, lane); break;
        default: dn_prologue<MT, 8>(P, a_lds, ss_lds, wave, lane); break;
    }
    __syncthreads();

    if (ks0 < ks1) {
        const int nfull = nst / D, rem = nst - nfull * D;
        int i = 0;
        for (int it = 0; it < nfull; ++it, i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                DN_MMA(d, i + d);
                DN_LOAD(d, i + D + d);
            }
        }
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (d < rem) DN_MMA(d, i + d);
#undef DN_LOAD
#undef DN_MMA
    }
    __syncthreads();                                           // every wave is done reading the activation image

    // ---- merge the 4 K-partials in LDS, epilogue ------------------------------------------------------------------------------
    float* mine = m_lds + (size_t)wave * (MT * 16) * LDW;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) mine[(mt * 16 + g4 * 4 + i) * LDW + t * 16 + r16] = acc[mt][t][i];
    __syncthreads();
    const int groups = nu * 2;                                 // (row, 8-column group) items; a unit holds 2
    const int items = P.m * groups;
    for (int it = tid; it < items; it += DN_THREADS) {
        const int row = it / groups, grp = it % groups;
        const int unit = grp >> 1, half = grp & 1;
        const int col = (unit * TPU) * 16 + half * 8;
        float v[8], u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = 0.f; u[j] = 0.f; }
#pragma unroll
        for (int wv = 0; wv < DN_WAVES; ++wv) {
            const float* src = m_lds + ((size_t)wv * (MT * 16) + row) * LDW + col;
            const floatx4 a = *(const floatx4*)src, b = *(const floatx4*)(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += a[j]; v[4 + j] += b[j]; }
            if (SWIGLU) {
                const floatx4 p = *(const floatx4*)(src + 16), q = *(const floatx4*)(src + 20);
#pragma unroll
                for (int j = 0; j < 4; ++j) { u[j] += p[j]; u[4 + j] += q[j]; }
            }
        }
        const int ocol = (u0 + unit) * 16 + half * 8;
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t h = (half_t)v[j];
            if (SWIGLU) {
                const float gf = (float)h;
                const half_t sg = (half_t)(gf / (1.0f + expf(-gf)));
                h = (half_t)((float)sg * (float)(half_t)u[j]);
            }
            o[j] = h;
        }
        if (SWIGLU) *(half8*)(P.out + frag_chunk_offset(row, ocol >> 3, P.mtp)) = o;
        else *(half8*)(P.out + (size_t)row * P.ldo + ocol) = o;
    }
}

static size_t dn_lds_bytes(int mt, int nt, int k) {
    // the prologue's cross-wave sums sit at the END of the 160 KB window (fixed address: the kernel does not need the
    // size), so the launch always asks for the whole window; the activation image / the merge tiles start at 0
    (void)mt; (void)nt; (void)k;
    return 160 * 1024;
}

template <int MT, int NT, bool SWIGLU>
static void dn_go(const DnParams& P, hipStream_t st) {
    const size_t lds = dn_lds_bytes(MT, NT, P.k);
    auto kern = dn_linear_kernel<MT, NT, SWIGLU>;
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(P.tiles), dim3(DN_THREADS), lds, st, P);
}

template <int MT>
static int dn_dispatch(const DnParams& P, int swiglu, int nt, hipStream_t st) {
    if (swiglu) {
        if (nt <= 2) dn_go<MT, 2, true>(P, st);
        else if (nt <= 4) dn_go<MT, 4, true>(P, st);
        else return SQ_EUNSUPPORTED;
        return SQ_OK;
    }
    if (nt <= 1) dn_go<MT, 1, false>(P, st);
    else if (nt <= 2) dn_go<MT, 2, false>(P, st);
    else if (nt <= 4) dn_go<MT, 4, false>(P, st);
    else if (nt <= 8) dn_go<MT, 8, false>(P, st);
    else return SQ_EUNSUPPORTED;
    return SQ_OK;
}

extern "C" int sq_norm_linear_f16(const void* x, const int64_t* d_ids, const void* embed, int vocab, void* x_out,
                                  const void* norm_weight, float eps, const void* w_frag, void* out, int ldo, int m, int n_out,
                                  int k, int swiglu, int tiles, void* stream) {
    if (!norm_weight || !w_frag || !out || m <= 0 || n_out <= 0 || k <= 0 || tiles < 1) return SQ_EINVAL;
    if (!d_ids && !x) return SQ_EINVAL;
    if (d_ids && (!embed || vocab <= 0)) return SQ_EINVAL;
    if (m > DN_MAX_MT * 16 || k > DN_MAX_K || (k & 255) || (n_out & 15) || (swiglu && (n_out & 31)) || (!swiglu && (ldo < n_out || (ldo & 7))) ||
        ((uintptr_t)w_frag & 15) || ((uintptr_t)out & 15) || (x && ((uintptr_t)x & 15)) || ((uintptr_t)norm_weight & 15) ||
        (size_t)(swiglu ? 2 : 1) * n_out * k * 2 >= (1ull << 32))
        return SQ_EUNSUPPORTED;
    DnParams P;
    P.x = (const half_t*)x; P.ids = d_ids; P.embed = (const half_t*)embed; P.vocab = vocab; P.x_out = (half_t*)x_out;
    P.g = (const half_t*)norm_weight; P.w = (const half_t*)w_frag; P.out = (half_t*)out;
    P.m = m; P.mtp = (m + 15) / 16; P.n_out = n_out; P.k = k; P.ldo = ldo; P.eps = eps;
    P.units = n_out / 16;
    P.tiles = tiles > P.units ? P.units : tiles;
    const int max_units = (P.units + P.tiles - 1) / P.tiles;
    const int nt = max_units * (swiglu ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (P.mtp <= 1) rc = dn_dispatch<1>(P, swiglu, nt, st);
    else if (P.mtp == 2) rc = dn_dispatch<2>(P, swiglu, nt, st);
    else rc = dn_dispatch<3>(P, swiglu, nt, st);
    if (rc != SQ_OK) return rc;
    return sq_check_launch();
}
