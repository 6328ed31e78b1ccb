// Row-wise fused elementwise ops of the Llama block around the attention hot path: RMSNorm
// (optionally fused with the residual add) and SiLU(gate)*up.  They exist to remove launches
// from the launch-bound draft forward; each is one pass over the row with
// 16-byte accesses.  Rounding points follow the reference's fp16/fp32 expression
// (Engine/Llama_modules.py:274-288: fp32 variance, cast to fp16, fp16 multiply by the weight;
// :271: act_fn(gate) * up in fp16).
#include "common.h"

#define ROW_THREADS 256
#define ROW_WAVES (ROW_THREADS / 64)

// x: [rows][hidden] fp16; optional residual add: h = x + res (fp16 add), h written back to res_out.
// FRAG: `out` is the fragment-major activation image of the tall-skinny linear layer (frag_chunk_offset, common.h)
template <bool ADD, bool FRAG>
__global__ void __launch_bounds__(ROW_THREADS)
rmsnorm_kernel(const half_t* __restrict__ x, const half_t* res, half_t* sum_out,   // res and sum_out may alias
               const half_t* __restrict__ w, half_t* __restrict__ out, int hidden, float eps, int mtp) {
    __shared__ float s_f[ROW_WAVES];
    const size_t row = blockIdx.x;
    const half_t* xr = x + row * hidden;
    const int chunks = hidden >> 3;
    float ss = 0.f;
    // pass 1: (optional add) + sum of squares; rows are small (<= 16 KB) and stay in L1/L2 for pass 2
    for (int c = threadIdx.x; c < chunks; c += ROW_THREADS) {
        half8 v = *(const half8*)(xr + c * 8);
        if (ADD) {
            const half8 r = *(const half8*)(res + row * hidden + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] + (float)r[j]);
            *(half8*)(sum_out + row * hidden + c * 8) = v;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
    }
    const float tot = block_sum_f32<ROW_WAVES>(ss, s_f);
    const float inv = rsqrtf(tot / (float)hidden + eps);
    const half_t* src = ADD ? (const half_t*)(sum_out + row * hidden) : xr;
    for (int c = threadIdx.x; c < chunks; c += ROW_THREADS) {
        // ADD: re-read what this same thread wrote above (same c) -> program order suffices
        const half8 v = *(const half8*)(src + c * 8);
        const half8 wv = *(const half8*)(w + c * 8);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const half_t n = (half_t)((float)v[j] * inv);
            o[j] = (half_t)((float)wv[j] * (float)n);
        }
        *(half8*)(out + (FRAG ? frag_chunk_offset(row, c, mtp) : row * hidden + c * 8)) = o;
    }
}

extern "C" int sq_rmsnorm_f16(const void* x, const void* weight, void* out, int rows, int hidden, float eps,
                              void* stream) {
    if (!x || !weight || !out || rows < 0 || hidden <= 0) return SQ_EINVAL;
    if (hidden & 7) return SQ_EUNSUPPORTED;
    if (rows == 0) return SQ_OK;
    hipLaunchKernelGGL((rmsnorm_kernel<false, false>), dim3(rows), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (const half_t*)nullptr, (half_t*)nullptr, (const half_t*)weight, (half_t*)out,
                       hidden, eps, 0);
    return sq_check_launch();
}

// same, output written fragment-major ([hidden / 32][ceil(rows / 16)][64][8]) for sq_linear_ts_f16
extern "C" int sq_rmsnorm_frag_f16(const void* x, const void* weight, void* out_frag, int rows, int hidden, float eps,
                                   void* stream) {
    if (!x || !weight || !out_frag || rows < 0 || hidden <= 0) return SQ_EINVAL;
    if (hidden & 31) return SQ_EUNSUPPORTED;
    if (rows == 0) return SQ_OK;
    hipLaunchKernelGGL((rmsnorm_kernel<false, true>), dim3(rows), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (const half_t*)nullptr, (half_t*)nullptr, (const half_t*)weight,
                       (half_t*)out_frag, hidden, eps, (rows + 15) / 16);
    return sq_check_launch();
}

// h = x + residual (written to sum_out, may alias residual), out = rmsnorm(h) * weight
extern "C" int sq_add_rmsnorm_f16(const void* x, const void* residual, void* sum_out, const void* weight, void* out,
                                   int rows, int hidden, float eps, void* stream) {
    if (!x || !residual || !sum_out || !weight || !out || rows < 0 || hidden <= 0) return SQ_EINVAL;
    if (hidden & 7) return SQ_EUNSUPPORTED;
    if (rows == 0) return SQ_OK;
    hipLaunchKernelGGL((rmsnorm_kernel<true, false>), dim3(rows), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (const half_t*)residual, (half_t*)sum_out, (const half_t*)weight,
                       (half_t*)out, hidden, eps, 0);
    return sq_check_launch();
}

extern "C" int sq_add_rmsnorm_frag_f16(const void* x, const void* residual, void* sum_out, const void* weight,
                                        void* out_frag, int rows, int hidden, float eps, void* stream) {
    if (!x || !residual || !sum_out || !weight || !out_frag || rows < 0 || hidden <= 0) return SQ_EINVAL;
    if (hidden & 31) return SQ_EUNSUPPORTED;
    if (rows == 0) return SQ_OK;
    hipLaunchKernelGGL((rmsnorm_kernel<true, true>), dim3(rows), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (const half_t*)residual, (half_t*)sum_out, (const half_t*)weight,
                       (half_t*)out_frag, hidden, eps, (rows + 15) / 16);
    return sq_check_launch();
}

// First norm of a forward fused with the embedding lookup (Engine/Llama_model.py:151 `embed_tokens(input_ids)` +
// the first decoder layer's input_layernorm): row r of the residual stream is embed[ids[r]]; it is written out (the
// residual stream) and normalised in the same pass, row-major or fragment-major.  Same arithmetic as rmsnorm_kernel.
// STAGE: the forward's inputs are staged here as well (what stage_tree_inputs_kernel does in a launch of its own, csrc/kv_ops.hip:
// the device-driven speculation step): row i is the token at slot gt + rel_slot0 + i of the tree's token buffer, gt from the
// device step block; its id / storage id / position id and the {q_slot0, gt, kv_len} context block are written for the
// kernels that follow (RoPE, attention), the step block advances when asked.  All of it is wave-uniform scalar work in
// front of the embedding row's load -- one launch per forward fewer (7 per speculation step).
struct StageArgs {
    int64_t* dst_ids; int64_t* dst_pos; int64_t* dst_sto; int32_t* ctx;
    const int64_t* tokens; const int32_t* depth; int32_t* d_step;
    int n_tree, rel_slot0, rel_kv_len, advance;
};
template <int THREADS, int CPT, bool STAGE>
__global__ void __launch_bounds__(THREADS)
embed_rmsnorm_kernel(const int64_t* __restrict__ ids, const half_t* __restrict__ embed, int vocab,
                     const half_t* __restrict__ w, half_t* __restrict__ x_out, half_t* __restrict__ out, int hidden,
                     float eps, int frag_mtp, const StageArgs sa) {
    __shared__ float s_f[THREADS / 64];
    const size_t row = blockIdx.x;
    int64_t id;
    if (STAGE) {
        const int gt = sa.advance ? sa.d_step[SQ_STEP_NEXT_GT] : sa.d_step[SQ_STEP_GT];
        const int q_slot0 = gt + sa.rel_slot0;
        const int slot = q_slot0 + (int)row;
        const int t = slot - (gt - 1);
        id = sa.tokens[slot];
        if (threadIdx.x == 0) {
            sa.dst_ids[row] = id;
            sa.dst_sto[row] = slot;
            sa.dst_pos[row] = (t >= 0 && t < sa.n_tree) ? (int64_t)sa.depth[t] + gt - 1 : (int64_t)slot;
        }
        if (row == 0) {
            if (threadIdx.x < 3) sa.ctx[threadIdx.x] = threadIdx.x == 0 ? q_slot0 : (threadIdx.x == 1 ? gt : gt + sa.rel_kv_len);
            if (sa.advance && threadIdx.x == 0) {           // (the other rows read NEXT_GT, nobody else touches INDEX)
                sa.d_step[SQ_STEP_GT] = gt;
                sa.d_step[SQ_STEP_INDEX] = sa.d_step[SQ_STEP_INDEX] + 1;
            }
        }
    } else {
        id = ids[row];
    }
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const half_t* src = embed + (size_t)id * hidden;
    const int chunks = hidden >> 3;
    half8 v[CPT], wv[CPT];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = threadIdx.x + i * THREADS;
        if (c < chunks) {
            v[i] = *(const half8*)(src + c * 8);
            wv[i] = *(const half8*)(w + c * 8);
            *(half8*)(x_out + row * hidden + c * 8) = v[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += (float)v[i][j] * (float)v[i][j];
        }
    }
    const float tot = block_sum_f32<THREADS / 64>(ss, s_f);
    const float inv = rsqrtf(tot / (float)hidden + eps);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = threadIdx.x + i * THREADS;
        if (c < chunks) {
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const half_t n = (half_t)((float)v[i][j] * inv);
                o[j] = (half_t)((float)wv[i][j] * (float)n);
            }
            *(half8*)(out + (frag_mtp ? frag_chunk_offset(row, c, frag_mtp) : row * hidden + c * 8)) = o;
        }
    }
}

template <bool STAGE>
static int embed_launch(const int64_t* d_ids, const void* embed, int vocab, const void* weight, void* x_out, void* out,
                        int out_frag, int rows, int hidden, float eps, const StageArgs& sa, void* stream) {
    if (!embed || !weight || !x_out || !out || rows < 0 || hidden <= 0 || vocab <= 0) return SQ_EINVAL;
    if ((hidden & 7) || (out_frag && (hidden & 31)) || hidden > 8 * 1024 * 4) return SQ_EUNSUPPORTED;
    if (rows == 0) return SQ_OK;
    const int chunks = hidden >> 3;
    const int mtp = out_frag ? (rows + 15) / 16 : 0;
    hipStream_t st = (hipStream_t)stream;
#define SQ_EMB(T_, C_)                                                                                             \
    hipLaunchKernelGGL((embed_rmsnorm_kernel<T_, C_, STAGE>), dim3(rows), dim3(T_), 0, st, d_ids, (const half_t*)embed,  \
                       vocab, (const half_t*)weight, (half_t*)x_out, (half_t*)out, hidden, eps, mtp, sa)
    if (chunks <= 256) SQ_EMB(256, 1);
    else if (chunks <= 512) SQ_EMB(512, 1);
    else if (chunks <= 1024) SQ_EMB(1024, 1);
    else if (chunks <= 2048) SQ_EMB(1024, 2);
    else SQ_EMB(1024, 4);
#undef SQ_EMB
    return sq_check_launch();
}

extern "C" int sq_embed_rmsnorm_f16(const int64_t* d_ids, const void* embed, int vocab, const void* weight, void* x_out,
                                    void* out, int out_frag, int rows, int hidden, float eps, void* stream) {
    if (!d_ids) return SQ_EINVAL;
    return embed_launch<false>(d_ids, embed, vocab, weight, x_out, out, out_frag, rows, hidden, eps, StageArgs{}, stream);
}

extern "C" int sq_embed_stage_rmsnorm_f16(int64_t* dst_ids, int64_t* dst_pos, int64_t* dst_storage, int32_t* d_ctx,
                                          const int64_t* tokens, const int32_t* d_depth, int n_tree, int rel_slot0,
                                          int rel_kv_len, int32_t* d_step, int advance, const void* embed, int vocab,
                                          const void* weight, void* x_out, void* out, int out_frag, int rows, int hidden,
                                          float eps, void* stream) {
    if (!dst_ids || !dst_pos || !dst_storage || !d_ctx || !tokens || !d_depth || !d_step) return SQ_EINVAL;
    if (rows <= 0 || n_tree <= 0 || n_tree > SQ_MAX_TREE) return SQ_EINVAL;
    StageArgs sa;
    sa.dst_ids = dst_ids; sa.dst_pos = dst_pos; sa.dst_sto = dst_storage; sa.ctx = d_ctx; sa.tokens = tokens;
    sa.depth = d_depth; sa.d_step = d_step; sa.n_tree = n_tree; sa.rel_slot0 = rel_slot0; sa.rel_kv_len = rel_kv_len;
    sa.advance = advance ? 1 : 0;
    return embed_launch<true>(nullptr, embed, vocab, weight, x_out, out, out_frag, rows, hidden, eps, sa, stream);
}

// Same as rmsnorm_kernel<true>, with x arriving as `splits` fp32 partial products of a split-K linear layer
// (sq_linear_ts_f16): x = h(((s0 + s1) + s2) + ...) -- the layer's fp16 output rounding -- then h = x + res.
// NORM = false stops after the add (the sum feeds a later, separate normalisation).
// One pass: a thread owns CPT 8-element chunks of the row (THREADS x CPT >= hidden / 8), every load of the row -- the
// slabs, the residual, the norm weight -- is issued before the first use, the summed row stays in registers across the
// block reduction, so the kernel is one memory latency + one reduction deep (64 launches per 7B verify).
template <int THREADS, int CPT, bool NORM>
__global__ void __launch_bounds__(THREADS)
rmsnorm_slabs_kernel(const float* __restrict__ slab, int splits, size_t split_stride, const half_t* res,
                     half_t* sum_out,   // res and sum_out may alias
                     const half_t* __restrict__ w, half_t* __restrict__ out, int hidden, float eps, int frag_mtp) {
    __shared__ float s_f[THREADS / 64];
    const size_t row = blockIdx.x;
    const int chunks = hidden >> 3;
    half8 v[CPT], wv[CPT];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = threadIdx.x + i * THREADS;
        if (c < chunks) {
            const half8 r = *(const half8*)(res + row * hidden + c * 8);
            if (NORM) wv[i] = *(const half8*)(w + c * 8);
            const float* const sp[1] = {slab + row * hidden + c * 8};
            floatx4 a[1], b[1];
            slab_sum8<1>(sp, splits, split_stride, a, b);      // all partials in flight together with r and wv
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[i][j] = (half_t)((float)(half_t)a[0][j] + (float)r[j]);
                v[i][4 + j] = (half_t)((float)(half_t)b[0][j] + (float)r[4 + j]);
            }
            *(half8*)(sum_out + row * hidden + c * 8) = v[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += (float)v[i][j] * (float)v[i][j];
        }
    }
    if (!NORM) return;
    const float tot = block_sum_f32<THREADS / 64>(ss, s_f);
    const float inv = rsqrtf(tot / (float)hidden + eps);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = threadIdx.x + i * THREADS;
        if (c < chunks) {
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const half_t n = (half_t)((float)v[i][j] * inv);
                o[j] = (half_t)((float)wv[i][j] * (float)n);
            }
            *(half8*)(out + (frag_mtp ? frag_chunk_offset(row, c, frag_mtp) : row * hidden + c * 8)) = o;
        }
    }
}

extern "C" int sq_add_rmsnorm_slabs_f16(const void* slab, int splits, const void* residual, const void* weight,
                                         void* sum_out, void* out, int out_frag, int rows, int hidden, float eps,
                                         void* stream) {
    if (!slab || !residual || !sum_out || splits < 1 || rows < 0 || hidden <= 0 || (out && !weight)) return SQ_EINVAL;
    if ((hidden & 7) || ((uintptr_t)slab & 15) || (out_frag && (hidden & 31)) || hidden > 8 * 1024 * 4) return SQ_EUNSUPPORTED;
    if (rows == 0) return SQ_OK;
    const size_t stride = (size_t)rows * hidden;
    const int chunks = hidden >> 3;
    const int mtp = out_frag ? (rows + 15) / 16 : 0;
    hipStream_t st = (hipStream_t)stream;
#define SQ_SLABS(T_, C_)                                                                                           \
    {                                                                                                              \
        if (out)                                                                                                   \
            hipLaunchKernelGGL((rmsnorm_slabs_kernel<T_, C_, true>), dim3(rows), dim3(T_), 0, st, (const float*)slab,  \
                               splits, stride, (const half_t*)residual, (half_t*)sum_out, (const half_t*)weight,       \
                               (half_t*)out, hidden, eps, mtp);                                                        \
        else                                                                                                       \
            hipLaunchKernelGGL((rmsnorm_slabs_kernel<T_, C_, false>), dim3(rows), dim3(T_), 0, st, (const float*)slab, \
                               splits, stride, (const half_t*)residual, (half_t*)sum_out, (const half_t*)nullptr,      \
                               (half_t*)nullptr, hidden, eps, 0);                                                      \
    }
    if (chunks <= 256) SQ_SLABS(256, 1)
    else if (chunks <= 512) SQ_SLABS(512, 1)
    else if (chunks <= 1024) SQ_SLABS(1024, 1)
    else if (chunks <= 2048) SQ_SLABS(1024, 2)
    else SQ_SLABS(1024, 4)
#undef SQ_SLABS
    return sq_check_launch();
}

// gate_up: [rows][2*inter] (gate | up), out: [rows][inter];  out = h(h(silu(gate)) * up)
__global__ void __launch_bounds__(ROW_THREADS)
silu_mul_kernel(const half_t* __restrict__ gate_up, half_t* __restrict__ out, int inter, int frag_mtp) {
    const size_t row = blockIdx.y;
    const int c = blockIdx.x * ROW_THREADS + threadIdx.x;
    if (c * 8 >= inter) return;
    if (inter & 7) {      // odd widths (a tensor-parallel shard of a toy model): element-wise, same arithmetic
        const half_t* g = gate_up + row * 2 * inter;
        for (int j = c * 8; j < min(inter, c * 8 + 8); ++j) {
            const float gf = (float)g[j];
            const half_t sv = (half_t)(gf / (1.0f + expf(-gf)));
            out[row * inter + j] = (half_t)((float)sv * (float)g[inter + j]);
        }
        return;
    }
    const half8 gv = *(const half8*)(gate_up + row * 2 * inter + c * 8);
    const half8 uv = *(const half8*)(gate_up + row * 2 * inter + inter + c * 8);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gf = (float)gv[j];
        const half_t s = (half_t)(gf / (1.0f + expf(-gf)));
        o[j] = (half_t)((float)s * (float)uv[j]);
    }
    *(half8*)(out + (frag_mtp ? frag_chunk_offset(row, c, frag_mtp) : row * inter + c * 8)) = o;
}

extern "C" int sq_silu_mul_f16(const void* gate_up, void* out, int rows, int inter, void* stream) {
    if (!gate_up || !out || rows < 0 || inter <= 0) return SQ_EINVAL;
    if (rows == 0) return SQ_OK;
    const int chunks = (inter + 7) >> 3;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((chunks + ROW_THREADS - 1) / ROW_THREADS, rows), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, (const half_t*)gate_up, (half_t*)out, inter, 0);
    return sq_check_launch();
}

extern "C" int sq_silu_mul_frag_f16(const void* gate_up, void* out_frag, int rows, int inter, void* stream) {
    if (!gate_up || !out_frag || rows < 0 || inter <= 0) return SQ_EINVAL;
    if (inter & 31) return SQ_EUNSUPPORTED;
    if (rows == 0) return SQ_OK;
    const int chunks = inter >> 3;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((chunks + ROW_THREADS - 1) / ROW_THREADS, rows), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, (const half_t*)gate_up, (half_t*)out_frag, inter, (rows + 15) / 16);
    return sq_check_launch();
}

// SwiGLU fed by a split-K gate|up projection: the projection ran as a plain [2 inter] x k layer with `splits` K-splits
// (sq_linear_ts_f16, silu = 0) and left fp32 partials slab[s][rows][2 inter] (gate columns, then up columns); here
// g = h(sum_s gate), u = h(sum_s up) -- the layer's fp16 output roundings, splits summed in order -- and
// out = h(h(silu(g)) * u) (Engine/Llama_modules.py:271), written fragment-major (frag_mtp > 0) or row-major.
__global__ void __launch_bounds__(ROW_THREADS)
silu_mul_slabs_kernel(const float* __restrict__ slab, int splits, size_t split_stride, half_t* __restrict__ out, int inter,
                      int frag_mtp) {
    const size_t row = blockIdx.y;
    const int c = blockIdx.x * ROW_THREADS + threadIdx.x;
    if (c * 8 >= inter) return;
    const float* gp = slab + row * 2 * inter + c * 8;
    const float* up = gp + inter;
    const float* const sp[2] = {gp, up};
    floatx4 lo[2], hi[2];
    slab_sum8<2>(sp, splits, split_stride, lo, hi);
    const floatx4 g0 = lo[0], g1 = hi[0], u0 = lo[1], u1 = hi[1];
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gf = (float)(half_t)(j < 4 ? g0[j] : g1[j - 4]);
        const half_t sg = (half_t)(gf / (1.0f + expf(-gf)));
        o[j] = (half_t)((float)sg * (float)(half_t)(j < 4 ? u0[j] : u1[j - 4]));
    }
    *(half8*)(out + (frag_mtp ? frag_chunk_offset(row, c, frag_mtp) : row * inter + c * 8)) = o;
}

extern "C" int sq_silu_mul_slabs_f16(const void* slab, int splits, void* out, int out_frag, int rows, int inter, void* stream) {
    if (!slab || !out || splits < 1 || rows < 0 || inter <= 0) return SQ_EINVAL;
    if ((inter & 7) || (out_frag && (inter & 31)) || ((uintptr_t)slab & 15)) return SQ_EUNSUPPORTED;
    if (rows == 0) return SQ_OK;
    const int chunks = inter >> 3;
    hipLaunchKernelGGL(silu_mul_slabs_kernel, dim3((chunks + ROW_THREADS - 1) / ROW_THREADS, rows), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, (const float*)slab, splits, (size_t)rows * 2 * inter, (half_t*)out, inter,
                       out_frag ? (rows + 15) / 16 : 0);
    return sq_check_launch();
}
