// KV-slot movement, dense tree mask, RoPE + KV write.  All kernels here are HBM-bound byte
// movers: 16-byte accesses per lane, one (layer, head) tile per workgroup.
#include "common.h"
#include <string.h>

// ---- library-level helpers -------------------------------------------------------------------
static thread_local char g_err[256] = "";
void sq_set_error(hipError_t e) {
    const char* s = hipGetErrorString(e);
    strncpy(g_err, s ? s : "unknown", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" int sq_version(void) { return 100; }  // 0.1.0
extern "C" const char* sq_last_error(void) { return g_err; }
extern "C" int sq_device_ready(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

__global__ void store_i32_kernel(int32_t* dst, int n, int v0, int v1, int v2, int v3) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = i == 0 ? v0 : (i == 1 ? v1 : (i == 2 ? v2 : v3));
}
extern "C" int sq_store_i32(int32_t* dst, int n, int v0, int v1, int v2, int v3, void* stream) {
    if (!dst || n < 0 || n > 4) return SQ_EINVAL;
    if (n == 0) return SQ_OK;
    hipLaunchKernelGGL(store_i32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, n, v0, v1, v2, v3);
    return sq_check_launch();
}

// Stages the inputs of a captured forward in one launch: input_ids / position_ids / storage_ids of the q_len new tokens are
// copied from the tree's buffers into the graph's static buffers, and the {q_slot0, gt, kv_len} context block is written
// (what the reference's capture_graph closure does with four copy_ calls, Engine/Engine.py:156-163).
__global__ void stage_inputs_kernel(int64_t* __restrict__ dst_ids, const int64_t* __restrict__ src_ids,
                                    int64_t* __restrict__ dst_pos, const int64_t* __restrict__ src_pos,
                                    int64_t* __restrict__ dst_sto, const int64_t* __restrict__ src_sto, int q_len,
                                    int32_t* __restrict__ ctx, int c0, int c1, int c2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q_len) {
        dst_ids[i] = src_ids[i];
        dst_pos[i] = src_pos[i];
        dst_sto[i] = src_sto[i];
    }
    if (ctx && i < 3) ctx[i] = i == 0 ? c0 : (i == 1 ? c1 : c2);
}
extern "C" int sq_stage_inputs(int64_t* dst_ids, const int64_t* src_ids, int64_t* dst_pos, const int64_t* src_pos,
                               int64_t* dst_storage, const int64_t* src_storage, int q_len, int32_t* d_ctx, int q_slot0,
                               int gt, int kv_len, void* stream) {
    if (!dst_ids || !src_ids || !dst_pos || !src_pos || !dst_storage || !src_storage || q_len <= 0) return SQ_EINVAL;
    hipLaunchKernelGGL(stage_inputs_kernel, dim3((q_len + 127) / 128), dim3(128), 0, (hipStream_t)stream, dst_ids, src_ids,
                       dst_pos, src_pos, dst_storage, src_storage, q_len, d_ctx, q_slot0, gt, kv_len);
    return sq_check_launch();
}

// Device-driven staging of a tree forward (speculation step without host-side scalars): the q_len queries sit at slots
// [gt + rel_slot0, gt + rel_slot0 + q_len) with gt read from the device step block; input ids come from the tree's token
// buffer, storage ids are the slots, position ids follow Tree/SpecTree.py:61,264-270 (committed text: the slot itself;
// tree node t at slot gt-1+t: depth[t] + gt - 1), the forward's context block becomes {q_slot0, gt, gt + rel_kv_len}.
// advance != 0 first moves the step block to the next step (gt <- next_gt, index += 1): used by the forward that follows
// the verification (the 1-token draft forward of prepare_for_next_iter, Tree/SpecTree.py:261-281).
__global__ void stage_tree_inputs_kernel(int64_t* __restrict__ dst_ids, int64_t* __restrict__ dst_pos,
                                         int64_t* __restrict__ dst_sto, int32_t* __restrict__ ctx,
                                         const int64_t* __restrict__ tokens, const int32_t* __restrict__ depth, int n_tree,
                                         int q_len, int rel_slot0, int rel_kv_len, int32_t* d_step, int advance) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int gt = advance ? d_step[SQ_STEP_NEXT_GT] : d_step[SQ_STEP_GT];
    const int q_slot0 = gt + rel_slot0;
    if (i < q_len) {
        const int slot = q_slot0 + i;
        const int t = slot - (gt - 1);
        dst_ids[i] = tokens[slot];
        dst_sto[i] = slot;
        dst_pos[i] = (t >= 0 && t < n_tree) ? (int64_t)depth[t] + gt - 1 : (int64_t)slot;
    }
    if (i < 3) ctx[i] = i == 0 ? q_slot0 : (i == 1 ? gt : gt + rel_kv_len);
    if (advance && i == 0) {
        d_step[SQ_STEP_GT] = gt;
        d_step[SQ_STEP_INDEX] = d_step[SQ_STEP_INDEX] + 1;
    }
}
extern "C" int sq_stage_tree_inputs(int64_t* dst_ids, int64_t* dst_pos, int64_t* dst_storage, int32_t* d_ctx,
                                    const int64_t* tokens, const int32_t* d_depth, int n_tree, int q_len, int rel_slot0,
                                    int rel_kv_len, int32_t* d_step, int advance, void* stream) {
    if (!dst_ids || !dst_pos || !dst_storage || !d_ctx || !tokens || !d_depth || !d_step) return SQ_EINVAL;
    if (q_len <= 0 || n_tree <= 0 || n_tree > SQ_MAX_TREE) return SQ_EINVAL;
    hipLaunchKernelGGL(stage_tree_inputs_kernel, dim3((q_len + 127) / 128), dim3(128), 0, (hipStream_t)stream, dst_ids, dst_pos,
                       dst_storage, d_ctx, tokens, d_depth, n_tree, q_len, rel_slot0, rel_kv_len, d_step, advance);
    return sq_check_launch();
}

// ---- a1: bitmask from children CSR (host) ----------------------------------------------------
extern "C" int sq_tree_bitmask_from_successors(const int32_t* child_off, const int32_t* child_ids,
                                               int n, uint64_t* out, int words) {
    if (!child_off || !out || n <= 0 || n > SQ_MAX_TREE || words < SQ_MASK_WORDS(n)) return SQ_EINVAL;
    if (child_off[n] > 0 && !child_ids) return SQ_EINVAL;
    memset(out, 0, (size_t)n * words * sizeof(uint64_t));
    // BFS order (parents precede children, tree_search.py output): a child's row is its
    // parent's row plus its own bit.
    out[0] = 1ull;
    for (int p = 0; p < n; ++p) {
        for (int e = child_off[p]; e < child_off[p + 1]; ++e) {
            const int c = child_ids[e];
            if (c <= p || c >= n) return SQ_EINVAL;
            for (int w = 0; w < words; ++w) out[(size_t)c * words + w] = out[(size_t)p * words + w];
            out[(size_t)c * words + (c >> 6)] |= 1ull << (c & 63);
        }
    }
    return SQ_OK;
}

// ---- a1: dense additive mask rows ------------------------------------------------------------
__global__ void tree_mask_dense_kernel(half_t* out, int out_stride, int n_cols, int q_slot0, int q_len,
                                       int gt, int n_tree, const uint64_t* bitmask, int words) {
    const int row = blockIdx.x;
    const int slot = q_slot0 + row;
    const int t = slot - (gt - 1);
    const bool tree_row = slot >= gt;
    const bool dead_row = tree_row && t >= n_tree;
    const uint64_t* bm = (tree_row && !dead_row) ? bitmask + (size_t)t * words : nullptr;
    const half_t neg = (half_t)(-65504.0f);
    for (int c = threadIdx.x; c < n_cols; c += blockDim.x) {
        const bool vis = !dead_row && tree_visible(slot, c, gt, n_tree, bm);
        out[(size_t)row * out_stride + c] = vis ? (half_t)0.0f : neg;
    }
}

extern "C" int sq_tree_mask_dense_f16(void* out, int out_stride, int n_cols, int q_slot0, int q_len, int gt,
                                      int n_tree, const uint64_t* d_bitmask, int words, void* stream) {
    if (!out || q_len < 0 || n_cols < 0 || out_stride < n_cols || gt < 1 || n_tree < 1) return SQ_EINVAL;
    if (n_tree > 1 && (!d_bitmask || words < SQ_MASK_WORDS(n_tree))) return SQ_EINVAL;
    if (q_len == 0 || n_cols == 0) return SQ_OK;
    hipLaunchKernelGGL(tree_mask_dense_kernel, dim3(q_len), dim3(256), 0, (hipStream_t)stream, (half_t*)out,
                       out_stride, n_cols, q_slot0, q_len, gt, n_tree, d_bitmask, words);
    return sq_check_launch();
}

// ---- a5: scatter new K/V rows into their slots -----------------------------------------------
// grid (q_len, H_kv); one 16-byte chunk per lane.
__global__ void kv_scatter_kernel(half_t* k_layer, half_t* v_layer, const half_t* new_k, const half_t* new_v,
                                  const int64_t* storage_ids, int q_len, int m, int d) {
    const int i = blockIdx.x, hh = blockIdx.y;
    const int64_t slot = storage_ids[i];
    if (slot < 0 || slot >= m) return;
    const int chunks = d >> 3;
    const u32x4* sk = (const u32x4*)(new_k + ((size_t)hh * q_len + i) * d);
    const u32x4* sv = (const u32x4*)(new_v + ((size_t)hh * q_len + i) * d);
    u32x4* dk = (u32x4*)(k_layer + ((size_t)hh * m + slot) * d);
    u32x4* dv = (u32x4*)(v_layer + ((size_t)hh * m + slot) * d);
    for (int c = threadIdx.x; c < 2 * chunks; c += blockDim.x) {
        if (c < chunks) dk[c] = sk[c];
        else dv[c - chunks] = sv[c - chunks];
    }
}

extern "C" int sq_kv_scatter_f16(void* k_layer, void* v_layer, const void* new_k, const void* new_v,
                                 const int64_t* d_storage_ids, int q_len, int h_kv, int m, int d, void* stream) {
    if (!k_layer || !v_layer || !new_k || !new_v || !d_storage_ids || q_len < 0 || h_kv <= 0 || m <= 0) return SQ_EINVAL;
    if (d <= 0 || (d & 7)) return SQ_EUNSUPPORTED;
    if (q_len == 0) return SQ_OK;
    hipLaunchKernelGGL(kv_scatter_kernel, dim3(q_len, h_kv), dim3(64), 0, (hipStream_t)stream, (half_t*)k_layer,
                       (half_t*)v_layer, (const half_t*)new_k, (const half_t*)new_v, d_storage_ids, q_len, m, d);
    return sq_check_launch();
}

// ---- a5: accepted-path compaction --------------------------------------------------------------
// grid (L*H_kv, 2): blockIdx.y selects K or V.  256 threads; a row of D halves is D/8 lanes.
// Rows are moved in passes of ROWS_PER_PASS: all loads of a pass complete (registers) before any
// store of that pass, and passes run in ascending order, which is safe because the slots are
// ascending and dst_j <= slot_j (BFS slot order), so a store can only overwrite a source that was
// already consumed (SURVEY.md §7 "KV compaction aliasing").
__device__ __forceinline__ void kv_compact_tile(half_t* tile, int m, int d, const int32_t* slots, const int32_t* d_count,
                                                int max_count, int dst_offset, int zero_end, const int32_t* d_dst_offset) {
    if (d_dst_offset) dst_offset = *d_dst_offset;          // device-driven step: the ground-truth length lives on the device
    int count = d_count ? *d_count : max_count;
    if (count > max_count) count = max_count;
    if (count < 0) count = 0;
    const int lanes_per_row = d >> 3;
    const int rows_per_pass = 256 / lanes_per_row;
    const int rr = threadIdx.x / lanes_per_row, cc = threadIdx.x % lanes_per_row;
    for (int base = 0; base < count; base += rows_per_pass) {
        const int j = base + rr;
        u32x4 val;
        bool act = (rr < rows_per_pass) && (j < count);
        int src = 0;
        if (act) {
            src = slots[j];
            act = src >= 0 && src < m && (dst_offset + j) < m;
        }
        if (act) val = *(const u32x4*)(tile + (size_t)src * d + cc * 8);
        __syncthreads();
        if (act && src != dst_offset + j) *(u32x4*)(tile + (size_t)(dst_offset + j) * d + cc * 8) = val;
        __syncthreads();
    }
    const int z0 = dst_offset + count;
    if (zero_end > z0) {
        const int n_chunks = (zero_end - z0) * lanes_per_row;
        u32x4 zero = {0, 0, 0, 0};
        u32x4* p = (u32x4*)(tile + (size_t)z0 * d);
        for (int c = threadIdx.x; c < n_chunks; c += 256) p[c] = zero;
    }
}

__global__ void __launch_bounds__(256) kv_compact_kernel(half_t* k_cache, half_t* v_cache, int m, int d,
                                                         const int32_t* slots, const int32_t* d_count, int max_count,
                                                         int dst_offset, int zero_end, const int32_t* d_dst_offset) {
    half_t* tile = (blockIdx.y == 0 ? k_cache : v_cache) + (size_t)blockIdx.x * m * d;
    kv_compact_tile(tile, m, d, slots, d_count, max_count, dst_offset, zero_end, d_dst_offset);
}

// Both caches of a speculation step (draft + target: the same accepted slots, the same destination) in ONE launch: the
// device-driven step rolls the draft cache and the target cache back to the accepted path back to back
// (Tree/SpecTree.py:226-227); as two dependent graph nodes the second costs a kernel boundary for ~1 us of work.
struct KvPair { half_t *k0, *v0, *k1, *v1; int tiles0, m0, d0, tiles1, m1, d1; };
__global__ void __launch_bounds__(256) kv_compact2_kernel(const KvPair P, const int32_t* slots, const int32_t* d_count, int max_count,
                                                          int dst_offset, const int32_t* d_dst_offset) {
    const int b = blockIdx.x;
    const bool second = b >= P.tiles0;
    const int t = second ? b - P.tiles0 : b, m = second ? P.m1 : P.m0, d = second ? P.d1 : P.d0;
    half_t* base = blockIdx.y == 0 ? (second ? P.k1 : P.k0) : (second ? P.v1 : P.v0);
    kv_compact_tile(base + (size_t)t * m * d, m, d, slots, d_count, max_count, dst_offset, 0, d_dst_offset);
}

extern "C" int sq_kv_compact2_f16(void* k0, void* v0, int n_layers0, int h_kv0, int m0, int d0, void* k1, void* v1, int n_layers1,
                                  int h_kv1, int m1, int d1, const int32_t* d_slots, const int32_t* d_count, int max_count,
                                  int dst_offset, const int32_t* d_dst_offset, void* stream) {
    if (!k0 || !v0 || !k1 || !v1 || n_layers0 <= 0 || h_kv0 <= 0 || m0 <= 0 || n_layers1 <= 0 || h_kv1 <= 0 || m1 <= 0 ||
        max_count < 0 || dst_offset < 0) return SQ_EINVAL;
    if (max_count > 0 && !d_slots) return SQ_EINVAL;
    for (int d : {d0, d1})
        if (d <= 0 || (d & 7) || d > 2048 || (256 % (d >> 3)) != 0) return SQ_EUNSUPPORTED;
    if (max_count == 0 && !d_dst_offset) return SQ_OK;
    KvPair P{(half_t*)k0, (half_t*)v0, (half_t*)k1, (half_t*)v1, n_layers0 * h_kv0, m0, d0, n_layers1 * h_kv1, m1, d1};
    hipLaunchKernelGGL(kv_compact2_kernel, dim3(P.tiles0 + P.tiles1, 2), dim3(256), 0, (hipStream_t)stream, P, d_slots, d_count,
                       max_count, dst_offset, d_dst_offset);
    return sq_check_launch();
}

extern "C" int sq_kv_compact_f16(void* k_cache, void* v_cache, int n_layers, int h_kv, int m, int d,
                                 const int32_t* d_slots, const int32_t* d_count, int max_count, int dst_offset,
                                 int zero_end, const int32_t* d_dst_offset, void* stream) {
    if (!k_cache || !v_cache || n_layers <= 0 || h_kv <= 0 || m <= 0 || max_count < 0 || dst_offset < 0) return SQ_EINVAL;
    if (max_count > 0 && !d_slots) return SQ_EINVAL;
    if (d <= 0 || (d & 7) || d > 2048 || (256 % (d >> 3)) != 0) return SQ_EUNSUPPORTED;
    if (zero_end > m) return SQ_EINVAL;
    if (max_count == 0 && zero_end <= dst_offset && !d_dst_offset) return SQ_OK;
    if (d_dst_offset && zero_end > 0) return SQ_EUNSUPPORTED;      // tail zeroing needs the host's view of the lengths
    hipLaunchKernelGGL(kv_compact_kernel, dim3(n_layers * h_kv, 2), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)k_cache, (half_t*)v_cache, m, d, d_slots, d_count, max_count, dst_offset, zero_end,
                       d_dst_offset);
    return sq_check_launch();
}

__global__ void __launch_bounds__(256) kv_clear_kernel(half_t* k_cache, half_t* v_cache, int m, int d, int used_rows) {
    half_t* tile = (blockIdx.y == 0 ? k_cache : v_cache) + (size_t)blockIdx.x * m * d;
    const int n_chunks = used_rows * (d >> 3);
    u32x4 zero = {0, 0, 0, 0};
    u32x4* p = (u32x4*)tile;
    for (int c = threadIdx.x; c < n_chunks; c += 256) p[c] = zero;
}

extern "C" int sq_kv_clear_f16(void* k_cache, void* v_cache, int n_layers, int h_kv, int m, int d, int used_rows,
                               void* stream) {
    if (!k_cache || !v_cache || n_layers <= 0 || h_kv <= 0 || m <= 0 || used_rows < 0 || used_rows > m) return SQ_EINVAL;
    if (d <= 0 || (d & 7)) return SQ_EUNSUPPORTED;
    if (used_rows == 0) return SQ_OK;
    hipLaunchKernelGGL(kv_clear_kernel, dim3(n_layers * h_kv, 2), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)k_cache, (half_t*)v_cache, m, d, used_rows);
    return sq_check_launch();
}

// ---- a3/a4: RoPE + KV write ------------------------------------------------------------------
// grid (q_len, ceil((H + 2*H_kv) * D/16 / 256)); a thread owns one (head, chunk) of a token row:
// chunk c covers the 8 halves [8c, 8c+8) of the first half and the matching 8 of the second half
// (rotate_half pairs element e with e + D/2).  fp16 rounding after every op, like the reference's
// fp16 tensor expression (q * cos) + (rotate_half(q) * sin) (Engine/offload_engine.py:63-66).
// SLAB: the packed q | k | v rows arrive as the split-K partials of the tall-skinny projection (fp32 [splits][q_len][stride],
// csrc/ts_linear.hip): a value is the sum of its partials in split order, rounded to fp16 -- what the projection itself
// would have written -- so the projection needs no pass of its own over its output.
// the two 8-element groups a thread owns (first half / second half of a head row), from fp16 rows or from the partials
template <bool SLAB>
__device__ __forceinline__ void rope_src8x2(const half_t* qkv, const float* slab, int splits, size_t split_stride, size_t off1,
                                            size_t off2, half8& x1, half8& x2) {
    if (!SLAB) { x1 = *(const half8*)(qkv + off1); x2 = *(const half8*)(qkv + off2); return; }
    const float* const sp[2] = {slab + off1, slab + off2};
    floatx4 lo[2], hi[2];
    slab_sum8<2>(sp, splits, split_stride, lo, hi);          // every partial of both groups in flight at once
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x1[j] = (half_t)lo[0][j]; x1[4 + j] = (half_t)hi[0][j];
        x2[j] = (half_t)lo[1][j]; x2[4 + j] = (half_t)hi[1][j];
    }
}

template <bool SLAB>
__global__ void rope_kv_write_kernel(const half_t* qkv, const float* slab, int splits, int qkv_stride, half_t* q_out,
                                     half_t* k_layer, half_t* v_layer, const half_t* cos_tab, const half_t* sin_tab,
                                     const int64_t* position_ids, const int64_t* storage_ids, int q_len, int n_heads,
                                     int h_kv, int d, int m) {
    const size_t split_stride = (size_t)q_len * qkv_stride;
    const int i = blockIdx.x;
    const int half_d = d >> 1;
    const int chunks = half_d >> 3;            // 16-byte chunks in the first half of a head row
    const int idx = blockIdx.y * blockDim.x + threadIdx.x;
    const int hj = idx / chunks;               // head index in the packed q | k | v row
    const int c = idx - hj * chunks;           // chunk of 8 inside the first half
    if (hj >= n_heads + 2 * h_kv) return;
    const size_t src = (size_t)i * qkv_stride + (size_t)hj * d;
    const int64_t slot = storage_ids[i];
    if (hj >= n_heads + h_kv) {                // V: plain copy into the slot
        if (slot < 0 || slot >= m) return;
        half_t* dst = v_layer + ((size_t)(hj - n_heads - h_kv) * m + slot) * d;
        half8 v1, v2;
        rope_src8x2<SLAB>(qkv, slab, splits, split_stride, src + c * 8, src + half_d + c * 8, v1, v2);
        *(half8*)(dst + c * 8) = v1;
        *(half8*)(dst + half_d + c * 8) = v2;
        return;
    }
    const int64_t pos = position_ids[i];
    const half8 c1 = *(const half8*)(cos_tab + (size_t)pos * d + c * 8);
    const half8 c2 = *(const half8*)(cos_tab + (size_t)pos * d + half_d + c * 8);
    const half8 s1 = *(const half8*)(sin_tab + (size_t)pos * d + c * 8);
    const half8 s2 = *(const half8*)(sin_tab + (size_t)pos * d + half_d + c * 8);
    half8 x1, x2;
    rope_src8x2<SLAB>(qkv, slab, splits, split_stride, src + c * 8, src + half_d + c * 8, x1, x2);
    half8 o1, o2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // first half: x1*cos + (-x2)*sin ; second half: x2*cos + x1*sin
        const half_t a1 = (half_t)((float)x1[e] * (float)c1[e]);
        const half_t b1 = (half_t)((float)(-x2[e]) * (float)s1[e]);
        o1[e] = (half_t)((float)a1 + (float)b1);
        const half_t a2 = (half_t)((float)x2[e] * (float)c2[e]);
        const half_t b2 = (half_t)((float)x1[e] * (float)s2[e]);
        o2[e] = (half_t)((float)a2 + (float)b2);
    }
    half_t* dst;
    if (hj < n_heads) {
        dst = q_out + ((size_t)hj * q_len + i) * d;
    } else {
        if (slot < 0 || slot >= m) return;
        dst = k_layer + ((size_t)(hj - n_heads) * m + slot) * d;
    }
    *(half8*)(dst + c * 8) = o1;
    *(half8*)(dst + half_d + c * 8) = o2;
}

static int rope_launch(const void* qkv, const float* slab, int splits, int qkv_stride, void* q_out, void* k_layer,
                       void* v_layer, const void* cos_tab, const void* sin_tab, const int64_t* d_position_ids,
                       const int64_t* d_storage_ids, int q_len, int n_heads, int h_kv, int d, int m, void* stream) {
    if (!q_out || !k_layer || !v_layer || !cos_tab || !sin_tab || !d_position_ids || !d_storage_ids) return SQ_EINVAL;
    if (q_len < 0 || n_heads <= 0 || h_kv <= 0 || m <= 0 || qkv_stride < (n_heads + 2 * h_kv) * d) return SQ_EINVAL;
    if (d <= 0 || (d & 15) || d > 1024 || (qkv_stride & 7)) return SQ_EUNSUPPORTED;
    if (q_len == 0) return SQ_OK;
    const int work = (n_heads + 2 * h_kv) * (d >> 4);
    const dim3 g(q_len, (work + 255) / 256), b(256);
    if (slab)
        hipLaunchKernelGGL(rope_kv_write_kernel<true>, g, b, 0, (hipStream_t)stream, (const half_t*)nullptr, slab, splits,
                           qkv_stride, (half_t*)q_out, (half_t*)k_layer, (half_t*)v_layer, (const half_t*)cos_tab,
                           (const half_t*)sin_tab, d_position_ids, d_storage_ids, q_len, n_heads, h_kv, d, m);
    else
        hipLaunchKernelGGL(rope_kv_write_kernel<false>, g, b, 0, (hipStream_t)stream, (const half_t*)qkv, (const float*)nullptr,
                           0, qkv_stride, (half_t*)q_out, (half_t*)k_layer, (half_t*)v_layer, (const half_t*)cos_tab,
                           (const half_t*)sin_tab, d_position_ids, d_storage_ids, q_len, n_heads, h_kv, d, m);
    return sq_check_launch();
}

extern "C" int sq_rope_kv_write_f16(const void* qkv, int qkv_stride, void* q_out, void* k_layer, void* v_layer,
                                    const void* cos_tab, const void* sin_tab, const int64_t* d_position_ids,
                                    const int64_t* d_storage_ids, int q_len, int n_heads, int h_kv, int d, int m,
                                    void* stream) {
    if (!qkv) return SQ_EINVAL;
    return rope_launch(qkv, nullptr, 0, qkv_stride, q_out, k_layer, v_layer, cos_tab, sin_tab, d_position_ids, d_storage_ids,
                       q_len, n_heads, h_kv, d, m, stream);
}

extern "C" int sq_rope_kv_write_slabs_f16(const float* qkv_slab, int splits, int qkv_stride, void* q_out, void* k_layer,
                                          void* v_layer, const void* cos_tab, const void* sin_tab,
                                          const int64_t* d_position_ids, const int64_t* d_storage_ids, int q_len, int n_heads,
                                          int h_kv, int d, int m, void* stream) {
    if (!qkv_slab || splits < 1 || ((uintptr_t)qkv_slab & 15)) return SQ_EINVAL;
    return rope_launch(nullptr, qkv_slab, splits, qkv_stride, q_out, k_layer, v_layer, cos_tab, sin_tab, d_position_ids,
                       d_storage_ids, q_len, n_heads, h_kv, d, m, stream);
}
