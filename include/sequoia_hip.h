/*
 * sequoia_hip.h — C ABI of libsequoia_hip.so, the MI355X (gfx950) native hot path of
 * Sequoia tree speculative decoding.
 *
 * The reference (Infini-AI-Lab/Sequoia) has no FFI layer: its hot path is stock PyTorch ops
 * called from Python.  Each entry point below replaces one op sequence of the reference; the
 * comment on every function cites the reference lines it stands in for (paths relative to the
 * reference checkout).  The reference-side binding is a ctypes stub.
 *
 * Conventions
 *   - plain C: raw device pointers + sizes, no torch / C++ types.
 *   - fp16 tensors are passed as `const void*` / `void*` to IEEE binary16 data.
 *   - token ids are int64 (the reference keeps `tokens` as torch.long); index arrays int32.
 *   - every device entry point is stream-ordered, allocation-free and sync-free
 *     (`stream` is a hipStream_t passed as void*; NULL = default stream), so that all of
 *     them can be captured into a hipGraph.
 *   - return value: SQ_OK (0) or a negative SQ_E* code; nothing throws.
 *   - KV cache layout is the reference's: [L][1][H_kv][M][D] fp16 (Engine/Llama_KV.py:16-34);
 *     a "layer" pointer addresses one [H_kv][M][D] slab.
 */
#ifndef SEQUOIA_HIP_H
#define SEQUOIA_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQ_OK            0
#define SQ_EINVAL       -1   /* bad argument (null pointer, negative size, ...)              */
#define SQ_EUNSUPPORTED -2   /* shape outside what the kernels are built for               */
#define SQ_ELAUNCH      -3   /* HIP reported an error at launch; see sq_last_error()        */

#define SQ_MAX_TREE      512 /* max tree nodes (ancestor bitmask = 8 x u64 words)           */
#define SQ_MASK_WORDS(n) (((n) + 63) / 64)
#define SQ_MAX_TOPK      128 /* max children per parent / samples per row                   */
#define SQ_RESULT_INTS   64  /* header of the int32 step-result record, see below; the buffer
                                passed as d_result holds SQ_RESULT_INTS + n_tree ints            */

/* Device-resident state of the device-driven speculation step (int32[SQ_STEP_INTS], owned by the caller): with it no
 * launch argument of a speculation step depends on the step, so construct_grow_map() + verify() + the next step's
 * preparation (Tree/SpecTree.py:245-281) replay as ONE hipGraph and the host reads the result record one step late. */
#define SQ_STEP_GT        0  /* ground_truth_len the current step runs with                                     */
#define SQ_STEP_NEXT_GT   1  /* written by sq_verify_*: gt of the next step (a + 1); a when terminal -- a step in
                                flight behind a terminal one then works beyond the finished text tokens[0, a)     */
#define SQ_STEP_INDEX     2  /* step counter: indexes the bonus-uniform table and the result ring               */
#define SQ_STEP_ACTIVE    3  /* cleared by sq_verify_* on a terminal step; while 0, sq_verify_* commits nothing
                                (no token write, no accepted slots: the KV compactions move 0 rows) and reports
                                a terminal record with reason SQ_REASON_SKIPPED                                  */
#define SQ_STEP_INTS      8
#define SQ_RESULT_RING    4  /* d_result_ring holds SQ_RESULT_RING records of SQ_RESULT_INTS ints, slot = index % ring */

/* Step-result record written by sq_verify_* (device memory, int32[SQ_RESULT_INTS]).        */
#define SQ_RES_ACCEPT_LEN 0  /* a = len(accept_list) = gt + #accepted tree nodes            */
#define SQ_RES_N_TREE     1  /* #accepted tree nodes                                        */
#define SQ_RES_BONUS      2  /* bonus token id written to tokens[a], -1 when terminal       */
#define SQ_RES_TERMINAL   3  /* 0 / 1                                                       */
#define SQ_RES_REASON     4  /* 0 none, 1 EOS token accepted, 2 NaN residual, 3 no slot left for the bonus token
                                (a >= token_capacity: the reference raises IndexError, Tree/SpecTree.py:222),
                                4 skipped: the step ran behind a terminal one (device-driven loop)            */
#define SQ_REASON_SKIPPED 4
#define SQ_RES_GT         5  /* echo of the ground_truth_len the step ran with              */
#define SQ_RES_LAST_NODE  6  /* tree-local id of the node the walk stopped at               */
#define SQ_RES_SLOTS      8  /* [8, 8+min(N_TREE,56)): absolute slots of the accepted nodes  */
/* [SQ_RESULT_INTS, SQ_RESULT_INTS + N_TREE): the complete list of accepted slots (ascending).  */

/* ---- library ---------------------------------------------------------------------------- */
int         sq_version(void);                 /* 10000*major + 100*minor + patch            */
const char* sq_last_error(void);              /* last HIP error string seen by this thread  */
int         sq_device_ready(void);            /* 1 if a gfx950 device is visible, else 0    */

/* ---- a1: tree-causal mask from the growmap ----------------------------------------------
 * Host helper.  Builds the ancestor-or-self bitmask of every tree node from the children
 * CSR (`Successors`, tree_search.py:121-128).  bit j of row i is set iff j is an ancestor of i
 * or j == i  ==  growmap["mask"][i][j] (Tree/SpecTree.py:45-48).
 * out: uint64[n][words], words >= SQ_MASK_WORDS(n).                                         */
int sq_tree_bitmask_from_successors(const int32_t* child_off, const int32_t* child_ids,
                                    int n, uint64_t* out, int words);

/* Device.  Materialises rows of the reference's dense additive mask
 * (Tree/Tree.py:20-27, Tree/SpecTree.py:54-58): out[i][c] = 0 where query slot q_slot0+i may
 * attend key slot c, else -65504.  Rule: rows < gt are causal (c <= slot); a tree row
 * (slot >= gt, tree id t = slot-(gt-1)) sees every c < gt plus c = gt-1+j for ancestors-or-self
 * j of t; nothing at c >= gt+n-1.  out: fp16 [q_len][out_stride].                            */
int sq_tree_mask_dense_f16(void* out, int out_stride, int n_cols,
                           int q_slot0, int q_len, int gt, int n_tree,
                           const uint64_t* d_bitmask, int words, void* stream);

/* ---- a5: KV slot scatter / accepted-path compaction ------------------------------------- */
/* KV_Cache.update_kv_cache for one layer (Engine/Llama_KV.py:72-89):
 * cache[h][storage_ids[i]][:] = new[h][i][:].  new_k/new_v: fp16 [H_kv][q_len][D].          */
int sq_kv_scatter_f16(void* k_layer, void* v_layer, const void* new_k, const void* new_v,
                      const int64_t* d_storage_ids, int q_len, int h_kv, int m, int d,
                      void* stream);

/* KV_Cache.gather_kv_incremental over all layers (Engine/Llama_KV.py:60-68):
 * for j < count: cache[..., dst_offset+j, :] = cache[..., slots[j], :] (all reads of a
 * (layer, head) tile happen before its writes, so overlapping src/dst is safe);
 * then rows [dst_offset+count, zero_end) are zeroed (zero_end <= m; pass zero_end = m for the
 * reference's full-tail clear, gt+n-1 for dirty-range only, 0 for none).
 * count = *d_count if d_count != NULL else max_count.  slots are ascending absolute slots.
 * d_dst_offset (optional, device int32) overrides dst_offset with a value read on the device (the device-driven
 * step's ground-truth length); zero_end must be 0 then.                                                          */
int sq_kv_compact_f16(void* k_cache, void* v_cache, int n_layers, int h_kv, int m, int d,
                      const int32_t* d_slots, const int32_t* d_count, int max_count,
                      int dst_offset, int zero_end, const int32_t* d_dst_offset, void* stream);

/* The same roll-back for TWO caches in one launch (the speculation step compacts the draft cache and the target cache
 * with the same accepted slots and the same destination, Tree/SpecTree.py:226-227): cache 0 and cache 1 may differ in
 * layers / KV heads / M / D.  No tail zeroing in this form (the device-driven step never zeroes).                    */
int sq_kv_compact2_f16(void* k0, void* v0, int n_layers0, int h_kv0, int m0, int d0, void* k1, void* v1, int n_layers1,
                       int h_kv1, int m1, int d1, const int32_t* d_slots, const int32_t* d_count, int max_count,
                       int dst_offset, const int32_t* d_dst_offset, void* stream);

/* KV_Cache.clear (Engine/Llama_KV.py:91-94) restricted to rows [0, used_rows) of every
 * (layer, head): rows never written are already zero.                                       */
int sq_kv_clear_f16(void* k_cache, void* v_cache, int n_layers, int h_kv, int m, int d,
                    int used_rows, void* stream);

/* ---- a3/a4: RoPE + KV write + tree-batched attention ------------------------------------ */
/* Fused replacement of apply_rotary_pos_emb (Engine/offload_engine.py:42-67 semantics, called
 * at Engine/Llama_modules.py:118,214) + update_kv_cache (:120,217).
 * qkv: fp16 [q_len][(H + 2*H_kv)*D] (row stride qkv_stride elements), packed q | k | v.
 * q_out: fp16 [H][q_len][D]; rotated K and V rows are written to cache slot storage_ids[i].
 * cos/sin: fp16 [max_pos][D] tables exactly as LlamaRotaryEmbedding_FI builds them
 * (Engine/Llama_modules.py:31-45); arithmetic rounds to fp16 after every op like the
 * reference's fp16 tensor expression q*cos + rotate_half(q)*sin.                             */
int sq_rope_kv_write_f16(const void* qkv, int qkv_stride, void* q_out,
                         void* k_layer, void* v_layer,
                         const void* cos_tab, const void* sin_tab,
                         const int64_t* d_position_ids, const int64_t* d_storage_ids,
                         int q_len, int n_heads, int h_kv, int d, int m, void* stream);

/* The same with the packed q | k | v rows supplied as the split-K partials of the tall-skinny projection
 * (sq_linear_ts_f16 with splits > 1): fp32 [splits][q_len][qkv_stride]; a value is the sum of its partials in split
 * order rounded to fp16, i.e. exactly what the projection would have written.                                       */
int sq_rope_kv_write_slabs_f16(const float* qkv_slab, int splits, int qkv_stride, void* q_out, void* k_layer,
                               void* v_layer, const void* cos_tab, const void* sin_tab,
                               const int64_t* d_position_ids, const int64_t* d_storage_ids, int q_len,
                               int n_heads, int h_kv, int d, int m, void* stream);

/* Host-side view of sq_tree_attention_f16's block map (no device work): *n_blocks = the grid size for these head counts;
 * for 0 <= block < *n_blocks, (head, 16-query tile) the block computes, or (-1, -1) for a block without work.  The map
 * keeps every query tile of a KV head on one XCD (block % 8) -- on as few XCDs as give each workgroup a compute unit when
 * there are fewer than 8 KV heads (tensor-parallel shards) -- and must cover every (head, tile) exactly once.
 * block < 0 (or head / tile NULL): only *n_blocks is written.                                                          */
int sq_tree_attention_block_decode(int block, int q_len, int n_heads, int h_kv, int* head, int* tile, int* n_blocks);

/* Tree-batched attention for one layer (LlamaAttention_FI.forward Engine/Llama_modules.py:
 * 124-134 and LlamaAttention_TG.forward :220-248): out = softmax(q k^T * scale + mask) v over
 * key slots [0, kv_len), fp32 softmax, MFMA for q k^T and p v.
 * q: fp16 [H][q_len][D]; out: fp16 [q_len][H*D] (the layout o_proj consumes).
 * mask_mode 0: dense additive fp16 mask [q_len][mask_stride] (the reference's attn_mask);
 * mask_mode 1: implicit tree mask = same rule as sq_tree_mask_dense_f16, query i sits at slot
 *              q_slot0 + i.
 * mask_mode | SQ_ATT_OUT_FRAG: out is written as the fragment-major activation image
 *              [H*D / 32][ceil(q_len / 16)][64][8] that sq_linear_ts_f16 (the o_proj) consumes.
 * d_ctx (optional, device int32[3]): when non-NULL the kernel takes {q_slot0, gt, kv_len} from
 * it instead of the by-value arguments, so a captured launch can be replayed for other steps. */
#define SQ_ATT_OUT_FRAG 0x100
int sq_tree_attention_f16(const void* q, const void* k_layer, const void* v_layer, void* out,
                          int q_len, int n_heads, int h_kv, int d, int m, int kv_len,
                          float scale, int mask_mode,
                          const void* dense_mask, int mask_stride,
                          int q_slot0, int gt, int n_tree,
                          const uint64_t* d_bitmask, int words, const int32_t* d_ctx,
                          void* stream);

/* Stream-ordered store of up to four int32 values into device memory (dst[0..n)); used to
 * update a d_ctx block between graph replays without a host->device copy.                    */
int sq_store_i32(int32_t* dst, int n, int v0, int v1, int v2, int v3, void* stream);

/* One launch that stages a captured forward's inputs: the q_len new tokens' input_ids / position_ids / storage_ids are
 * copied into the graph's static buffers and (d_ctx != NULL) its {q_slot0, gt, kv_len} block is written -- the four
 * copy_ calls of the reference's capture_graph closure (Engine/Engine.py:156-163).  All arrays int64, contiguous. */
int sq_stage_inputs(int64_t* dst_ids, const int64_t* src_ids, int64_t* dst_pos, const int64_t* src_pos,
                    int64_t* dst_storage, const int64_t* src_storage, int q_len, int32_t* d_ctx, int q_slot0, int gt,
                    int kv_len, void* stream);

/* The same staging for the device-driven step: queries at slots [gt + rel_slot0, gt + rel_slot0 + q_len) with
 * gt = d_step[SQ_STEP_GT]; ids from `tokens`, storage ids = slots, position ids = slot for committed text and
 * d_depth[t] + gt - 1 for tree node t (Tree/SpecTree.py:61,264-270), d_ctx = {q_slot0, gt, gt + rel_kv_len}.
 * advance != 0 first moves d_step to the next step (gt <- next_gt, index += 1).                                    */
int sq_stage_tree_inputs(int64_t* dst_ids, int64_t* dst_pos, int64_t* dst_storage, int32_t* d_ctx,
                         const int64_t* tokens, const int32_t* d_depth, int n_tree, int q_len, int rel_slot0,
                         int rel_kv_len, int32_t* d_step, int advance, void* stream);

/* ---- a2: draft expansion samplers ------------------------------------------------------- */
/* Every logits row is processed by ceil(V / 4096) workgroups ("parts"): per-part softmax statistics, keys + local top-k
 * per part, one-wave merge per row -- a tree level of 1-34 rows runs on 8-272 compute units.  Scratch is caller-owned:
 * sq_sample_workspace_bytes(n_rows, vocab, k) bytes, reusable by every later sampler call on the same stream.      */
size_t sq_sample_workspace_bytes(int n_rows, int vocab, int k);

/* Per-part statistics of softmax(logits / T): d_stats[(row * parts + p) * 2 + {0, 1}] = {max_p h(x/T), sum_p exp(. - max_p)}
 * with parts = ceil(V / 4096); `row` is the launch row r (stats_by_source_row == 0) or the source row d_row_ids[r]
 * (!= 0: a table indexed like the logits matrix).  copy_dst != NULL additionally copies launch row r to
 * copy_dst + r * ld_dst (the speculation step uses this launch instead of the copy
 * `draft_logits[first:first+total] = logits[-total:]`, Tree/SpecTree.py:121).                                       */
int sq_logits_stats_f16(const void* logits, int64_t ld, const int32_t* d_row_ids, int n_rows, int vocab,
                        float temperature, float* d_stats, int stats_by_source_row, void* copy_dst, int64_t ld_dst,
                        void* stream);

/* utils.sampling_without_replacement (utils.py:10-18) for n_rows rows:
 * q = softmax(logits/T) (fp16), key = log(u)/q (fp16), take the k largest keys per row in
 * descending order (ties: lower token id first).
 * logits: fp16 rows of length V, row r at logits + row_ids[r]*ld_logits (row_ids NULL = r);
 * rand likewise with ld_rand.  Output, int64:
 *   d_branch == NULL : out[r*k + s] = s-th sample of row r               (callable contract)
 *   d_branch != NULL : out[d_out_off[r] + s] for s < d_branch[r]          (fused gather,
 *                      = tokens[num_nodes:...] = new[sample_gather_indices], SpecTree.py:104)
 * d_out_base (optional, device int32): element offset added to every output index, read on the device -- the
 *   speculation step passes its device-resident ground-truth length so that no launch argument depends on the step.
 * d_stats (optional): statistics of the SOURCE rows from an earlier sq_logits_stats_f16(..., stats_by_source_row = 1)
 *   over the same logits matrix; NULL = computed here (one more launch).                                             */
int sq_sample_wor_f16(const void* logits, int64_t ld_logits, const void* rand, int64_t ld_rand,
                      const int32_t* d_row_ids, int n_rows, int vocab, int k, float temperature,
                      int64_t* out, const int32_t* d_branch, const int32_t* d_out_off,
                      const int32_t* d_out_base, const float* d_stats, void* workspace, void* stream);

/* utils.sampling_argmax (utils.py:29-32): top-k token ids of the raw logits, descending,
 * ties: lower token id first.  Same addressing / output modes as sq_sample_wor_f16.          */
int sq_topk_f16(const void* logits, int64_t ld_logits, const int32_t* d_row_ids, int n_rows,
                int vocab, int k, int64_t* out, const int32_t* d_branch,
                const int32_t* d_out_off, const int32_t* d_out_base, void* workspace, void* stream);

/* ---- a6/a7/a8: verification ------------------------------------------------------------- */
size_t sq_verify_workspace_bytes(int n_tree);

/* SpecTree.verify after the target forward (Tree/SpecTree.py:196-227) with accept_step
 * (:136-157) and get_residual (utils.py:5-8), top_p = 1:
 *   p_t = softmax(target_logits[t]/T) for every node t; walk from the root: for the children c
 *   of the current node in order, accept c iff p[tok_c] > r[slot_c]*q[tok_c] (fp16, strict);
 *   on reject p <- relu(p-q)/sum(relu(p-q)), draft_logits[node][tok_c] <- -65504,
 *   q <- softmax(draft_logits[node]/T).  Stop at the first node with no accepted child; an
 *   accepted token in {0,2} makes the step terminal (reason 1); a NaN residual makes it
 *   terminal (reason 2); otherwise the bonus token is drawn from the residual by exact
 *   inverse-CDF with the caller's 24-bit uniform `bonus_u24` (replaces multinomial(1), :222).
 * Side effects (all in device memory, no host sync):
 *   tokens[a] = bonus, THEN tokens[gt .. a) = tokens[accepted slots]      (:222, :224 -- the reference's
 *     order: an accepted node sitting at slot a, i.e. tree node n_accepted + 1 on the accepted path, is
 *     committed with the bonus token's id.  OR SQ_VERIFY_GATHER_FIRST into bonus_u24 for the lossless
 *     order: gather first, then the bonus write)
 *   draft_logits rows of the walked nodes get the -65504 writes           (:156)
 *   result record filled (see SQ_RES_*).
 * target_logits: fp16 [n_tree][V]; draft_logits: fp16 [>= n_tree][V] (tree-local rows);
 * tokens: int64 [token_capacity]; r: fp16 [M]; children CSR over tree-local ids.
 * token_capacity > 0: a step whose bonus slot a would lie outside the token buffer is made terminal (reason 3)
 *   instead of writing out of bounds (the reference raises IndexError there); 0 = unchecked.
 * Device-driven step (d_step != NULL, see SQ_STEP_*): gt is read from d_step[SQ_STEP_GT] (the `gt` argument is
 *   ignored), the bonus uniform is d_bonus_u24[d_step[SQ_STEP_INDEX] % n_bonus] when d_bonus_u24 != NULL (else
 *   bonus_u24; its flag bits always apply), the walker writes d_step[SQ_STEP_NEXT_GT] and clears
 *   d_step[SQ_STEP_ACTIVE] on a terminal step, and copies the record header to slot index % SQ_RESULT_RING of
 *   d_result_ring (optional) for a host that reads results one step late.                                        */
#define SQ_VERIFY_GATHER_FIRST 0x80000000u
int sq_verify_stochastic_f16(const void* target_logits, void* draft_logits,
                             int64_t* tokens, int token_capacity, const void* r,
                             const int32_t* d_child_off, const int32_t* d_child_ids,
                             int n_tree, int vocab, int gt, float temperature,
                             uint32_t bonus_u24, void* workspace, int32_t* d_result,
                             int32_t* d_step, const uint32_t* d_bonus_u24, int n_bonus,
                             int32_t* d_result_ring, void* stream);

/* utils.get_sampling_logits (utils.py:65-77), applied by SpecTree.verify to the target logits
 * before the softmax (Tree/SpecTree.py:196): in place, rows [n_rows][ld]; a token is set to -inf
 * iff the softmax(logits/T) mass of the tokens ranked before it (descending logit, ties by token
 * id) exceeds top_p (compared in fp16 like the reference's fp16 cumsum).  Identity for top_p >= 1. */
int sq_top_p_filter_f16(void* logits, int64_t ld, int n_rows, int vocab, float top_p,
                        float temperature, void* stream);

/* GreedyTree.verify after the target forward (Tree/GreedyTree.py:186-207): argmax per node,
 * walk by token equality, bonus = target argmax at the last accepted node.  tokens are only
 * compacted / bonus written when not terminal... the compaction tokens[:a] happens always
 * (:204), the bonus write only when not terminal (:206-207).                                */
int sq_verify_greedy_f16(const void* target_logits, int64_t* tokens, int token_capacity,
                         const int32_t* d_child_off, const int32_t* d_child_ids,
                         int n_tree, int vocab, int gt, void* workspace, int32_t* d_result,
                         int32_t* d_step, int32_t* d_result_ring, void* stream);

/* ---- the comparison baselines of the paper on the same kernels ------------ */
/* SpecInferTree.collective_grow_static (Tree/SpecInferTree.py:104-109): k draws WITH replacement per
 * row from softmax(logits / T).  Draw j of row i is the exact inverse CDF (token order, integer
 * arithmetic on the 2^-24 grid) at d_u24[i*k + j] / 2^24 -- explicit uniforms replace torch's device
 * multinomial stream.  Rows / output placement as in sq_sample_wor_f16.                          */
int sq_sample_iid_f16(const void* logits, int64_t ld, const int32_t* d_row_ids, int n_rows, int vocab,
                      int k, float temperature, const uint32_t* d_u24, int64_t* d_out,
                      const int32_t* d_branch, const int32_t* d_out_off, void* stream);

/* SpecInferTree.verify / accept_step (Tree/SpecInferTree.py:141-164,167-247): as
 * sq_verify_stochastic_f16, except that a child is accepted iff p[tok] >= r q[tok] and a rejection
 * only replaces p by the residual -- q keeps the rejected token (the children were drawn with
 * replacement) and the draft logits are not modified.                                           */
int sq_verify_specinfer_f16(const void* target_logits, const void* draft_logits, int64_t* tokens,
                            int token_capacity, const void* r, const int32_t* d_child_off, const int32_t* d_child_ids,
                            int n_tree, int vocab, int gt, float temperature, uint32_t bonus_u24,
                            void* workspace, int32_t* d_result, void* stream);

/* The acceptance-rate probe (tests/test_accept.py:36-140 on SpecTreeTest, Tree/SpecTree.py:283-481): a
 * star tree whose children are sampled without replacement with FP32 noise -- torch evaluates rand.log() / q in fp32, so
 * the keys are fp32 (sq_sample_wor_f32noise_f16; arguments as sq_sample_wor_f16, rand fp32) -- and verified with r in
 * fp32 and the test p >= r q evaluated in fp32 (sq_verify_probe_f16; arguments as sq_verify_stochastic_f16, r fp32).   */
int sq_sample_wor_f32noise_f16(const void* logits, int64_t ld_logits, const float* rand, int64_t ld_rand,
                               const int32_t* d_row_ids, int n_rows, int vocab, int k, float temperature,
                               int64_t* out, const int32_t* d_branch, const int32_t* d_out_off, void* workspace,
                               void* stream);
int sq_verify_probe_f16(const void* target_logits, void* draft_logits, int64_t* tokens, int token_capacity,
                        const float* r32, const int32_t* d_child_off, const int32_t* d_child_ids, int n_tree,
                        int vocab, int gt, float temperature, uint32_t bonus_u24, void* workspace,
                        int32_t* d_result, void* stream);

/* GreedySTree.verify (Tree/GreedySTree.py:188-214): the walk of sq_verify_greedy_f16 against one
 * target token per node supplied by the caller (sampled from the target distribution instead of
 * the argmax); bonus = the target token of the last accepted node.  d_target_tokens: int64 [n_tree]. */
int sq_verify_tokens_f16(const int64_t* d_target_tokens, int64_t* tokens, int token_capacity,
                         const int32_t* d_child_off, const int32_t* d_child_ids, int n_tree, int gt,
                         void* workspace, int32_t* d_result, void* stream);

/* ---- row-wise glue of the Llama block (launch removal on the draft side) --- */
/* LlamaRMSNorm_FI.forward (Engine/Llama_modules.py:274-288): fp32 variance, normalised value
 * cast to fp16, then fp16 multiply by the weight.  x, out: fp16 [rows][hidden].              */
int sq_rmsnorm_f16(const void* x, const void* weight, void* out, int rows, int hidden, float eps,
                   void* stream);
/* residual add + RMSNorm of the decoder layer (Engine/Llama_modules.py:341-346):
 * sum_out = x + residual (fp16 add; sum_out may alias residual), out = rmsnorm(sum_out)*weight */
int sq_add_rmsnorm_f16(const void* x, const void* residual, void* sum_out, const void* weight,
                       void* out, int rows, int hidden, float eps, void* stream);
/* The same three row-wise ops with the output written as the fragment-major activation image
 * [hidden / 32][ceil(rows / 16)][64][8] of sq_linear_ts_f16 (hidden % 32 == 0); identical arithmetic.    */
int sq_rmsnorm_frag_f16(const void* x, const void* weight, void* out_frag, int rows, int hidden, float eps, void* stream);
int sq_add_rmsnorm_frag_f16(const void* x, const void* residual, void* sum_out, const void* weight, void* out_frag,
                            int rows, int hidden, float eps, void* stream);
int sq_silu_mul_frag_f16(const void* gate_up, void* out_frag, int rows, int inter, void* stream);

/* LlamaMLP_FI (Engine/Llama_modules.py:270-271): out = act_fn(gate) * up in fp16.
 * gate_up: fp16 [rows][2*inter] packed gate | up (one fused GEMM); out: fp16 [rows][inter].  */
int sq_silu_mul_f16(const void* gate_up, void* out, int rows, int inter, void* stream);

/* Fragment-major operand images of the tall-skinny linear layer (MFMA 16x16x32 operand order: lane =
 * (k / 8 % 4) * 16 + row % 16 holds 8 consecutive k of its row, so every wave-wide operand load is 1 KB contiguous):
 *   weights      w_f[n / 16][k / 32][lane][8]   <- w [n][k] row-major (nn.Linear.weight), once at load;
 *   activations  x_f[k / 32][ceil(m / 16)][lane][8], rows >= m zero  <- x [m][ldx] row-major.
 * n % 16 == 0, k % 32 == 0; out of place.                                                                    */
int sq_repack_linear_weight_f16(const void* w, void* w_frag, int n, int k, void* stream);
int sq_repack_rows_frag_f16(const void* x, int ldx, void* x_frag, int m, int k, void* stream);

/* Tall-skinny linear layer of a tree forward (m <= 144 rows: one tree / tree level), nn.Linear semantics
 * out = a . w^T, fp32 accumulation, fp16 output -- the dense projections of LlamaAttention_FI/TG and LlamaMLP_FI
 * (Engine/Llama_modules.py:104-112,138,199-207,256,262-271) when q_len <= 144, as an HBM weight stream over
 * fragment-major operands (above).  The launch has tiles x splits workgroups: the n_out / 16 column units are
 * partitioned over `tiles` workgroups, K over `splits`.  Units per workgroup (SQ_EUNSUPPORTED beyond): plain <= 8 up to
 * 128 rows, <= 6 for 129-144; silu (gate+up pairs) <= 4 up to 129 rows, <= 3 for 130-144.  m == 129 (a 128-node tree and
 * its root) runs 8 MFMA row tiles with the extra row on the vector ALU instead of 9 row tiles (csrc/ts_linear.hip).
 *   splits == 1, silu != 0   : w_frag holds gate tiles [0, n_out/16) then up tiles;
 *                              out = h(h(silu(h(g))) * h(u))                                         (:271)
 *   splits == 1, res != NULL : out = h(h(acc) + res) (fp16 add, the decoder layer's skip connection
 *                              :341-346; out may alias res; row-major output only)
 *   splits == 1, out_frag    : out is written as the fragment-major activation image of the next layer
 *                              ([n_out / 32][ceil(m / 16)][64][8]; n_out % 32 == 0), else row-major [m][ldo]
 *   splits  > 1              : no out / res / silu; split s writes its fp32 partial product to
 *                              slab[s][m][n_out] (sq_linear_ts_workspace_bytes bytes); the consumer
 *                              (sq_add_rmsnorm_slabs_f16) sums the splits in order.
 * k % 32 == 0, n_out % 16 == 0, 16-byte aligned pointers.                                                 */
size_t sq_linear_ts_workspace_bytes(int m, int n_out, int splits);
int sq_linear_ts_f16(const void* a_frag, const void* w_frag, const void* res, void* out, int ldo, int out_frag, int m,
                     int n_out, int k, int silu, int tiles, int splits, void* slab, size_t slab_bytes, void* stream);

#ifdef SEQUOIA_BUILD_PROBES
/* Measurement aid, NOT part of the library's default surface: built only with SEQUOIA_BUILD_PROBES=1 (sequoia_amd/build.py
 * then adds csrc/ts_probe.hip), used by tools/prefetch_probe.py.  Touches, from `grid` workgroups, one dword of every
 * 128-byte line that the workgroups of sq_linear_ts_f16(..., tiles, splits) load in their first `depth` k-steps, so that
 * those lines sit in the XCD L2s when the projection starts (profiles/r03_prefetch_probe.md).                        */
int sq_linear_ts_prefetch(const void* w_frag, int n_out, int k, int silu, int tiles, int splits, int depth, int grid,
                          void* sink, void* stream);
#endif

/* Embedding lookup + the first RMSNorm of a forward in one pass (Engine/Llama_model.py:151 + the first layer's
 * input_layernorm, Engine/Llama_modules.py:282-288): x_out[r] = embed[ids[r]] (the residual stream, row-major),
 * out = RMSNorm(x_out) * weight, row-major or (out_frag) fragment-major.  ids: int64 [rows], clamped to [0, vocab).   */
int sq_embed_rmsnorm_f16(const int64_t* d_ids, const void* embed, int vocab, const void* weight, void* x_out, void* out,
                         int out_frag, int rows, int hidden, float eps, void* stream);
/* The same pass with the forward's inputs staged in it (device-driven speculation step): sq_stage_tree_inputs's work --
 * ids / storage ids / position ids of the `rows` queries at slots [gt + rel_slot0, ...), the {q_slot0, gt, kv_len}
 * context block, the step block's advance -- done by the workgroups that then look up and normalise the rows, so a
 * forward starts with one launch instead of two (same arguments and meaning as sq_stage_tree_inputs + the call above). */
int sq_embed_stage_rmsnorm_f16(int64_t* dst_ids, int64_t* dst_pos, int64_t* dst_storage, int32_t* d_ctx,
                               const int64_t* tokens, const int32_t* d_depth, int n_tree, int rel_slot0, int rel_kv_len,
                               int32_t* d_step, int advance, const void* embed, int vocab, const void* weight, void* x_out,
                               void* out, int out_frag, int rows, int hidden, float eps, void* stream);

/* Residual add + RMSNorm fed by a split-K linear layer: x = h(sum_s slab[s]) (the layer's fp16 output),
 * then exactly sq_add_rmsnorm_f16: sum_out = x + residual (fp16), out = RMSNorm(sum_out) * weight
 * (Engine/Llama_modules.py:282-288,341-346).  out == NULL skips the normalisation (last layer's skip add
 * feeding the final norm is done by a separate call); out_frag != 0 writes `out` fragment-major.
 * slab: fp32 [splits][rows][hidden].                                                                  */
int sq_add_rmsnorm_slabs_f16(const void* slab, int splits, const void* residual, const void* weight, void* sum_out,
                             void* out, int out_frag, int rows, int hidden, float eps, void* stream);

/* SwiGLU fed by a split-K gate|up projection (the layer run as a plain [2 inter] x k sq_linear_ts_f16 with splits > 1:
 * shapes whose activation block outweighs the weights, e.g. tensor-parallel shards at 129 rows): slab = fp32
 * [splits][rows][2 inter] partials (gate columns, then up columns); out = h(h(silu(h(sum gate))) * h(sum up))
 * (Engine/Llama_modules.py:271), fragment-major when out_frag != 0 (inter % 32 == 0), else row-major [rows][inter].  */
int sq_silu_mul_slabs_f16(const void* slab, int splits, void* out, int out_frag, int rows, int inter, void* stream);

/* ---- e: tensor-parallel all-reduce over peer-mapped buffers (xGMI) ------------------------------------------
 * The 70B target replaces the reference's host offload (Engine/offload_engine.py:388-451) by tensor parallelism over
 * the GPUs of one node: the row-parallel projections (o_proj, down_proj: Engine/Llama_modules.py:138,256,271 on a
 * shard) leave a partial [q, hidden] fp16 product on every rank, 2 per layer.  sq_allreduce_sum_f16 sums them in
 * place over buffers the ranks map into each other's address space (hipIpc), every rank talking to every peer at
 * once (xGMI is point-to-point): reduce-scatter by direct peer stores, sum in fp32 in rank order (one rounding: all
 * ranks get bit-identical rows), all-gather by direct peer stores; flags carry a device-resident epoch, so the launch
 * replays from a hipGraph.  Stream-ordered, no host synchronisation, every spin bounded (sq_ar_status reports a
 * timeout instead of hanging).  RCCL (torch.distributed) remains the fallback.
 *   setup, once per rank: ws = sq_ar_alloc(sq_ar_workspace_bytes(world, max_elems, max_gather_elems)) (uncached, zeroed),
 *   sq_ar_ipc_export(ws, handle) -> exchange the 64-byte handles (any transport) -> sq_ar_ipc_open(peer handle);
 *   call: ws[world] = every rank's workspace as mapped in the calling process (ws[rank] = own), identical n / blocks
 *   on all ranks, n % 8 == 0, n <= max_elems; blocks <= 0 picks one block per 4 KB of a chunk (max 64).             */
size_t sq_ar_workspace_bytes(int world, size_t max_elems, size_t max_gather_elems);
int sq_ar_alloc(void** ptr, size_t bytes);
int sq_ar_free(void* ptr);
int sq_ar_ipc_export(void* ptr, void* handle64);
int sq_ar_ipc_open(const void* handle64, void** ptr);
int sq_ar_ipc_close(void* ptr);
int sq_ar_status(const void* own_ws, int* status);      /* 0 ok; bit 0 / 1: a phase-1 / phase-2 flag never arrived; bit 2 / 3: the
                                                           all-gather's "written" / "read" flag (host sync)                  */
/* Timeouts must not pass silently (a collective that gave up continues on stale areas): host_word = the device-visible
 * address of a uint32 in pinned host memory; every timeout ORs its status bits into it as well, so the host loop can test
 * it at every step without a device read (Tree/_native_tree.py raises on it).  NULL clears the registration.             */
int sq_ar_set_fault_word(void* own_ws, void* host_word);
/* Test rig (SEQUOIA_AR_WS=host): the workspace as a POSIX shared-memory object in fine-grained host memory, mapped by
 * every rank and registered with the HIP runtime -- every store, flag and poll of the protocol then leaves the device
 * (PCIe), which two ranks sharing ONE GPU's memory never do.  create != 0: create + zero `name`; returns the host mapping
 * and the device-visible pointer to pass in ws[].  Close: unregister + unmap (+ unlink when name_to_unlink != NULL).     */
int sq_ar_shared_host_open(const char* name, size_t bytes, int create, void** host_ptr, void** dev_ptr);
int sq_ar_shared_host_close(const char* name_to_unlink, void* host_ptr, size_t bytes);
int sq_allreduce_sum_f16(void* data, size_t n, int rank, int world, void* const* ws, size_t max_elems, int blocks,
                         void* stream);
/* The same all-reduce fed by a split-K row-parallel projection: slab = fp32 [splits][n] partial products of this rank
 * (sq_linear_ts_f16, splits > 1); the rank's contribution is h(sum_s slab[s]) -- the fp16 rows sq_add_rmsnorm_slabs_f16
 * would have materialised first -- and `out` [n] fp16 receives the sum over the ranks (one launch instead of two).      */
int sq_allreduce_sum_slabs_f16(const void* slab, int splits, void* out, size_t n, int rank, int world, void* const* ws,
                               size_t max_elems, int blocks, void* stream);
/* All-gather of the vocabulary-parallel lm_head (column-parallel over the ranks, Engine/Llama_model.py:280-283 on a
 * shard): slice = this rank's [rows][v] fp16 logits, out = the full [rows][world v] rows (rank r's columns at
 * [r v, (r + 1) v)), by direct peer stores into the same workspaces (their own area: rows world v <=
 * max_gather_elems, the value the workspace was sized with).  v % 8 == 0.
 * v must be THE SAME in consecutive gathers on one workspace unless the caller synchronises all ranks in between: the
 * per-(peer, block) "read" handshake orders a gather behind the previous one row by row (rows are dealt to blocks by
 * row % blocks), and a row's bytes in the peer's image sit at i world v + rank v -- with another v a block would overwrite
 * bytes a different block of the peer may still be reading (Engine/xgmi_allreduce.py::gather_cols enforces it).        */
int sq_allgather_cols_f16(const void* slice, void* out, int rows, int v, int rank, int world, void* const* ws,
                          size_t max_elems, size_t max_gather_elems, void* stream);
/* All-reduce of a row-parallel projection + the decoder layer's continuation in ONE launch (replaces the all-reduce
 * followed by sq_add_rmsnorm_f16 / _frag_f16; Engine/Llama_modules.py:282-288,341-346 on a tensor-parallel shard):
 *   x[r]  <- h(x[r] + sum over ranks of partial[r])        (skip connection, residual stream updated in place)
 *   out[r] = h(weight * h(x[r] * rsqrt(mean(x[r]^2) + eps)))  (row-major, or the fragment-major operand image: out_frag)
 * partial = fp32 split-K slabs [splits][rows][hidden] (slab != NULL) or fp16 rows (in_rows).  The reduction is cut along
 * rows so that the block that gathers a row also normalises it; arithmetic and summation order are those of the
 * three-launch form, bit for bit.  ceil(rows / world) * hidden must fit a rank's chunk of the workspace.              */
int sq_allreduce_add_rmsnorm_f16(const void* slab, int splits, const void* in_rows, void* x, const void* weight, void* out,
                                 int out_frag, int rows, int hidden, float eps, int rank, int world, void* const* ws,
                                 size_t max_elems, void* stream);

#ifdef SEQUOIA_BUILD_PROBES
/* ---- f1: RMSNorm folded into the projection that consumes it (small draft models) ------------------------------
 * EXPERIMENTAL: measured 8 % slower than the unfused launch sequence on MI355X (profiles/r03_draft_fused_not_adopted.md);
 * NOT part of the default surface since round 6 (built with SEQUOIA_BUILD_PROBES=1 only, csrc/draft_fused.hip; opt-in at run
 * time with SEQUOIA_DRAFT_FUSED=1), kept for the measurement and its tests; the signature may change or disappear.
 * out = epilogue( (RMSNorm(x) * norm_weight) . w^T ) for m <= 48 rows and k in {256, 512, 768, 1024} (the 68m / 160m drafts): every
 * workgroup normalises the whole activation block itself into LDS, so the separate norm launch in front of
 * q/k/v_proj, gate/up_proj and lm_head (Engine/Llama_modules.py:282-288,341-346; Engine/Llama_model.py:280-283)
 * disappears.  x: the residual stream [m][k] fp16 row-major -- or, first layer, d_ids != NULL: row r is
 * embed[d_ids[r]] (Engine/Llama_model.py:151), written to x_out [m][k] when x_out != NULL.  w_frag: the
 * fragment-major weight image of sq_repack_linear_weight_f16 (swiglu: gate tiles then up tiles).  swiglu == 0: out is
 * fp16 [m][ldo]; swiglu != 0: out is the fragment-major image [n_out/32][ceil(m/16)][64][8] of
 * h(h(silu(h(g))) * h(u)) (:271).  tiles workgroups share the n_out / 16 column units (<= 8 per workgroup; swiglu <= 2).
 * Rounding points are those of sq_rmsnorm_f16 + sq_linear_ts_f16; only fp32 summation orders differ.                */
int sq_norm_linear_f16(const void* x, const int64_t* d_ids, const void* embed, int vocab, void* x_out,
                       const void* norm_weight, float eps, const void* w_frag, void* out, int ldo, int m, int n_out,
                       int k, int swiglu, int tiles, void* stream);
#endif

/* ---- f1: the attention half of a small draft model's decoder layer in ONE launch ---------------------------------
 * For forwards whose rows never attend to each other -- the draft forward over ONE tree level (Tree/SpecTree.py:87-134:
 * the new nodes of a level are siblings / cousins, none is another's ancestor) and any one-row forward -- of a model with
 * heads of 64 and hidden in {512, 768, 1024} (JackFram/llama-68m / -160m).  One workgroup per (head, 16-row tile) runs
 *   q | k | v = a . Wqkv[head]^T  ->  RoPE (fp16 rounding per op, Engine/offload_engine.py:63-66)  ->  K / V rows into the
 *   cache slots d_storage_ids (Engine/Llama_KV.py:72-89)  ->  tree attention over the cached keys [0, q_slot0) plus the
 *   row's own key  ->  the head's slice of o_proj as an fp32 partial slab[head][row][hidden],
 * i.e. LlamaAttention_FI.forward (Engine/Llama_modules.py:87-140) without its four kernel boundaries; the residual add +
 * RMSNorm that follows sums the head partials in head order (sq_add_rmsnorm_slabs_f16 with splits = n_heads).  Rounding
 * points are those of sq_linear_ts_f16 + sq_rope_kv_write_f16 + sq_tree_attention_f16; fp32 summation orders differ.
 * a_frag: fragment-major image of the normalised input rows (sq_rmsnorm_frag_f16 ...); wqkv_frag / wo_frag:
 * sq_repack_linear_weight_f16 images of the packed q|k|v weight [(3 H 64)][hidden] and of o_proj [hidden][H 64].
 * The caller guarantees: query i sits at slot q_slot0 + i, and no query may see another query's key (only its own).
 * Mask rule for the cached keys: as sq_tree_attention_f16 (mask_mode 1); d_ctx optionally overrides {q_slot0, gt}.
 * kv_only != 0: stop after the K / V rows are written (wo_frag / slab unused).                                        */
int sq_draft_attn_block_f16(const void* a_frag, const void* wqkv_frag, const void* wo_frag, float* slab, size_t slab_bytes,
                            void* k_layer, void* v_layer, const void* cos_tab, const void* sin_tab,
                            const int64_t* d_position_ids, const int64_t* d_storage_ids, int q_len, int n_heads, int d,
                            int hidden, int m, float scale, int q_slot0, int gt, int n_tree, const uint64_t* d_bitmask,
                            int words, const int32_t* d_ctx, int kv_only, void* stream);

/* RoPE + KV write + tree attention of one layer in ONE launch, for forwards whose rows never attend to each other (one tree
 * level, or one row) of ANY model with heads of 64 or 128: replaces sq_rope_kv_write_f16 or sq_rope_kv_write_slabs_f16 followed by
 * sq_tree_attention_f16 (LlamaAttention_FI.forward, Engine/Llama_modules.py:104-136, without the kernel boundary between
 * them).  qkv: the projection's fp16 rows [q_len][qkv_stride] -- or qkv_slab: its fp32 split-K partials
 * [splits][q_len][qkv_stride] (sq_linear_ts_f16), summed in split order and rounded to fp16 first.  out: fp16 [q_len][H d],
 * or (out_frag != 0) its fragment-major image for sq_linear_ts_f16.  Same caller guarantees and mask rule as
 * sq_draft_attn_block_f16; GQA: the first query head of a KV group writes the group's K / V rows.                        */
int sq_level_attention_f16(const void* qkv, const float* qkv_slab, int splits, int qkv_stride, void* out, int out_frag,
                           void* k_layer, void* v_layer, const void* cos_tab, const void* sin_tab,
                           const int64_t* d_position_ids, const int64_t* d_storage_ids, int q_len, int n_heads, int h_kv, int d,
                           int m, float scale, int q_slot0, int gt, int n_tree, const uint64_t* d_bitmask, int words,
                           const int32_t* d_ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEQUOIA_HIP_H */
