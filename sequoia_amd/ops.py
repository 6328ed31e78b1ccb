"""Torch-tensor front end of the C ABI: marshals tensors to raw pointers and launches on the
current HIP stream.  PyTorch is plumbing here (device memory + streams); all compute is in
libsequoia_hip.so.

`get_ops()` returns the process-wide HipOps instance.  It raises when the library is missing
or a tensor is not on a HIP device: the product path has no CPU implementation.  Tests may
install a checker implementation with `set_ops_for_testing()` (the numpy oracle adapter in
oracle/ops_adapter.py) to exercise the *host logic* on CPU; nothing in this package does.
"""
from __future__ import annotations

import torch

from . import native
from .native import SQ_ATT_OUT_FRAG, check


def _ptr(t):
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str, contiguous: bool = True):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if t.device.type != "cuda":
        raise native.SequoiaNativeError(
            f"{name} lives on {t.device}: the Sequoia hot path only runs on a HIP device (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


class HipOps:
    name = "hip"

    def __init__(self):
        self.lib = native.load()

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    # ---- a1 ---------------------------------------------------------------------------------
    def tree_mask_dense(self, out, q_slot0, gt, n_tree, bitmask):
        _need(out, torch.float16, "out", contiguous=False)
        assert out.dim() == 2 and out.stride(1) == 1
        words = 0
        if bitmask is not None:
            _need(bitmask, torch.int64, "bitmask")
            words = bitmask.shape[1]
        check(self.lib.sq_tree_mask_dense_f16(out.data_ptr(), out.stride(0), out.shape[1], q_slot0, out.shape[0], gt,
                                              n_tree, _ptr(bitmask), words, self._stream()), "sq_tree_mask_dense_f16")
        return out

    # ---- a5 ---------------------------------------------------------------------------------
    def kv_scatter(self, k_layer, v_layer, new_k, new_v, storage_ids):
        _need(k_layer, torch.float16, "k_layer"); _need(v_layer, torch.float16, "v_layer")
        _need(new_k, torch.float16, "new_k"); _need(new_v, torch.float16, "new_v")
        _need(storage_ids, torch.int64, "storage_ids")
        h_kv, m, d = k_layer.shape[-3:]
        q_len = storage_ids.shape[0]
        check(self.lib.sq_kv_scatter_f16(k_layer.data_ptr(), v_layer.data_ptr(), new_k.data_ptr(), new_v.data_ptr(),
                                         storage_ids.data_ptr(), q_len, h_kv, m, d, self._stream()), "sq_kv_scatter_f16")

    def kv_compact(self, k_cache, v_cache, slots, count, max_count, dst_offset, zero_end, dst_offset_dev=None):
        """k_cache/v_cache: [L, 1, H, M, D]; slots: int32 device tensor; count: int32[1] device tensor or None;
        dst_offset_dev: int32 device scalar overriding dst_offset (device-driven step)."""
        _need(k_cache, torch.float16, "k_cache"); _need(v_cache, torch.float16, "v_cache")
        if max_count > 0:
            _need(slots, torch.int32, "slots")
        if count is not None:
            _need(count, torch.int32, "count", contiguous=False)
        n_layers = k_cache.shape[0]
        h_kv, m, d = k_cache.shape[-3:]
        check(self.lib.sq_kv_compact_f16(k_cache.data_ptr(), v_cache.data_ptr(), n_layers, h_kv, m, d, _ptr(slots),
                                         _ptr(count), max_count, dst_offset, zero_end, _ptr(dst_offset_dev),
                                         self._stream()), "sq_kv_compact_f16")

    def kv_compact2(self, kv0, kv1, slots, count, max_count, dst_offset, dst_offset_dev=None):
        """Both caches of a step (kv0, kv1: objects with k_cache / v_cache [L, 1, H, M, D]) rolled back in one launch."""
        for kv in (kv0, kv1):
            _need(kv.k_cache, torch.float16, "k_cache"); _need(kv.v_cache, torch.float16, "v_cache")
        if max_count > 0:
            _need(slots, torch.int32, "slots")
        if count is not None:
            _need(count, torch.int32, "count", contiguous=False)
        a = []
        for kv in (kv0, kv1):
            h_kv, m, d = kv.k_cache.shape[-3:]
            a += [kv.k_cache.data_ptr(), kv.v_cache.data_ptr(), kv.k_cache.shape[0], h_kv, m, d]
        check(self.lib.sq_kv_compact2_f16(*a, _ptr(slots), _ptr(count), max_count, dst_offset, _ptr(dst_offset_dev),
                                          self._stream()), "sq_kv_compact2_f16")

    def kv_clear(self, k_cache, v_cache, used_rows):
        _need(k_cache, torch.float16, "k_cache"); _need(v_cache, torch.float16, "v_cache")
        n_layers = k_cache.shape[0]
        h_kv, m, d = k_cache.shape[-3:]
        check(self.lib.sq_kv_clear_f16(k_cache.data_ptr(), v_cache.data_ptr(), n_layers, h_kv, m, d, used_rows,
                                       self._stream()), "sq_kv_clear_f16")

    # ---- a3 / a4 ----------------------------------------------------------------------------
    def rope_kv_write(self, qkv, q_out, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads, h_kv, d):
        _need(qkv, torch.float16, "qkv", contiguous=False)
        assert qkv.dim() == 2 and qkv.stride(1) == 1
        _need(q_out, torch.float16, "q_out"); _need(k_layer, torch.float16, "k_layer"); _need(v_layer, torch.float16, "v_layer")
        _need(cos, torch.float16, "cos"); _need(sin, torch.float16, "sin")
        _need(position_ids, torch.int64, "position_ids"); _need(storage_ids, torch.int64, "storage_ids")
        m = k_layer.shape[-2]
        check(self.lib.sq_rope_kv_write_f16(qkv.data_ptr(), qkv.stride(0), q_out.data_ptr(), k_layer.data_ptr(),
                                            v_layer.data_ptr(), cos.data_ptr(), sin.data_ptr(), position_ids.data_ptr(),
                                            storage_ids.data_ptr(), qkv.shape[0], n_heads, h_kv, d, m, self._stream()),
              "sq_rope_kv_write_f16")

    def rope_kv_write_slabs(self, slab, splits, n_cols, q_out, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads,
                            h_kv, d):
        """rope_kv_write on the split-K partials [splits][q_len][n_cols] (fp32) of the qkv projection."""
        _need(slab, torch.float32, "slab")
        _need(q_out, torch.float16, "q_out"); _need(k_layer, torch.float16, "k_layer"); _need(v_layer, torch.float16, "v_layer")
        _need(cos, torch.float16, "cos"); _need(sin, torch.float16, "sin")
        _need(position_ids, torch.int64, "position_ids"); _need(storage_ids, torch.int64, "storage_ids")
        q_len, m = position_ids.numel(), k_layer.shape[-2]
        assert slab.numel() >= splits * q_len * n_cols
        check(self.lib.sq_rope_kv_write_slabs_f16(slab.data_ptr(), int(splits), int(n_cols), q_out.data_ptr(), k_layer.data_ptr(),
                                                  v_layer.data_ptr(), cos.data_ptr(), sin.data_ptr(), position_ids.data_ptr(),
                                                  storage_ids.data_ptr(), q_len, n_heads, h_kv, d, m, self._stream()),
              "sq_rope_kv_write_slabs_f16")

    def store_i32(self, dst, values):
        _need(dst, torch.int32, "dst")
        v = list(values) + [0] * (4 - len(values))
        check(self.lib.sq_store_i32(dst.data_ptr(), len(values), int(v[0]), int(v[1]), int(v[2]), int(v[3]),
                                    self._stream()), "sq_store_i32")

    def stage_inputs(self, dst_ids, src_ids, dst_pos, src_pos, dst_storage, src_storage, ctx=None, q_slot0=0, gt=1, kv_len=1):
        """Copy a forward's new-token inputs into a graph's static buffers and write its context block, one launch."""
        q_len = src_ids.numel()
        for t, n in ((dst_ids, "dst_ids"), (src_ids, "src_ids"), (dst_pos, "dst_pos"), (src_pos, "src_pos"),
                     (dst_storage, "dst_storage"), (src_storage, "src_storage")):
            _need(t, torch.int64, n)
            assert t.numel() == q_len
        check(self.lib.sq_stage_inputs(dst_ids.data_ptr(), src_ids.data_ptr(), dst_pos.data_ptr(), src_pos.data_ptr(),
                                       dst_storage.data_ptr(), src_storage.data_ptr(), q_len, _ptr(ctx), int(q_slot0),
                                       int(gt), int(kv_len), self._stream()), "sq_stage_inputs")

    def stage_tree_inputs(self, dst_ids, dst_pos, dst_storage, ctx, tokens, depth, n_tree, rel_slot0, rel_kv_len, step,
                          advance=False):
        """Device-driven staging of a tree forward: queries at slots [gt + rel_slot0, ... + q_len), gt from `step`."""
        q_len = dst_ids.numel()
        for t, n in ((dst_ids, "dst_ids"), (dst_pos, "dst_pos"), (dst_storage, "dst_storage"), (tokens, "tokens")):
            _need(t, torch.int64, n)
        _need(ctx, torch.int32, "ctx"); _need(depth, torch.int32, "depth"); _need(step, torch.int32, "step")
        assert dst_pos.numel() == q_len and dst_storage.numel() == q_len and depth.numel() >= n_tree
        check(self.lib.sq_stage_tree_inputs(dst_ids.data_ptr(), dst_pos.data_ptr(), dst_storage.data_ptr(), ctx.data_ptr(),
                                            tokens.data_ptr(), depth.data_ptr(), int(n_tree), q_len, int(rel_slot0),
                                            int(rel_kv_len), step.data_ptr(), 1 if advance else 0, self._stream()),
              "sq_stage_tree_inputs")

    def tree_attention(self, q, k_layer, v_layer, out, kv_len, scale, dense_mask=None, q_slot0=0, gt=0, n_tree=0,
                       bitmask=None, ctx=None, out_frag=False):
        """q: [H, q_len, D]; k/v_layer: [H_kv, M, D]; out: [q_len, H*D], or (out_frag) its fragment-major image
        [H*D/32, ceil(q_len/16), 64, 8] for linear_ts."""
        frag_flag = SQ_ATT_OUT_FRAG if out_frag else 0
        _need(q, torch.float16, "q"); _need(k_layer, torch.float16, "k_layer"); _need(v_layer, torch.float16, "v_layer")
        _need(out, torch.float16, "out")
        n_heads, q_len, d = q.shape
        h_kv, m, _ = k_layer.shape[-3:]
        if dense_mask is not None:
            _need(dense_mask, torch.float16, "dense_mask", contiguous=False)
            dm = dense_mask.reshape(dense_mask.shape[-2], dense_mask.shape[-1]) if dense_mask.dim() != 2 else dense_mask
            assert dm.stride(1) == 1 and dm.shape[0] == q_len and dm.shape[1] >= kv_len
            rc = self.lib.sq_tree_attention_f16(q.data_ptr(), k_layer.data_ptr(), v_layer.data_ptr(), out.data_ptr(),
                                                q_len, n_heads, h_kv, d, m, kv_len, scale, 0 | frag_flag, dm.data_ptr(),
                                                dm.stride(0), 0, 1, 1, None, 0, None, self._stream())
        else:
            words = 0
            if bitmask is not None:
                _need(bitmask, torch.int64, "bitmask")
                words = bitmask.shape[1]
            rc = self.lib.sq_tree_attention_f16(q.data_ptr(), k_layer.data_ptr(), v_layer.data_ptr(), out.data_ptr(),
                                                q_len, n_heads, h_kv, d, m, kv_len, scale, 1 | frag_flag, None, 0, q_slot0, gt,
                                                n_tree, _ptr(bitmask), words, _ptr(ctx), self._stream())
        check(rc, "sq_tree_attention_f16")
        return out

    # ---- a2 ---------------------------------------------------------------------------------
    def _sample_ws(self, device, n_rows, vocab, k):
        """Sampler scratch, one buffer per device, grown outside graph captures (sized generously on first use)."""
        need = int(self.lib.sq_sample_workspace_bytes(max(int(n_rows), 1), int(vocab), int(k)))
        key = str(device)
        ws = getattr(self, "_samp_ws", {}).get(key)
        if ws is None or ws.numel() < need:
            if torch.cuda.is_current_stream_capturing():
                raise native.SequoiaNativeError("sampler workspace must be sized before graph capture (run the step eagerly once)")
            want = max(need, int(self.lib.sq_sample_workspace_bytes(native.SQ_MAX_TREE, int(vocab), 32)))
            old = ws
            ws = torch.empty(want, dtype=torch.uint8, device=device)
            if not hasattr(self, "_samp_ws"):
                self._samp_ws, self._samp_ws_retired = {}, []
            if old is not None:
                # a captured graph (whole-step graph, sampler callables) may have the old buffer's address baked into
                # its launches: it must stay allocated for the life of the process, not return to the caching allocator
                self._samp_ws_retired.append(old)
            self._samp_ws[key] = ws
        return ws

    @staticmethod
    def stats_shape(n_rows, vocab):
        return (n_rows, (vocab + 4095) // 4096, 2)

    def logits_stats(self, logits, temperature, stats, row_ids=None, by_source_row=False, copy_dst=None):
        """Per-part softmax statistics of logits rows (optionally copying the rows to copy_dst[r])."""
        _need(logits, torch.float16, "logits", contiguous=False); _need(stats, torch.float32, "stats")
        assert logits.dim() == 2 and logits.stride(1) == 1
        n_rows = row_ids.shape[0] if row_ids is not None else logits.shape[0]
        if row_ids is not None:
            _need(row_ids, torch.int32, "row_ids")
        ld_dst = 0
        if copy_dst is not None:
            _need(copy_dst, torch.float16, "copy_dst", contiguous=False)
            assert copy_dst.stride(1) == 1 and copy_dst.shape[0] >= n_rows and copy_dst.shape[1] == logits.shape[1]
            ld_dst = copy_dst.stride(0)
        check(self.lib.sq_logits_stats_f16(logits.data_ptr(), logits.stride(0), _ptr(row_ids), n_rows, logits.shape[1],
                                           float(temperature), stats.data_ptr(), 1 if by_source_row else 0,
                                           _ptr(copy_dst), ld_dst, self._stream()), "sq_logits_stats_f16")
        return stats

    def sample_wor(self, logits, rand, row_ids, k, temperature, out, branch=None, out_off=None, out_base=None, stats=None):
        """logits/rand: 2-D fp16 with unit inner stride; row_ids: int32 device tensor or None; out_base: int32 device
        scalar added to every output index; stats: per-part statistics of the source rows (logits_stats(by_source_row))."""
        _need(logits, torch.float16, "logits", contiguous=False); _need(rand, torch.float16, "rand", contiguous=False)
        assert logits.stride(1) == 1 and rand.stride(1) == 1
        _need(out, torch.int64, "out", contiguous=False)
        n_rows = row_ids.shape[0] if row_ids is not None else logits.shape[0]
        if row_ids is not None:
            _need(row_ids, torch.int32, "row_ids")
        if stats is not None:
            _need(stats, torch.float32, "stats")
        ws = self._sample_ws(logits.device, n_rows, logits.shape[1], k)
        check(self.lib.sq_sample_wor_f16(logits.data_ptr(), logits.stride(0), rand.data_ptr(), rand.stride(0),
                                         _ptr(row_ids), n_rows, logits.shape[1], k, float(temperature), out.data_ptr(),
                                         _ptr(branch), _ptr(out_off), _ptr(out_base), _ptr(stats), ws.data_ptr(),
                                         self._stream()), "sq_sample_wor_f16")
        return out

    def sample_wor_f32noise(self, logits, rand32, row_ids, k, temperature, out, branch=None, out_off=None):
        """The acceptance probe's sampler: fp32 noise, fp32 keys (SpecTreeTest, Tree/SpecTree.py:349-360)."""
        _need(logits, torch.float16, "logits", contiguous=False); _need(rand32, torch.float32, "rand32", contiguous=False)
        assert logits.stride(1) == 1 and rand32.stride(1) == 1
        _need(out, torch.int64, "out", contiguous=False)
        n_rows = row_ids.shape[0] if row_ids is not None else logits.shape[0]
        ws = self._sample_ws(logits.device, n_rows, logits.shape[1], k)
        check(self.lib.sq_sample_wor_f32noise_f16(logits.data_ptr(), logits.stride(0), rand32.data_ptr(), rand32.stride(0),
                                                  _ptr(row_ids), n_rows, logits.shape[1], k, float(temperature),
                                                  out.data_ptr(), _ptr(branch), _ptr(out_off), ws.data_ptr(), self._stream()),
              "sq_sample_wor_f32noise_f16")
        return out

    def verify_probe(self, target_logits, draft_logits, tokens, r32, child_off, child_ids, n_tree, gt, temperature, u24,
                     workspace, result):
        """The acceptance probe's verifier: r fp32, p >= r q in fp32 (SpecTreeTest.accept_step, Tree/SpecTree.py:396-417)."""
        _need(target_logits, torch.float16, "target_logits"); _need(draft_logits, torch.float16, "draft_logits")
        _need(tokens, torch.int64, "tokens"); _need(r32, torch.float32, "r32")
        _need(child_off, torch.int32, "child_off"); _need(result, torch.int32, "result")
        vocab = target_logits.shape[-1]
        check(self.lib.sq_verify_probe_f16(target_logits.data_ptr(), draft_logits.data_ptr(), tokens.data_ptr(), tokens.numel(),
                                           r32.data_ptr(), child_off.data_ptr(), _ptr(child_ids), n_tree, vocab, int(gt),
                                           float(temperature), int(u24), workspace.data_ptr(), result.data_ptr(),
                                           self._stream()), "sq_verify_probe_f16")
        return result

    def topk(self, logits, row_ids, k, out, branch=None, out_off=None, out_base=None):
        _need(logits, torch.float16, "logits", contiguous=False)
        assert logits.stride(1) == 1
        _need(out, torch.int64, "out", contiguous=False)
        n_rows = row_ids.shape[0] if row_ids is not None else logits.shape[0]
        if row_ids is not None:
            _need(row_ids, torch.int32, "row_ids")
        ws = self._sample_ws(logits.device, n_rows, logits.shape[1], k)
        check(self.lib.sq_topk_f16(logits.data_ptr(), logits.stride(0), _ptr(row_ids), n_rows, logits.shape[1], k,
                                   out.data_ptr(), _ptr(branch), _ptr(out_off), _ptr(out_base), ws.data_ptr(),
                                   self._stream()), "sq_topk_f16")
        return out

    # ---- a6 / a7 / a8 -----------------------------------------------------------------------
    def verify_workspace(self, n_tree, device):
        nbytes = int(self.lib.sq_verify_workspace_bytes(n_tree))
        return torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=device)

    @staticmethod
    def _check_ring(ring):
        """The result ring is device memory or PINNED host memory (device-accessible: the walker writes the record where
        the host polls it)."""
        if ring.dtype != torch.int32 or not ring.is_contiguous() or not (ring.device.type == "cuda" or ring.is_pinned()):
            raise TypeError("result_ring: contiguous int32 tensor on the device or in pinned host memory")
        assert ring.numel() >= native.SQ_RESULT_RING * native.SQ_RESULT_INTS

    def verify_stochastic(self, target_logits, draft_logits, tokens, r, child_off, child_ids, n_tree, gt, temperature,
                          u24, workspace, result, step=None, bonus_table=None, result_ring=None):
        """step / bonus_table / result_ring: the device-driven form (gt and the bonus uniform read on the device)."""
        _need(target_logits, torch.float16, "target_logits"); _need(draft_logits, torch.float16, "draft_logits")
        _need(tokens, torch.int64, "tokens"); _need(r, torch.float16, "r")
        _need(child_off, torch.int32, "child_off"); _need(result, torch.int32, "result")
        vocab = target_logits.shape[-1]
        assert draft_logits.shape[-1] == vocab and target_logits.shape[0] >= n_tree and draft_logits.shape[0] >= n_tree
        if step is not None:
            _need(step, torch.int32, "step")
        if bonus_table is not None:
            _need(bonus_table, torch.int32, "bonus_table")
        if result_ring is not None:
            self._check_ring(result_ring)
        check(self.lib.sq_verify_stochastic_f16(target_logits.data_ptr(), draft_logits.data_ptr(), tokens.data_ptr(),
                                                tokens.numel(), r.data_ptr(), child_off.data_ptr(), _ptr(child_ids), n_tree,
                                                vocab, int(gt), float(temperature), int(u24), workspace.data_ptr(),
                                                result.data_ptr(), _ptr(step), _ptr(bonus_table),
                                                0 if bonus_table is None else bonus_table.numel(), _ptr(result_ring),
                                                self._stream()), "sq_verify_stochastic_f16")
        return result

    def sample_iid(self, logits, u24, row_ids, k, temperature, out, branch=None, out_off=None):
        """k draws with replacement per row of softmax(logits / T); u24: int32 [n_rows, k] uniforms in [0, 2^24)."""
        _need(logits, torch.float16, "logits", contiguous=False); _need(u24, torch.int32, "u24")
        assert logits.stride(1) == 1
        _need(out, torch.int64, "out", contiguous=False)
        n_rows = row_ids.shape[0] if row_ids is not None else logits.shape[0]
        assert u24.numel() >= n_rows * k
        if row_ids is not None:
            _need(row_ids, torch.int32, "row_ids")
        check(self.lib.sq_sample_iid_f16(logits.data_ptr(), logits.stride(0), _ptr(row_ids), n_rows, logits.shape[1], k,
                                         float(temperature), u24.data_ptr(), out.data_ptr(), _ptr(branch), _ptr(out_off),
                                         self._stream()), "sq_sample_iid_f16")
        return out

    def verify_specinfer(self, target_logits, draft_logits, tokens, r, child_off, child_ids, n_tree, gt, temperature,
                         u24, workspace, result):
        _need(target_logits, torch.float16, "target_logits"); _need(draft_logits, torch.float16, "draft_logits")
        _need(tokens, torch.int64, "tokens"); _need(r, torch.float16, "r")
        _need(child_off, torch.int32, "child_off"); _need(result, torch.int32, "result")
        vocab = target_logits.shape[-1]
        assert draft_logits.shape[-1] == vocab and target_logits.shape[0] >= n_tree and draft_logits.shape[0] >= n_tree
        check(self.lib.sq_verify_specinfer_f16(target_logits.data_ptr(), draft_logits.data_ptr(), tokens.data_ptr(),
                                               tokens.numel(), r.data_ptr(), child_off.data_ptr(), _ptr(child_ids), n_tree, vocab, gt,
                                               float(temperature), int(u24), workspace.data_ptr(), result.data_ptr(),
                                               self._stream()), "sq_verify_specinfer_f16")
        return result

    def verify_tokens(self, target_tokens, tokens, child_off, child_ids, n_tree, gt, workspace, result):
        _need(target_tokens, torch.int64, "target_tokens"); _need(tokens, torch.int64, "tokens")
        _need(child_off, torch.int32, "child_off"); _need(result, torch.int32, "result")
        assert target_tokens.numel() >= n_tree
        check(self.lib.sq_verify_tokens_f16(target_tokens.data_ptr(), tokens.data_ptr(), tokens.numel(), child_off.data_ptr(),
                                            _ptr(child_ids), n_tree, gt, workspace.data_ptr(), result.data_ptr(),
                                            self._stream()), "sq_verify_tokens_f16")
        return result

    def top_p_filter(self, logits, top_p, temperature):
        """In-place nucleus filter on 2-D fp16 logits rows."""
        _need(logits, torch.float16, "logits", contiguous=False)
        assert logits.dim() == 2 and logits.stride(1) == 1
        check(self.lib.sq_top_p_filter_f16(logits.data_ptr(), logits.stride(0), logits.shape[0], logits.shape[1],
                                           float(top_p), float(temperature), self._stream()), "sq_top_p_filter_f16")
        return logits

    def verify_greedy(self, target_logits, tokens, child_off, child_ids, n_tree, gt, workspace, result, step=None,
                      result_ring=None):
        _need(target_logits, torch.float16, "target_logits"); _need(tokens, torch.int64, "tokens")
        _need(child_off, torch.int32, "child_off"); _need(result, torch.int32, "result")
        vocab = target_logits.shape[-1]
        if step is not None:
            _need(step, torch.int32, "step")
        if result_ring is not None:
            self._check_ring(result_ring)
        check(self.lib.sq_verify_greedy_f16(target_logits.data_ptr(), tokens.data_ptr(), tokens.numel(), child_off.data_ptr(),
                                            _ptr(child_ids), n_tree, vocab, int(gt), workspace.data_ptr(), result.data_ptr(),
                                            _ptr(step), _ptr(result_ring), self._stream()), "sq_verify_greedy_f16")
        return result

    # ---- row-wise glue ------------------------------------------------------------------------
    def rmsnorm(self, x, weight, out, eps):
        _need(x, torch.float16, "x"); _need(weight, torch.float16, "weight"); _need(out, torch.float16, "out")
        hidden = x.shape[-1]
        check(self.lib.sq_rmsnorm_f16(x.data_ptr(), weight.data_ptr(), out.data_ptr(), x.numel() // hidden, hidden,
                                      float(eps), self._stream()), "sq_rmsnorm_f16")
        return out

    def add_rmsnorm(self, x, residual, sum_out, weight, out, eps):
        for t, n in ((x, "x"), (residual, "residual"), (sum_out, "sum_out"), (weight, "weight"), (out, "out")):
            _need(t, torch.float16, n)
        hidden = x.shape[-1]
        check(self.lib.sq_add_rmsnorm_f16(x.data_ptr(), residual.data_ptr(), sum_out.data_ptr(), weight.data_ptr(),
                                          out.data_ptr(), x.numel() // hidden, hidden, float(eps), self._stream()),
              "sq_add_rmsnorm_f16")
        return out

    # ---- tall-skinny linear layers (fragment-major operands) -----------------------------------
    @staticmethod
    def frag_shape(rows, cols):
        """Shape of the fragment-major image of a [rows, cols] activation."""
        return (cols // 32, (rows + 15) // 16, 64, 8)

    def repack_weight(self, w):
        """nn.Linear weight [n, k] -> fragment-major image [n/16, k/32, 64, 8] (new tensor)."""
        _need(w, torch.float16, "w")
        n, k = w.shape
        out = torch.empty((n // 16, k // 32, 64, 8), dtype=w.dtype, device=w.device)
        check(self.lib.sq_repack_linear_weight_f16(w.data_ptr(), out.data_ptr(), n, k, self._stream()),
              "sq_repack_linear_weight_f16")
        return out

    def repack_rows(self, x):
        """Row-major activations [m, k] -> fragment-major image [k/32, ceil(m/16), 64, 8]."""
        _need(x, torch.float16, "x", contiguous=False)
        assert x.dim() == 2 and x.stride(1) == 1
        m, k = x.shape
        out = torch.empty(self.frag_shape(m, k), dtype=x.dtype, device=x.device)
        check(self.lib.sq_repack_rows_frag_f16(x.data_ptr(), x.stride(0), out.data_ptr(), m, k, self._stream()),
              "sq_repack_rows_frag_f16")
        return out

    def linear_ts_workspace(self, m, n_out, splits, device):
        nbytes = int(self.lib.sq_linear_ts_workspace_bytes(m, n_out, splits))
        return torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=device)

    def linear_ts(self, a_frag, w_frag, m, n_out, k, out=None, res=None, silu=False, out_frag=False, tiles=256, splits=1,
                  slab=None):
        """out[m, n_out] = a . w^T on fragment-major operands (m <= 128).  splits > 1: fp32 partials go to `slab`
        ([splits, m, n_out]) for add_rmsnorm_slabs; otherwise `out` is row-major [m, n_out] or (out_frag) the
        fragment-major image of the next layer's input."""
        _need(a_frag, torch.float16, "a_frag"); _need(w_frag, torch.float16, "w_frag")
        assert a_frag.numel() >= ((m + 15) // 16) * 16 * k and w_frag.numel() == (2 if silu else 1) * n_out * k
        ldo = 0
        if splits == 1:
            _need(out, torch.float16, "out", contiguous=out_frag)
            ldo = n_out if out_frag else out.stride(0)
            assert out.numel() >= (((m + 15) // 16) * 16 * n_out if out_frag else m * n_out)
        else:
            _need(slab, torch.float32, "slab")
        if res is not None:
            _need(res, torch.float16, "res", contiguous=False)
            assert res.stride(0) == ldo
        check(self.lib.sq_linear_ts_f16(a_frag.data_ptr(), w_frag.data_ptr(), _ptr(res), _ptr(out), ldo,
                                        1 if out_frag else 0, m, n_out, k, 1 if silu else 0, int(tiles), int(splits),
                                        _ptr(slab), slab.numel() * 4 if slab is not None else 0, self._stream()),
              "sq_linear_ts_f16")
        return slab if splits > 1 else out

    def norm_linear(self, x, norm_weight, eps, w_frag, out, m, n_out, k, swiglu=False, tiles=256, ids=None, embed=None,
                    x_out=None):
        """out = epilogue((RMSNorm(x) * norm_weight) . w^T), the norm computed inside the projection (m <= 48, k <= 1024).
        ids / embed: first layer, x = embed[ids] (x_out receives the residual stream)."""
        if not hasattr(self.lib, "sq_norm_linear_f16"):
            raise native.SequoiaNativeError("sq_norm_linear_f16 is an experiment outside the default library: build with "
                                            "SEQUOIA_BUILD_PROBES=1 (csrc/draft_fused.hip)")
        _need(norm_weight, torch.float16, "norm_weight"); _need(w_frag, torch.float16, "w_frag")
        _need(out, torch.float16, "out", contiguous=swiglu)
        if ids is not None:
            _need(ids, torch.int64, "ids"); _need(embed, torch.float16, "embed")
            assert ids.numel() >= m and embed.shape[1] == k
            if x_out is not None:
                _need(x_out, torch.float16, "x_out")
        else:
            _need(x, torch.float16, "x")
            assert x.numel() >= m * k
        ldo = n_out if swiglu else out.stride(0)
        check(self.lib.sq_norm_linear_f16(_ptr(x), _ptr(ids), _ptr(embed), 0 if embed is None else embed.shape[0], _ptr(x_out),
                                          norm_weight.data_ptr(), float(eps), w_frag.data_ptr(), out.data_ptr(), ldo, int(m),
                                          int(n_out), int(k), 1 if swiglu else 0, int(tiles), self._stream()),
              "sq_norm_linear_f16")
        return out

    def draft_attn_block(self, a_frag, wqkv_frag, wo_frag, slab, k_layer, v_layer, cos, sin, position_ids, storage_ids,
                         q_len, n_heads, d, hidden, scale, q_slot0, gt, n_tree, bitmask=None, ctx=None, kv_only=False):
        """The attention half of a small draft's decoder layer in one launch (csrc/draft_block.hip): q|k|v projection,
        RoPE, KV write, tree attention over the cached keys + the row's own key, per-head o_proj partials into
        slab[n_heads][q_len][hidden] (summed by add_rmsnorm_slabs with splits = n_heads).  Rows must not see each other."""
        _need(a_frag, torch.float16, "a_frag"); _need(wqkv_frag, torch.float16, "wqkv_frag")
        _need(k_layer, torch.float16, "k_layer"); _need(v_layer, torch.float16, "v_layer")
        _need(cos, torch.float16, "cos"); _need(sin, torch.float16, "sin")
        _need(position_ids, torch.int64, "position_ids"); _need(storage_ids, torch.int64, "storage_ids")
        assert position_ids.numel() >= q_len and storage_ids.numel() >= q_len
        assert a_frag.numel() >= ((q_len + 15) // 16) * 16 * hidden and wqkv_frag.numel() == 3 * n_heads * d * hidden
        m = k_layer.shape[-2]
        words = 0
        if bitmask is not None:
            _need(bitmask, torch.int64, "bitmask")
            words = bitmask.shape[1]
        slab_bytes = 0
        if not kv_only:
            _need(wo_frag, torch.float16, "wo_frag"); _need(slab, torch.float32, "slab")
            assert wo_frag.numel() == hidden * n_heads * d
            slab_bytes = slab.numel() * 4
        check(self.lib.sq_draft_attn_block_f16(a_frag.data_ptr(), wqkv_frag.data_ptr(), _ptr(wo_frag), _ptr(slab), slab_bytes,
                                               k_layer.data_ptr(), v_layer.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                               position_ids.data_ptr(), storage_ids.data_ptr(), int(q_len), int(n_heads),
                                               int(d), int(hidden), int(m), float(scale), int(q_slot0), int(gt), int(n_tree),
                                               _ptr(bitmask), int(words), _ptr(ctx), 1 if kv_only else 0, self._stream()),
              "sq_draft_attn_block_f16")
        return slab

    def level_attention(self, qkv, out, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads, h_kv, d, scale, q_slot0, gt,
                        n_tree, bitmask=None, ctx=None, out_frag=False, qkv_slab=None):
        """RoPE + KV write + tree attention in one launch for rows that never see each other (csrc/draft_block.hip).
        qkv: [q, (H + 2 H_kv) D] fp16 rows, or qkv_slab = (fp32 slab, splits, q_len, n_cols): split-K partials."""
        _need(out, torch.float16, "out"); _need(k_layer, torch.float16, "k_layer"); _need(v_layer, torch.float16, "v_layer")
        _need(cos, torch.float16, "cos"); _need(sin, torch.float16, "sin")
        _need(position_ids, torch.int64, "position_ids"); _need(storage_ids, torch.int64, "storage_ids")
        m = k_layer.shape[-2]
        words = 0
        if bitmask is not None:
            _need(bitmask, torch.int64, "bitmask")
            words = bitmask.shape[1]
        if qkv_slab is not None:
            slab, splits, q_len, stride = qkv_slab
            _need(slab, torch.float32, "slab")
            assert slab.numel() >= splits * q_len * stride
            src, sl = None, slab.data_ptr()
        else:
            _need(qkv, torch.float16, "qkv", contiguous=False)
            assert qkv.dim() == 2 and qkv.stride(1) == 1
            q_len, stride, splits = qkv.shape[0], qkv.stride(0), 0
            src, sl = qkv.data_ptr(), None
        assert position_ids.numel() >= q_len and storage_ids.numel() >= q_len
        check(self.lib.sq_level_attention_f16(src, sl, int(splits), int(stride), out.data_ptr(), 1 if out_frag else 0,
                                              k_layer.data_ptr(), v_layer.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                              position_ids.data_ptr(), storage_ids.data_ptr(), int(q_len), int(n_heads), int(h_kv),
                                              int(d), int(m), float(scale), int(q_slot0), int(gt), int(n_tree), _ptr(bitmask),
                                              int(words), _ptr(ctx), self._stream()), "sq_level_attention_f16")
        return out

    def add_rmsnorm_slabs(self, slab, splits, residual, sum_out, weight, out, eps, out_frag=False):
        """x = h(sum of the split-K partials); sum_out = x + residual; out = RMSNorm(sum_out) * weight
        (out None: add only)."""
        _need(slab, torch.float32, "slab"); _need(residual, torch.float16, "residual"); _need(sum_out, torch.float16, "sum_out")
        rows, hidden = residual.shape
        if out is not None:
            _need(out, torch.float16, "out"); _need(weight, torch.float16, "weight")
        check(self.lib.sq_add_rmsnorm_slabs_f16(slab.data_ptr(), int(splits), residual.data_ptr(), _ptr(weight),
                                                sum_out.data_ptr(), _ptr(out), 1 if out_frag else 0, rows, hidden,
                                                float(eps), self._stream()), "sq_add_rmsnorm_slabs_f16")
        return out

    def embed_rmsnorm(self, ids, embed, weight, x_out, out, eps, out_frag=False):
        """x_out = embed[ids]; out = RMSNorm(x_out) * weight (row-major or fragment-major), one launch."""
        _need(ids, torch.int64, "ids"); _need(embed, torch.float16, "embed"); _need(weight, torch.float16, "weight")
        _need(x_out, torch.float16, "x_out"); _need(out, torch.float16, "out")
        rows, hidden = x_out.shape
        assert ids.numel() == rows and embed.shape[1] == hidden
        check(self.lib.sq_embed_rmsnorm_f16(ids.data_ptr(), embed.data_ptr(), embed.shape[0], weight.data_ptr(),
                                            x_out.data_ptr(), out.data_ptr(), 1 if out_frag else 0, rows, hidden,
                                            float(eps), self._stream()), "sq_embed_rmsnorm_f16")
        return out

    def embed_stage_rmsnorm(self, stage, embed, weight, x_out, out, eps, out_frag=False):
        """embed_rmsnorm with the forward's inputs staged in the same launch.  stage = (dst_ids, dst_pos, dst_storage, ctx,
        tokens, depth, n_tree, rel_slot0, rel_kv_len, step, advance): the arguments of stage_tree_inputs."""
        dst_ids, dst_pos, dst_storage, ctx, tokens, depth, n_tree, rel_slot0, rel_kv_len, step, advance = stage
        for t, n in ((dst_ids, "dst_ids"), (dst_pos, "dst_pos"), (dst_storage, "dst_storage"), (tokens, "tokens")):
            _need(t, torch.int64, n)
        _need(ctx, torch.int32, "ctx"); _need(depth, torch.int32, "depth"); _need(step, torch.int32, "step")
        _need(embed, torch.float16, "embed"); _need(weight, torch.float16, "weight")
        _need(x_out, torch.float16, "x_out"); _need(out, torch.float16, "out")
        rows, hidden = x_out.shape
        assert dst_ids.numel() == rows and dst_pos.numel() == rows and dst_storage.numel() == rows and depth.numel() >= n_tree
        assert embed.shape[1] == hidden
        check(self.lib.sq_embed_stage_rmsnorm_f16(dst_ids.data_ptr(), dst_pos.data_ptr(), dst_storage.data_ptr(), ctx.data_ptr(),
                                                  tokens.data_ptr(), depth.data_ptr(), int(n_tree), int(rel_slot0),
                                                  int(rel_kv_len), step.data_ptr(), 1 if advance else 0, embed.data_ptr(),
                                                  embed.shape[0], weight.data_ptr(), x_out.data_ptr(), out.data_ptr(),
                                                  1 if out_frag else 0, rows, hidden, float(eps), self._stream()),
              "sq_embed_stage_rmsnorm_f16")
        return out

    def rmsnorm_frag(self, x, weight, out_frag, eps):
        _need(x, torch.float16, "x"); _need(weight, torch.float16, "weight"); _need(out_frag, torch.float16, "out_frag")
        hidden = x.shape[-1]
        check(self.lib.sq_rmsnorm_frag_f16(x.data_ptr(), weight.data_ptr(), out_frag.data_ptr(), x.numel() // hidden, hidden,
                                           float(eps), self._stream()), "sq_rmsnorm_frag_f16")
        return out_frag

    def add_rmsnorm_frag(self, x, residual, sum_out, weight, out_frag, eps):
        for t, n in ((x, "x"), (residual, "residual"), (sum_out, "sum_out"), (weight, "weight"), (out_frag, "out_frag")):
            _need(t, torch.float16, n)
        hidden = x.shape[-1]
        check(self.lib.sq_add_rmsnorm_frag_f16(x.data_ptr(), residual.data_ptr(), sum_out.data_ptr(), weight.data_ptr(),
                                               out_frag.data_ptr(), x.numel() // hidden, hidden, float(eps),
                                               self._stream()), "sq_add_rmsnorm_frag_f16")
        return out_frag

    def silu_mul_frag(self, gate_up, out_frag, rows, inter):
        _need(gate_up, torch.float16, "gate_up"); _need(out_frag, torch.float16, "out_frag")
        check(self.lib.sq_silu_mul_frag_f16(gate_up.data_ptr(), out_frag.data_ptr(), rows, inter, self._stream()),
              "sq_silu_mul_frag_f16")
        return out_frag

    def silu_mul_slabs(self, slab, splits, out, rows, inter, out_frag=False):
        """SwiGLU on the fp32 split-K partials [splits][rows][2 inter] of a gate|up projection."""
        _need(slab, torch.float32, "slab"); _need(out, torch.float16, "out")
        assert slab.numel() >= splits * rows * 2 * inter
        check(self.lib.sq_silu_mul_slabs_f16(slab.data_ptr(), int(splits), out.data_ptr(), 1 if out_frag else 0, int(rows),
                                             int(inter), self._stream()), "sq_silu_mul_slabs_f16")
        return out

    def silu_mul(self, gate_up, out):
        _need(gate_up, torch.float16, "gate_up"); _need(out, torch.float16, "out")
        inter = out.shape[-1]
        check(self.lib.sq_silu_mul_f16(gate_up.data_ptr(), out.data_ptr(), out.numel() // inter, inter, self._stream()),
              "sq_silu_mul_f16")
        return out


_OPS = None


def get_ops():
    global _OPS
    if _OPS is None:
        _OPS = HipOps()
    return _OPS


def set_ops_for_testing(ops):
    """Install a checker implementation (tests only).  Pass None to restore the HIP path."""
    global _OPS
    _OPS = ops
