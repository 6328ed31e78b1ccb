"""CPU stand-in for sequoia_amd.ops.HipOps built on the numpy oracle — TEST INFRASTRUCTURE ONLY.

tests/ install it with sequoia_amd.ops.set_ops_for_testing() to run the *host logic* of the
framework (engines, trees, KV protocol, index algebra, TP sharding) on CPU tensors and compare
it with the reference's traces.  It is never selected by the product code.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops_np as O


def _np(t):
    return t.detach().numpy()


def _succ_from_csr(child_off, child_ids, n):
    off = _np(child_off)
    ids = _np(child_ids) if child_ids is not None else np.zeros(0, np.int32)
    return [[int(c) for c in ids[off[i]:off[i + 1]]] for i in range(n)]


class OracleOps:
    name = "oracle"

    def tree_mask_dense(self, out, q_slot0, gt, n_tree, bitmask):
        bm = _np(bitmask).view(np.uint64) if bitmask is not None else np.ones((1, 1), np.uint64)
        out.copy_(torch.from_numpy(O.tree_mask_dense(q_slot0, out.shape[0], out.shape[1], gt, n_tree, bm)))
        return out

    def kv_scatter(self, k_layer, v_layer, new_k, new_v, storage_ids):
        O.kv_scatter(_np(k_layer), _np(v_layer), _np(new_k), _np(new_v), _np(storage_ids))

    def kv_compact(self, k_cache, v_cache, slots, count, max_count, dst_offset, zero_end, dst_offset_dev=None):
        if dst_offset_dev is not None:
            dst_offset = int(dst_offset_dev.reshape(-1)[0])
        c = max_count if count is None else min(int(count.reshape(-1)[0]), max_count)
        sl = [int(s) for s in _np(slots)[:c]] if c > 0 else []
        O.kv_compact(_np(k_cache)[:, 0], _np(v_cache)[:, 0], sl, dst_offset, zero_end)

    def kv_compact2(self, kv0, kv1, slots, count, max_count, dst_offset, dst_offset_dev=None):
        for kv in (kv0, kv1):
            self.kv_compact(kv.k_cache, kv.v_cache, slots, count, max_count, dst_offset, 0, dst_offset_dev=dst_offset_dev)

    def kv_clear(self, k_cache, v_cache, used_rows):
        O.kv_clear(_np(k_cache)[:, 0], _np(v_cache)[:, 0], used_rows)

    def rope_kv_write(self, qkv, q_out, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads, h_kv, d):
        q = O.rope_kv_write(_np(qkv), n_heads, h_kv, d, _np(cos), _np(sin), _np(position_ids), _np(storage_ids),
                            _np(k_layer), _np(v_layer))
        q_out.copy_(torch.from_numpy(q))

    def rope_kv_write_slabs(self, slab, splits, n_cols, q_out, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads,
                            h_kv, d):
        q_len = position_ids.numel()
        part = _np(slab)[:splits * q_len * n_cols].reshape(splits, q_len, n_cols)
        acc = np.zeros((q_len, n_cols), dtype=np.float32)
        for s_ in range(splits):
            acc = acc + part[s_]
        self.rope_kv_write(torch.from_numpy(O.h(acc)), q_out, k_layer, v_layer, cos, sin, position_ids, storage_ids, n_heads,
                           h_kv, d)

    def store_i32(self, dst, values):
        for i, v in enumerate(values):
            dst[i] = int(v)

    def stage_tree_inputs(self, dst_ids, dst_pos, dst_storage, ctx, tokens, depth, n_tree, rel_slot0, rel_kv_len, step,
                          advance=False):
        """sq_stage_tree_inputs (csrc/kv_ops.hip) restated: the device-driven step's input staging."""
        gt = int(step[1] if advance else step[0])
        q_len = dst_ids.numel()
        slots = np.arange(gt + rel_slot0, gt + rel_slot0 + q_len)
        t = slots - (gt - 1)
        d = _np(depth)
        pos = np.where((t >= 0) & (t < n_tree), d[np.clip(t, 0, n_tree - 1)].astype(np.int64) + gt - 1, slots)
        dst_ids.reshape(-1).copy_(tokens[torch.from_numpy(slots)])
        dst_storage.reshape(-1).copy_(torch.from_numpy(slots))
        dst_pos.reshape(-1).copy_(torch.from_numpy(pos))
        ctx[0], ctx[1], ctx[2] = gt + rel_slot0, gt, gt + rel_kv_len
        if advance:
            step[0] = gt
            step[2] = int(step[2]) + 1

    @staticmethod
    def stats_shape(n_rows, vocab):
        return (n_rows, (vocab + 4095) // 4096, 2)

    def logits_stats(self, logits, temperature, stats, row_ids=None, by_source_row=False, copy_dst=None):
        """The oracle's samplers compute their own softmax statistics: only the row copy is performed."""
        if copy_dst is not None:
            rows = _np(row_ids).astype(np.int64) if row_ids is not None else np.arange(logits.shape[0])
            copy_dst[:len(rows)].copy_(logits[torch.from_numpy(rows)])
        return stats

    def tree_attention(self, q, k_layer, v_layer, out, kv_len, scale, dense_mask=None, q_slot0=0, gt=0, n_tree=0,
                       bitmask=None, ctx=None):
        if ctx is not None:
            q_slot0, gt, kv_len = (int(x) for x in ctx[:3])
        q_len = q.shape[1]
        if dense_mask is not None:
            dm = dense_mask.reshape(dense_mask.shape[-2], dense_mask.shape[-1])
            mask = _np(dm)[:, :kv_len]
        else:
            bm = _np(bitmask).view(np.uint64) if bitmask is not None else np.ones((1, 1), np.uint64)
            mask = O.tree_mask_dense(q_slot0, q_len, kv_len, gt, n_tree, bm)
        out.copy_(torch.from_numpy(O.tree_attention(_np(q), _np(k_layer), _np(v_layer), kv_len, scale, mask)))
        return out

    @staticmethod
    def _emit(samples, out, branch, out_off, out_base=None):
        o = _np(out)
        if out_base is not None:
            o = o[int(out_base.reshape(-1)[0]):]
        if branch is None:
            o[:samples.size] = samples.reshape(-1)
        else:
            br, off = _np(branch), _np(out_off)
            for r in range(samples.shape[0]):
                o[off[r]:off[r] + br[r]] = samples[r, :br[r]]

    def sample_wor(self, logits, rand, row_ids, k, temperature, out, branch=None, out_off=None, out_base=None, stats=None):
        rows = _np(row_ids).astype(np.int64) if row_ids is not None else np.arange(logits.shape[0])
        self._emit(O.sample_wor(_np(logits)[rows], _np(rand)[rows], k, temperature), out, branch, out_off, out_base)
        return out

    def sample_wor_f32noise(self, logits, rand32, row_ids, k, temperature, out, branch=None, out_off=None):
        rows = _np(row_ids).astype(np.int64) if row_ids is not None else np.arange(logits.shape[0])
        self._emit(O.sample_wor_f32noise(_np(logits)[rows], _np(rand32)[rows], k, temperature)[0], out, branch, out_off)
        return out

    def verify_probe(self, target_logits, draft_logits, tokens, r32, child_off, child_ids, n_tree, gt, temperature, u24,
                     workspace, result):
        succ = _succ_from_csr(child_off, child_ids, n_tree)
        self._fill(result, O.verify_probe(_np(target_logits), _np(draft_logits), _np(tokens), _np(r32), succ, gt, temperature,
                                          int(u24) & 0xffffff))
        return result

    def topk(self, logits, row_ids, k, out, branch=None, out_off=None, out_base=None):
        rows = _np(row_ids).astype(np.int64) if row_ids is not None else np.arange(logits.shape[0])
        self._emit(O.topk_ids(_np(logits)[rows], k), out, branch, out_off, out_base)
        return out

    def verify_workspace(self, n_tree, device):
        return torch.zeros(1, dtype=torch.int64, device=device)

    @staticmethod
    def _fill(result, res):
        r = _np(result)
        r[:] = 0
        r[0], r[1], r[2], r[3], r[4], r[5], r[6] = (res["accept_len"], res["n_tree"], res["bonus"], res["terminal"],
                                                    res["reason"], res["gt"], res["last_node"])
        for j, s in enumerate(res["slots"]):
            if j < 56:
                r[8 + j] = s
            r[64 + j] = s

    @staticmethod
    def _step_out(step, result, result_ring, res):
        """What the walker writes for the device-driven step (csrc/verify.hip): next gt, active flag, ring record."""
        if step is None:
            return
        r = _np(result)
        r[7] = int(step[2])
        nxt = res["accept_len"] + (0 if res["terminal"] else 1)
        step[1] = nxt
        if res["terminal"]:
            step[3] = 0
        if result_ring is not None:
            slot = int(step[2]) % 4
            _np(result_ring)[slot * 64:(slot + 1) * 64] = r[:64]

    def _skipped(self, step, result, result_ring):
        """A step in flight behind a terminal one (SQ_STEP_ACTIVE == 0): the walker commits nothing (csrc/verify.hip)."""
        if step is None or int(step[3]) != 0:
            return False
        gt = int(step[0])
        self._fill(result, dict(accept_len=gt, n_tree=0, bonus=-1, terminal=1, reason=4, gt=gt, last_node=0, slots=[]))
        r = _np(result)
        r[7] = int(step[2])
        step[1] = gt
        if result_ring is not None:
            slot = int(step[2]) % 4
            _np(result_ring)[slot * 64:(slot + 1) * 64] = r[:64]
        return True

    def verify_stochastic(self, target_logits, draft_logits, tokens, r, child_off, child_ids, n_tree, gt, temperature,
                          u24, workspace, result, step=None, bonus_table=None, result_ring=None):
        if self._skipped(step, result, result_ring):
            return result
        succ = _succ_from_csr(child_off, child_ids, n_tree)
        uni = int(u24) & 0xffffff
        if step is not None:
            gt = int(step[0])
            if bonus_table is not None:
                uni = int(bonus_table[int(step[2]) % bonus_table.numel()]) & 0xffffff
        res = O.verify_stochastic(_np(target_logits), _np(draft_logits), _np(tokens), _np(r), succ, gt, temperature,
                                  uni, gather_first=bool(int(u24) & 0x80000000))
        self._fill(result, res)
        self._step_out(step, result, result_ring, res)
        return result

    def sample_iid(self, logits, u24, row_ids, k, temperature, out, branch=None, out_off=None):
        rows = _np(row_ids).astype(np.int64) if row_ids is not None else np.arange(logits.shape[0])
        u = _np(u24).reshape(-1)[:len(rows) * k].reshape(len(rows), k) & 0xffffff
        self._emit(O.sample_iid(_np(logits)[rows], u, k, temperature), out, branch, out_off)
        return out

    def verify_specinfer(self, target_logits, draft_logits, tokens, r, child_off, child_ids, n_tree, gt, temperature,
                         u24, workspace, result):
        succ = _succ_from_csr(child_off, child_ids, n_tree)
        res = O.verify_stochastic(_np(target_logits), _np(draft_logits), _np(tokens), _np(r), succ, gt, temperature,
                                  int(u24) & 0xffffff, replace=True, gather_first=bool(int(u24) & 0x80000000))
        self._fill(result, res)
        return result

    def verify_tokens(self, target_tokens, tokens, child_off, child_ids, n_tree, gt, workspace, result):
        succ = _succ_from_csr(child_off, child_ids, n_tree)
        self._fill(result, O.verify_tokens(_np(target_tokens), _np(tokens), succ, gt))
        return result

    def top_p_filter(self, logits, top_p, temperature):
        if top_p < 1.0:
            logits.copy_(torch.from_numpy(O.top_p_filter(_np(logits), top_p, temperature)))
        return logits

    def verify_greedy(self, target_logits, tokens, child_off, child_ids, n_tree, gt, workspace, result, step=None,
                      result_ring=None):
        if self._skipped(step, result, result_ring):
            return result
        succ = _succ_from_csr(child_off, child_ids, n_tree)
        if step is not None:
            gt = int(step[0])
        res = O.verify_greedy(_np(target_logits), _np(tokens), succ, gt)
        self._fill(result, res)
        self._step_out(step, result, result_ring, res)
        return result

    # row-wise glue: the reference's fp16/fp32 expressions (Engine/Llama_modules.py:274-288, 270-271)
    def rmsnorm(self, x, weight, out, eps):
        xf = _np(x).astype(np.float32)
        var = (xf * xf).mean(-1, keepdims=True, dtype=np.float32)
        nrm = O.h(xf * (np.float32(1.0) / np.sqrt(var + np.float32(eps))))
        out.copy_(torch.from_numpy(O.h(O.f(_np(weight)) * O.f(nrm))))
        return out

    def add_rmsnorm(self, x, residual, sum_out, weight, out, eps):
        s = O.h(O.f(_np(x)) + O.f(_np(residual)))
        sum_out.copy_(torch.from_numpy(s))
        return self.rmsnorm(sum_out, weight, out, eps)

    def silu_mul(self, gate_up, out):
        inter = out.shape[-1]
        g = O.f(_np(gate_up)[:, :inter])
        u = O.f(_np(gate_up)[:, inter:])
        s = O.h(g / (np.float32(1.0) + np.exp(-g)))
        out.copy_(torch.from_numpy(O.h(O.f(s) * u)))
        return out
