"""Llama causal LM over the static slot KV cache, without HF model internals.

Two thin classes keep the reference's names and forward signature
(Engine/Llama_model.py:136-300): LlamaForCausalLM_FI (draft flavour: attends over all M slots
through the mask) and LlamaForCausalLM_TG (target flavour: attends over the first kv_len
slots).  With a masked attention kernel the two are the same computation; they differ only in
how the key range is derived and validated.

Weights: a HF checkpoint directory (config.json + *.safetensors) or, because this environment
has no weights, a seeded random init of a named architecture ("random:<arch>[:seed=N]").
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from ..ops import get_ops
from .Llama_modules import LayerWeights, TreeContext, attention_block, mlp_block, rope_tables
from .ts_linear import MAX_ROWS as TS_MAX_ROWS
from .ts_linear import TsLinearSet, forward_ts

# public HF configs of the model families the reference's scripts name (tests/run_A100.sh etc.)
KNOWN_ARCHS = {
    "JackFram/llama-68m": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12,
                               num_key_value_heads=12),
    "JackFram/llama-160m": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                                num_key_value_heads=12),
    "princeton-nlp/Sheared-LLaMA-1.3B": dict(hidden_size=2048, intermediate_size=5504, num_hidden_layers=24,
                                             num_attention_heads=16, num_key_value_heads=16),
    "meta-llama/Llama-2-7b-hf": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                                     num_attention_heads=32, num_key_value_heads=32),
    "meta-llama/Llama-2-13b-hf": dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40,
                                      num_attention_heads=40, num_key_value_heads=40),
    "meta-llama/Llama-2-70b-hf": dict(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80,
                                      num_attention_heads=64, num_key_value_heads=8),
}


@dataclass
class LlamaDims:
    """The subset of LlamaConfig the path needs (attribute names follow HF so KV_Cache and user
    code written against `model.config` keep working)."""
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    max_position_embeddings: int = 2048
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    tp_world: int = 1
    tp_rank: int = 0

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def local_heads(self):
        return self.num_attention_heads // self.tp_world

    @property
    def local_kv_heads(self):
        return max(1, self.num_key_value_heads // self.tp_world)

    @staticmethod
    def from_any(cfg, **over) -> "LlamaDims":
        if isinstance(cfg, LlamaDims):
            d = LlamaDims(**{**cfg.__dict__, **over})
            return d
        keys = LlamaDims.__dataclass_fields__.keys()
        src = cfg if isinstance(cfg, dict) else {k: getattr(cfg, k) for k in keys if hasattr(cfg, k)}
        vals = {k: src[k] for k in keys if k in src and src[k] is not None}
        vals.update(over)
        return LlamaDims(**vals)


class KVConfigView:
    """What KV_Cache reads from a config: per-rank head counts under tensor parallelism."""

    def __init__(self, dims: LlamaDims):
        self.num_hidden_layers = dims.num_hidden_layers
        self.num_key_value_heads = dims.local_kv_heads
        self.num_attention_heads = dims.local_heads
        self.hidden_size = dims.head_dim * dims.local_heads


def _shard_rows(w, rank, world, groups=1):
    """column-parallel: split the output rows of a [out, in] weight."""
    if world == 1:
        return w
    out = w.shape[0]
    assert out % world == 0
    n = out // world
    return w[rank * n:(rank + 1) * n].contiguous()


def _shard_cols(w, rank, world):
    """row-parallel: split the input columns."""
    if world == 1:
        return w
    n = w.shape[1] // world
    return w[:, rank * n:(rank + 1) * n].contiguous()


class LlamaWeights:
    """Fused, device-resident weights of one (shard of a) model."""

    def __init__(self, dims: LlamaDims, dtype, device):
        self.dims, self.dtype, self.device = dims, dtype, device
        self.embed = None
        self.layers: list[LayerWeights] = []
        self.norm = None
        self.lm_head = None

    # ---- from a HF-style state dict / a streamed checkpoint directory (names as in the reference's LlamaForCausalLM_*) ----
    @staticmethod
    def from_state_dict(sd, dims: LlamaDims, dtype, device) -> "LlamaWeights":
        from .checkpoint import StateDictSource
        return LlamaWeights.from_source(StateDictSource(sd), dims, dtype, device)

    @staticmethod
    def from_source(src, dims: LlamaDims, dtype, device) -> "LlamaWeights":
        """Fused, sharded weights from a tensor source (Engine/checkpoint.py).  Every rank asks the source for the slices it
        owns only -- column-parallel projections by output rows, row-parallel ones by input columns, K / V by KV-head
        group, lm_head by vocabulary rows -- and each fused tensor goes to the device before the next one is read."""
        w = LlamaWeights(dims, dtype, device)
        r, ws = dims.tp_rank, dims.tp_world
        put = lambda t: t.to(dtype).to(device).contiguous()

        def rows(name, rank, world):
            n = src.shape(name)[0]
            assert n % world == 0, f"{name}: {n} rows do not split over {world} ranks"
            per = n // world
            return src.rows(name, rank * per, (rank + 1) * per) if world > 1 else src.full(name)

        def cols(name, rank, world):
            n = src.shape(name)[1]
            assert n % world == 0, f"{name}: {n} columns do not split over {world} ranks"
            per = n // world
            return src.cols(name, rank * per, (rank + 1) * per) if world > 1 else src.full(name)

        w.embed = put(src.full("model.embed_tokens.weight"))
        kv_world = min(ws, dims.num_key_value_heads)
        kv_rank = r * kv_world // ws
        for i in range(dims.num_hidden_layers):
            p = f"model.layers.{i}."
            w.layers.append(LayerWeights(
                ln1=put(src.full(p + "input_layernorm.weight")),
                wqkv=put(torch.cat([rows(p + "self_attn.q_proj.weight", r, ws).to(dtype),
                                    rows(p + "self_attn.k_proj.weight", kv_rank, kv_world).to(dtype),
                                    rows(p + "self_attn.v_proj.weight", kv_rank, kv_world).to(dtype)], dim=0)),
                wo=put(cols(p + "self_attn.o_proj.weight", r, ws)),
                ln2=put(src.full(p + "post_attention_layernorm.weight")),
                w_gate_up=put(torch.cat([rows(p + "mlp.gate_proj.weight", r, ws).to(dtype),
                                         rows(p + "mlp.up_proj.weight", r, ws).to(dtype)], dim=0)),
                w_down=put(cols(p + "mlp.down_proj.weight", r, ws))))
        w.norm = put(src.full("model.norm.weight"))
        # tied embeddings (no lm_head.weight in the checkpoint): the head is the embedding matrix
        head = "lm_head.weight" if src.has("lm_head.weight") else "model.embed_tokens.weight"
        w.lm_head = put(rows(head, r, ws))
        return w

    # ---- seeded random init (HF initializer_range = 0.02), generated on the target device --------
    @staticmethod
    def random(dims: LlamaDims, dtype, device, seed: int, logit_gain: float = 1.0) -> "LlamaWeights":
        w = LlamaWeights(dims, dtype, device)
        gen = torch.Generator(device=device)
        ws, r = dims.tp_world, dims.tp_rank

        on_cpu = str(device) == "cpu"

        def normal(shape, tag):
            # one stream per (tensor, rank-independent) so that shards of different ranks are
            # slices of the same full matrix
            gen.manual_seed(seed * 1000003 + tag)
            numel = 1
            for d_ in shape:
                numel *= d_
            if on_cpu and numel > (1 << 22):
                # CPU (baseline timing only): fp16 normal_ is single-threaded and takes minutes for
                # 7B parameters; tile a 4M-element N(0, 0.02) pool instead (memcpy speed)
                pool = torch.empty(1 << 22, dtype=torch.float32).normal_(0.0, 0.02, generator=gen).to(dtype)
                reps = (numel + pool.numel() - 1) // pool.numel()
                return pool.repeat(reps)[:numel].view(shape)
            return torch.empty(shape, dtype=dtype, device=device).normal_(0.0, 0.02, generator=gen)

        h, inter, d = dims.hidden_size, dims.intermediate_size, dims.head_dim
        hq, hkv = dims.num_attention_heads, dims.num_key_value_heads
        w.embed = normal((dims.vocab_size, h), 1)
        ones = lambda: torch.ones(h, dtype=dtype, device=device)
        for i in range(dims.num_hidden_layers):
            t = 100 + i * 10
            if ws == 1:
                wqkv = normal(((hq + 2 * hkv) * d, h), t)
                wo = normal((h, hq * d), t + 1)
                wgu = normal((2 * inter, h), t + 2)
                wd = normal((h, inter), t + 3)
            else:
                full_q = normal((hq * d, h), t); full_k = normal((hkv * d, h), t + 4); full_v = normal((hkv * d, h), t + 5)
                kv_world = min(ws, hkv); kv_rank = r * kv_world // ws
                wqkv = torch.cat([_shard_rows(full_q, r, ws), _shard_rows(full_k, kv_rank, kv_world),
                                  _shard_rows(full_v, kv_rank, kv_world)], dim=0).contiguous()
                del full_q, full_k, full_v
                wo = _shard_cols(normal((h, hq * d), t + 1), r, ws)
                fg = normal((inter, h), t + 2); fu = normal((inter, h), t + 6)
                wgu = torch.cat([_shard_rows(fg, r, ws), _shard_rows(fu, r, ws)], dim=0).contiguous()
                del fg, fu
                wd = _shard_cols(normal((h, inter), t + 3), r, ws)
            w.layers.append(LayerWeights(ln1=ones(), wqkv=wqkv, wo=wo, ln2=ones(), w_gate_up=wgu, w_down=wd))
        w.norm = ones()
        head = normal((dims.vocab_size, h), 2)
        if logit_gain != 1.0:
            head.mul_(logit_gain)
        w.lm_head = _shard_rows(head, r, ws)
        return w


def parse_model_spec(spec):
    """-> (kind, payload).  kind in {'dir', 'random', 'dims', 'state'}."""
    if isinstance(spec, (LlamaDims, dict)) and not (isinstance(spec, dict) and "state_dict" in spec):
        return "dims", spec
    if isinstance(spec, dict):
        return "state", spec
    s = str(spec)
    if s.startswith("random:"):
        parts = s.split(":")
        arch = parts[1]
        opts = dict(p.split("=", 1) for p in parts[2:] if "=" in p)
        return "random", (arch, opts)
    if os.path.isdir(s):
        return "dir", s
    raise FileNotFoundError(
        f"model '{s}' is not a local checkpoint directory; this build has no network access. Use a local HF "
        f"directory or 'random:<arch>[:seed=N]' with <arch> in {sorted(KNOWN_ARCHS)}")


def load_weights(spec, dtype, device, tp_world=1, tp_rank=0, vocab_size=32000):
    if isinstance(spec, dict) and isinstance(spec.get("weights"), LlamaWeights):
        return spec["weights"]                    # prebuilt (sequoia_amd.synthetic)
    if isinstance(spec, LlamaWeights):
        return spec
    kind, payload = parse_model_spec(spec)
    if kind == "random":
        arch, opts = payload
        if arch not in KNOWN_ARCHS:
            raise KeyError(f"unknown architecture {arch}")
        dims = LlamaDims(vocab_size=vocab_size, tp_world=tp_world, tp_rank=tp_rank, **KNOWN_ARCHS[arch])
        return LlamaWeights.random(dims, dtype, device, int(opts.get("seed", 0)), float(opts.get("gain", 1.0)))
    if kind == "dims":
        seed = payload.get("seed", 0) if isinstance(payload, dict) else 0
        src = {k: v for k, v in payload.items() if k != "seed"} if isinstance(payload, dict) else payload
        dims = LlamaDims.from_any(src, tp_world=tp_world, tp_rank=tp_rank)
        return LlamaWeights.random(dims, dtype, device, seed)
    if kind == "state":
        dims = LlamaDims.from_any(payload["config"], tp_world=tp_world, tp_rank=tp_rank)
        return LlamaWeights.from_state_dict(payload["state_dict"], dims, dtype, device)
    # HF directory: config.json + *.safetensors, streamed slice by slice (every rank reads its own shard only)
    from .checkpoint import CheckpointDirSource, read_config
    dims = LlamaDims.from_any(read_config(payload), tp_world=tp_world, tp_rank=tp_rank)
    src = CheckpointDirSource(payload)
    try:
        w = LlamaWeights.from_source(src, dims, dtype, device)
        w.checkpoint_bytes_read = src.bytes_read
    finally:
        src.close()
    return w


class _LlamaForCausalLM:
    """Shared forward.  `key_range` distinguishes the FI / TG flavours."""
    key_range = "kv_len"

    def __init__(self, weights: LlamaWeights, reduce_fn=None, gather_logits_fn=None):
        self.weights = weights
        self.dims = weights.dims
        self.config = weights.dims
        self.vocab_size = weights.dims.vocab_size
        self.dtype, self.device = weights.dtype, weights.device
        self.cos, self.sin = rope_tables(self.dims.head_dim, self.dims.max_position_embeddings, self.dims.rope_theta,
                                         self.device, self.dtype)
        # tall-skinny projections for tree forwards (<= 128 rows): fragment-major weight stream, Engine/ts_linear.py
        self.ts = TsLinearSet(weights, self.dims) if TsLinearSet.supported(weights, self.dims) else None
        # One copy of the layer projections (fragment-major only, prefill in 128-row chunks on the same kernel) or two (the
        # row-major nn.Linear layout next to it for hipBLASLt prefill)?  SEQUOIA_TS_EXCLUSIVE = 1 / 0 decides; unset ("auto")
        # the second copy is dropped when the projections exceed a quarter of the device's memory (70B on one GPU: 138 of
        # 288 GB) -- judged on TOTAL memory, so every rank of a tensor-parallel job decides alike.
        if self.ts is not None:
            mode = os.environ.get("SEQUOIA_TS_EXCLUSIVE", "auto")
            if mode == "1" or (mode == "auto" and self.ts.layer_weight_bytes()
                               > 0.25 * torch.cuda.get_device_properties(self.device).total_memory):
                self.ts.make_exclusive()
        self.reduce_fn = reduce_fn                  # TP all-reduce hook (None on one GPU)
        self.reduce_slabs_fn = None                 # TP: all-reduce straight from split-K partials (xGMI kernel), optional
        self.reduce_norm_fn = None                  # TP: (partial, splits, x, weight, out, eps, out_frag) all-reduce + skip + RMSNorm
        self.gather_logits_fn = gather_logits_fn    # TP vocab all-gather hook

    def eval(self):
        return self

    def ensure_rope(self, max_length: int):
        """cos/sin tables cover max(max_position_embeddings, max_length) positions: the RoPE kernels index them by
        position id unchecked, and an engine may be built with max_length beyond the config's table size."""
        if max_length > self.cos.shape[0]:
            self.cos, self.sin = rope_tables(self.dims.head_dim, max_length, self.dims.rope_theta, self.device,
                                             self.dtype)

    @torch.no_grad()
    def forward(self, input_ids, max_length, storage_ids, attention_mask=None, position_ids=None, kv_cache=None,
                debug=False, tree: TreeContext | None = None):
        ops = get_ops()
        W, dims = self.weights, self.dims
        if input_ids.dim() != 2 or input_ids.shape[0] != 1:
            raise ValueError("batch size must be 1 (Engine/Llama_KV.py:8)")
        q_len = input_ids.shape[1]
        pos = position_ids.reshape(-1)
        if pos.shape[0] != q_len or storage_ids.shape[0] != q_len:
            raise ValueError("position_ids / storage_ids must have one entry per input token")
        dense = None
        if tree is None:
            if attention_mask is None:
                raise ValueError("attention_mask is required when no TreeContext is given")
            dense = attention_mask.reshape(attention_mask.shape[-2], attention_mask.shape[-1])
            kv_len = kv_cache.kv_offset + q_len
            if self.key_range == "kv_len":
                # LlamaAttention_TG validates the mask shape (Engine/Llama_modules.py:238-242)
                if tuple(attention_mask.shape[-2:]) != (q_len, kv_len):
                    raise ValueError(f"Attention mask should be of size {(1, 1, q_len, kv_len)}, but is "
                                     f"{tuple(attention_mask.size())}")
            else:
                if attention_mask.shape[-1] != max_length or attention_mask.shape[-2] != q_len:
                    raise ValueError(f"Attention mask should be of size {(1, 1, q_len, max_length)}, but is "
                                     f"{tuple(attention_mask.size())}")
            if dense.dtype != self.dtype:
                dense = dense.to(self.dtype)
        # tree forwards (and tensor-parallel shards: the hooks are applied inside forward_ts) on the tall-skinny path
        if self.ts is not None and q_len <= TS_MAX_ROWS:
            return forward_ts(self, self.ts, input_ids[0], q_len, pos, storage_ids, dense, tree, kv_cache)
        if tree is not None and tree.stage is not None:
            ops.stage_tree_inputs(*tree.stage)         # device-driven step on the general path: staging is its own launch
        if self.ts is not None and self.ts.exclusive:
            return self._forward_chunked(input_ids, q_len, pos, storage_ids, dense, tree, kv_cache)
        x = F.embedding(input_ids[0], W.embed)                      # [q, hidden]
        hbuf = torch.empty_like(x)
        pending = None                                               # branch output not yet added to x
        for li, lw in enumerate(W.layers):
            if pending is None:
                ops.rmsnorm(x, lw.ln1, hbuf, dims.rms_norm_eps)
            else:
                ops.add_rmsnorm(pending, x, x, lw.ln1, hbuf, dims.rms_norm_eps)
            attn = attention_block(hbuf, lw, li, dims, kv_cache, self.cos, self.sin, pos, storage_ids, dense, tree,
                                   self.reduce_fn)
            ops.add_rmsnorm(attn, x, x, lw.ln2, hbuf, dims.rms_norm_eps)
            pending = mlp_block(hbuf, lw, dims, self.reduce_fn)
        if pending is None:
            ops.rmsnorm(x, W.norm, hbuf, dims.rms_norm_eps)
        else:
            ops.add_rmsnorm(pending, x, x, W.norm, hbuf, dims.rms_norm_eps)
        kv_cache.note_written(q_len)
        logits = F.linear(hbuf, W.lm_head)
        if self.gather_logits_fn is not None:
            logits = self.gather_logits_fn(logits)
        return logits.unsqueeze(0)

    def _forward_chunked(self, input_ids, q_len, pos, storage_ids, dense, tree, kv_cache):
        """More than MAX_ROWS new tokens when the fragment-major images are the only copy of the weights (exclusive
        mode): the rows run as consecutive chunks of <= MAX_ROWS.  A chunk's queries see the earlier chunks through the
        KV cache (causal prefix / tree mask by slot: a tree node's ancestors have smaller ids, i.e. sit in the same or an
        earlier chunk), so the result equals the one-pass forward up to accumulation order.  Legal inside a captured step
        (verify forwards of the 193- / 256- / 512-node growmaps on a target in exclusive mode)."""
        assert tree is None or tree.need_logits, "the chunked forward returns logits: need_logits = False is not supported here"
        outs = []
        for r0 in range(0, q_len, TS_MAX_ROWS):
            r1 = min(q_len, r0 + TS_MAX_ROWS)
            sub_tree, sub_dense = None, None
            if tree is not None:
                # a captured forward reads {q_slot0, gt, kv_len} from its device block: the chunk's block is that one shifted
                # by the chunk's rows (a device-side add: replayable), its host scalars likewise
                sub_ctx = None
                if tree.ctx is not None:
                    key = (r0, (r1 - q_len) if tree.contiguous_slots else 0, str(tree.ctx.device))
                    cache = self.__dict__.setdefault("_chunk_shift", {})
                    shift = cache.get(key)
                    if shift is None:      # made once, outside any capture (the eager warm-up of a graph comes first)
                        if torch.cuda.is_current_stream_capturing():
                            raise RuntimeError("chunked forward: run the forward eagerly once before capturing it")
                        shift = cache[key] = torch.tensor([key[0], 0, key[1]], dtype=torch.int32, device=tree.ctx.device)
                    sub_ctx = tree.ctx + shift
                sub_tree = TreeContext(q_slot0=tree.q_slot0 + r0, gt=tree.gt, n_tree=tree.n_tree, bitmask=tree.bitmask,
                                       kv_len=tree.q_slot0 + r1 if tree.contiguous_slots else tree.kv_len, ctx=sub_ctx,
                                       contiguous_slots=tree.contiguous_slots)
            else:
                sub_dense = dense[r0:r1]
            outs.append(forward_ts(self, self.ts, input_ids[0, r0:r1], r1 - r0, pos[r0:r1], storage_ids[r0:r1], sub_dense,
                                   sub_tree, kv_cache)[0])
        return torch.cat(outs, dim=0).unsqueeze(0)

    __call__ = forward


class LlamaForCausalLM_FI(_LlamaForCausalLM):
    """Draft flavour (Engine/Llama_model.py:136-216): the mask spans all M slots."""
    key_range = "max_length"


class LlamaForCausalLM_TG(_LlamaForCausalLM):
    """Target flavour (Engine/Llama_model.py:219-300): keys are the first kv_len slots."""
    key_range = "kv_len"
