"""Tall-skinny linear layers for tree forwards (q_len <= 144): weight images, launch plans, forward.

The dense projections of a tree forward multiply <= 128 activation rows by every weight of the model — an
HBM stream of the weights.  `sq_linear_ts_f16` (csrc/ts_linear.hip) runs them from fragment-major operand
images; this module owns what surrounds the kernel on the host:

  * the fragment-major copies of a model's projection weights (made once, on first use);
  * a launch plan per (projection shape, row-tile count): (tiles, splits) of the kernel, or "torch" when
    PyTorch's hipBLASLt GEMM is the faster one for that shape (the 128-row gate_up / lm_head shapes).  Plans for
    the shapes of the bundled configurations are shipped (ts_plans_gfx950.json, measured on MI355X); unknown
    shapes are timed once, outside any graph capture, against torch on the same weights;
  * the decoder forward that threads fragment-major activations between the producers (RMSNorm, attention,
    SwiGLU epilogue) and the projections.  Rounding points are those of the general path
    (Llama_model._LlamaForCausalLM.forward); only the fp32 summation order inside a projection differs.

Reference lines replaced: LlamaAttention_FI/TG q/k/v/o projections (Engine/Llama_modules.py:104-112,138,199-207,256),
LlamaMLP_FI (:262-271), the decoder layer's residual adds and norms (:282-288,341-346), lm_head
(Engine/Llama_model.py:280-283).
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn.functional as F

from ..ops import get_ops

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAN_FILE = os.path.join(_PKG, "ts_plans_gfx950.json")
MAX_ROWS = 144
# the 129-row build (8 MFMA row tiles + the extra row on the vector ALU, csrc/ts_linear.hip) is what allows 4 gate+up units per
# workgroup at 129 rows; SEQUOIA_TS_TAIL=0 (read by the library too) rounds 129 rows up to 9 row tiles, which take <= 3 units
TAIL_ROWS = 129 if (os.environ.get("SEQUOIA_TS_TAIL") or "1")[0] != "0" else 128
ENABLED = os.environ.get("SEQUOIA_TS_LINEAR", "1") != "0"
# Tensor-parallel jobs replicate the draft model, the samplers and the verifier on every rank and rely on bit-identical
# results across ranks (no broadcast of decisions).  Launch plans picked by per-rank timing would break that (a
# different split order changes the last bit of a logit), so with this flag a shape without a shipped plan takes the
# deterministic default plan (lm_head: the PyTorch GEMM) instead of being timed.  Set by harness.build for TP configs.
DETERMINISTIC_PLANS = os.environ.get("SEQUOIA_TS_DETERMINISTIC", "0") == "1"

_SHIPPED = None


def shipped_plans() -> dict:
    global _SHIPPED
    if _SHIPPED is None:
        try:
            with open(PLAN_FILE) as f:
                _SHIPPED = json.load(f)["plans"]
        except (OSError, ValueError, KeyError):
            _SHIPPED = {}
    return _SHIPPED


def plan_key(n_out: int, k: int, silu: bool, mtp: int) -> str:
    return f"{n_out}x{k}{'s' if silu else ''}@{mtp}"


SPLITTABLE = ("qkv", "o", "down", "gate_up")   # projections whose consumer reads split-K slabs: RoPE + KV write (qkv), the
#   residual add + RMSNorm (o, down), the SwiGLU pass (gate_up: run as a plain [2 inter] x k layer, sq_silu_mul_slabs_f16)
MAX_SPLITS = 8
# K-splits a launch plan may ask for, per projection: the SwiGLU layer's split candidates stop at 4 (`candidates`), so its
# [2 inter]-wide partials never need more than 4 slabs -- the persistent slab buffer is sized by what can be used, not by
# MAX_SPLITS x the widest projection (70B widths: 132 MB instead of 264 MB per model)
SPLITS_CAP = {"qkv": MAX_SPLITS, "o": MAX_SPLITS, "down": MAX_SPLITS, "gate_up": 4}


def candidates(n_out: int, k: int, silu: bool, m: int, allow_split: bool = False):
    """(tiles, splits) launch shapes worth timing for one projection.  SwiGLU layers: splits == 1 is the fused epilogue
    (tiles over the n_out gate+up units, <= 4 each); splits > 1 runs the layer as a plain [2 n_out] x k projection
    (tiles over 2 n_out / 16 column units) followed by sq_silu_mul_slabs_f16."""
    ksteps = k // 32
    out = []

    def add(units, max_u, split_opts):
        tiles_opts = {(units + u - 1) // u for u in range(1, max_u + 1)}
        tiles_opts |= {t for t in (256, 512) if t <= units and (units + t - 1) // t <= max_u}
        for tiles in sorted(tiles_opts):
            for splits in split_opts:
                if ksteps < splits * 8 or not 48 <= tiles * splits <= 2100:
                    continue
                out.append((tiles, splits))
    wide = 6 if m > 64 else 4
    if silu:
        add(n_out // 16, 4 if m <= TAIL_ROWS else 3, (1,))   # fused epilogue: <= 4 gate+up units (8 MFMA column tiles; 3 beyond 129 rows:
        #                                                      129 = 8 MFMA row tiles + the extra row on the vector ALU, csrc/ts_linear.hip)
        if allow_split:
            add(2 * n_out // 16, wide, (2, 3, 4))
    else:
        add(n_out // 16, wide, (1, 2, 3, 4, 6, 8) if allow_split else (1,))
        if m <= 128 and n_out >= 8192:                         # 8-tile workgroups for the wide projections (qkv of the 13B / 70B widths)
            add(n_out // 16, 8, (1, 2, 3, 4) if allow_split else (1,))
    return sorted(set(out))


class TsLinearSet:
    """Fragment-major weights + launch plans of one model."""
    NAMES = ("qkv", "o", "gate_up", "down", "lm_head")

    def __init__(self, weights, dims):
        self.W, self.dims = weights, dims
        self.device = torch.device(weights.device)
        self._frag: dict = {}
        self._plans: dict = {}          # q_len -> {name: None | (tiles, splits)}
        self.tuned: dict = {}           # plan_key -> record of a timing run (for export)
        d = dims
        hd = d.local_heads * d.head_dim
        self.shapes = {                  # name -> (n_out, k, silu)
            "qkv": ((d.local_heads + 2 * d.local_kv_heads) * d.head_dim, d.hidden_size, False),
            "o": (d.hidden_size, hd, False),
            "gate_up": (weights.layers[0].w_down.shape[1], d.hidden_size, True),
            "down": (d.hidden_size, weights.layers[0].w_down.shape[1], False),
            "lm_head": (weights.lm_head.shape[0], d.hidden_size, False),
        }
        # exclusive: the fragment-major images are the ONLY copy of the layer projections (the row-major nn.Linear
        # layout is released after the repack: 70B on few GPUs); forwards of more than MAX_ROWS rows then run as row chunks
        self.exclusive = False
        self._zero_rows = torch.zeros((MAX_ROWS, d.hidden_size), dtype=torch.float16, device=self.device)
        # split-K partials [splits][rows][n_out] fp32: one buffer for the model's life (captured graphs hold it)
        self._slab = torch.empty(MAX_ROWS * max(SPLITS_CAP[n] * self.shapes[n][0] * (2 if self.shapes[n][2] else 1) for n in SPLITTABLE),
                                 dtype=torch.float32, device=self.device)

    @staticmethod
    def supported(weights, dims, reduce_fn=None) -> bool:
        """Tensor-parallel shards qualify too: the all-reduce of a row-parallel projection is applied between the
        projection and the residual add (forward_ts)."""
        if not ENABLED or torch.device(weights.device).type != "cuda" or weights.dtype != torch.float16:
            return False
        inter = weights.layers[0].w_down.shape[1]
        hd = dims.local_heads * dims.head_dim
        qkv = (dims.local_heads + 2 * dims.local_kv_heads) * dims.head_dim
        return (dims.hidden_size % 32 == 0 and inter % 32 == 0 and hd % 32 == 0 and qkv % 16 == 0
                and weights.lm_head.shape[0] % 16 == 0 and 2 * inter * dims.hidden_size * 2 < (1 << 32)
                and weights.lm_head.shape[0] * dims.hidden_size * 2 < (1 << 32))

    # ---- weights ------------------------------------------------------------------------------------------
    def _row_major(self, name, li):
        if name == "lm_head":
            return self.W.lm_head
        lw = self.W.layers[li]
        w = {"qkv": lw.wqkv, "o": lw.wo, "gate_up": lw.w_gate_up, "down": lw.w_down}[name]
        if w is None:
            raise RuntimeError(f"row-major {name} weights were released (exclusive tall-skinny mode)")
        return w

    def frag(self, name, li=0):
        key = (name, 0 if name == "lm_head" else li)
        t = self._frag.get(key)
        if t is None:
            t = get_ops().repack_weight(self._row_major(name, li))
            self._frag[key] = t
        return t

    def layer_weight_bytes(self) -> int:
        """Bytes of the layer projections (one copy): what a second, row-major copy would cost."""
        per_layer = sum((2 if self.shapes[n][2] else 1) * self.shapes[n][0] * self.shapes[n][1] * 2
                        for n in ("qkv", "o", "gate_up", "down"))
        return per_layer * len(self.W.layers)

    def make_exclusive(self):
        """Repack every layer projection now and release the row-major copies (layer by layer: peak = one extra layer)."""
        if self.exclusive:
            return
        for li, lw in enumerate(self.W.layers):
            for name, attr in (("qkv", "wqkv"), ("o", "wo"), ("gate_up", "w_gate_up"), ("down", "w_down")):
                self.frag(name, li)
                setattr(lw, attr, None)
        self.exclusive = True
        self._plans.clear()
        torch.cuda.empty_cache()

    def default_plan(self, name, q_len):
        """A launch shape that always works (exclusive mode cannot fall back to PyTorch's GEMM) for a shape nobody measured.
        SwiGLU layers: one resident wave of workgroups, the widest column tiles the row count allows.  The other layer
        projections follow what every measured plan of the 7B / 13B / 70B widths looks like (ts_plans_gfx950.json; round 5
        found the old "256 workgroups x all of K" default 40-60 % slower than the tuned plans on the full-width 70B shapes,
        where every workgroup then pulls the whole 2-7 MB activation image through its L2): about 4 column units per
        workgroup and as many K-splits as make ~256 workgroups -- each workgroup reads 1 / splits of the activation image."""
        n_out, k, silu = self.shapes[name]
        units = n_out // 16
        if silu or name not in SPLITTABLE:
            max_u = 3 if silu else (6 if q_len > 64 else 4)
            return (max((units + max_u - 1) // max_u, min(units, 256)), 1)
        if units < 256:                          # narrow layers (small drafts): one column unit group per workgroup, K split to ~192
            max_u = 6 if q_len > 64 else 4
            tiles = max((units + max_u - 1) // max_u, min(units, 256))
            splits = 1
            while tiles * splits < 192 and splits < MAX_SPLITS and (k // 32) >= (splits + 1) * 8:
                splits += 1
            return (tiles, splits)
        tiles = (units + 3) // 4
        splits = max(1, min(SPLITS_CAP[name], (256 + tiles // 2) // tiles))
        while splits > 1 and (k // 32) < splits * 8:
            splits -= 1
        return (tiles, splits)

    def _images_fit(self, name) -> bool:
        """A second copy of this projection's weights must leave the device comfortable (70B on one GPU does not)."""
        n_out, k, silu = self.shapes[name]
        n_layers = 1 if name == "lm_head" else len(self.W.layers)
        need = n_layers * (2 if silu else 1) * n_out * k * 2
        free, _ = torch.cuda.mem_get_info(self.device)
        return need < 0.5 * free

    # ---- plans --------------------------------------------------------------------------------------------
    def plan(self, q_len: int) -> dict:
        p = self._plans.get(q_len)
        if p is None:
            p = {}
            mtp = (q_len + 15) // 16
            capturing = torch.cuda.is_current_stream_capturing()
            for name in self.NAMES:
                n_out, k, silu = self.shapes[name]
                rec = shipped_plans().get(plan_key(n_out, k, silu, mtp))
                have_images = (name, 0) in self._frag
                if self.exclusive and name != "lm_head":
                    if rec is None or rec == "torch":
                        rec = self.default_plan(name, q_len)
                elif capturing and (rec is None or not have_images):
                    if DETERMINISTIC_PLANS:
                        # replicated ranks must run identical arithmetic: a fallback that depends on this rank's warm-up
                        # history is an error, not a choice
                        raise RuntimeError(f"{name}@{q_len}: no launch plan / weight image before graph capture "
                                           "(deterministic plans: run the forward eagerly once before capturing it)")
                    rec = "torch"          # no timing, no allocation, no repack inside a capture (the eager warm-ups
                    #                        of a graph runner come first and cache the real plan)
                elif rec != "torch" and not have_images and not self._images_fit(name):
                    if DETERMINISTIC_PLANS:
                        raise RuntimeError(f"{name}: the fragment-major weight image does not fit this rank's free memory "
                                           "(deterministic plans: a per-rank fallback would make the ranks' results differ; "
                                           "use SEQUOIA_TS_EXCLUSIVE=1)")
                    rec = "torch"
                elif rec is None and DETERMINISTIC_PLANS:
                    rec = "torch" if name == "lm_head" else self.default_plan(name, q_len)
                elif rec is None:
                    rec = self.autotune(name, q_len)
                if rec != "torch" and int(rec[1]) > SPLITS_CAP.get(name, 1):
                    raise ValueError(f"launch plan {plan_key(n_out, k, silu, mtp)} = {rec}: {name} takes at most "
                                     f"{SPLITS_CAP.get(name, 1)} K-splits (the split-K slab buffer is sized for that)")
                if rec != "torch":
                    # a plan is keyed by ROW TILES (ceil(q / 16)), the kernel's register budget by rows: 129 rows run the
                    # 8-tile + extra-row build (<= 8 SwiGLU column tiles, <= 6 plain), 130-144 rows the 9-tile build (<= 6 / <= 6)
                    # -- a plan measured at 129 rows must still launch for a 144-row chunk of a prefill
                    units = n_out // 16
                    max_u = (4 if q_len <= TAIL_ROWS else 3) if silu else (8 if q_len <= 128 else 6)
                    if int(rec[1]) == 1 and -(-units // int(rec[0])) > max_u:
                        rec = (-(-units // max_u), 1)
                p[name] = None if rec == "torch" else (int(rec[0]), int(rec[1]))
            if not capturing:
                for name, v in p.items():                   # materialise the weight images now, outside any capture
                    if v is not None:
                        for li in range(1 if name == "lm_head" else len(self.W.layers)):
                            self.frag(name, li)
                self._plans[q_len] = p
        return p

    @torch.inference_mode()
    def autotune(self, name: str, q_len: int, reps: int = 16):
        """Time the kernel's launch shapes and torch's GEMM for one projection at q_len rows, rotating over the
        layers' weights (32 x 100+ MB: nothing stays in the 256 MiB Infinity Cache).  Returns "torch" or
        [tiles, splits]; the record lands in self.tuned."""
        ops = get_ops()
        n_out, k, silu = self.shapes[name]
        n_layers = 1 if name == "lm_head" else len(self.W.layers)
        # images that exist BEFORE this call belong to other plans -- and, through them, to hipGraphs that were captured
        # with their addresses: they must outlive this call whatever it decides (see the end of the function)
        had_images = {li for li in range(n_layers) if (name, li) in self._frag}
        dev = self.device
        x = (torch.randn(q_len, k, device=dev) * 0.5).half()
        xf = ops.repack_rows(x)
        out = torch.empty((q_len, n_out), dtype=torch.float16, device=dev)
        act = torch.empty((q_len, n_out), dtype=torch.float16, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timeit(fn):
            """GPU time per call: `reps` calls captured into a hipGraph and replayed (an eager launch costs the host
            7-20 us, more than the small projections take on the GPU)."""
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for i in range(2):
                    fn(i % n_layers)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for i in range(reps):
                        fn(i % n_layers)
                g.replay()
                e0.record(side)
                g.replay()
                e1.record(side)
                e1.synchronize()
            torch.cuda.current_stream(dev).wait_stream(side)
            return e0.elapsed_time(e1) * 1e3 / reps

        def torch_fn(li):
            y = F.linear(x, self._row_major(name, li))
            if silu:
                ops.silu_mul(y, act)
        t_torch = timeit(torch_fn)
        best, t_best = "torch", t_torch
        best_ts, t_ts = None, float("inf")
        results = {}
        for tiles, splits in candidates(n_out, k, silu, q_len, allow_split=name in SPLITTABLE):
            slab = self._slab if splits > 1 else None

            def ts_fn(li, tiles=tiles, splits=splits, slab=slab):
                if silu and splits > 1:          # plain [2 n_out] x k layer + the SwiGLU pass over its partials
                    ops.linear_ts(xf, self.frag(name, li), q_len, 2 * n_out, k, tiles=tiles, splits=splits, slab=slab)
                    ops.silu_mul_slabs(slab, splits, act, q_len, n_out)
                else:
                    ops.linear_ts(xf, self.frag(name, li), q_len, n_out, k, out=out, silu=silu, tiles=tiles, splits=splits,
                                  slab=slab)
            for li in range(n_layers):                       # weight images exist before anything is captured
                self.frag(name, li)
            t = timeit(ts_fn)
            results[f"{tiles}x{splits}"] = round(t, 2)
            if t < t_ts:
                best_ts, t_ts = [tiles, splits], t
        # lm_head: PyTorch's GEMM on a tie (no second 0.26 GB image).  Layer projections: the kernel on a tie -- one code
        # path and one summation order for the whole decoder layer; the second weight image is nothing against 288 GB.
        margin = 0.97 if name == "lm_head" else 1.03
        if best_ts is not None and t_ts < t_torch * margin:
            best, t_best = best_ts, t_ts
        key = plan_key(n_out, k, silu, (q_len + 15) // 16)
        self.tuned[key] = dict(choice=best, us=round(t_best, 2), torch_us=round(t_torch, 2), q_len=q_len, ts_us=results)
        if best == "torch":
            # Drop the images this call made for the measurement -- and ONLY those.  Round 4 found the use-after-free the
            # unconditional pop was: a forward at a new row count (the 129-row prefill of a 2-node tree in
            # growmap_tuning --config D) autotunes, PyTorch's GEMM wins, the pop frees the images of ALL row counts, and the
            # hipGraph captured earlier for another row count replays on freed memory (GPU memory access fault).
            for li in range(n_layers):
                if li not in had_images:
                    self._frag.pop((name, li), None)
        return best


def plan_signature(*models) -> str:
    """Hash of every launch plan the given models have chosen so far (per q_len, per projection)."""
    import hashlib
    rows = []
    for i, m in enumerate(models):
        ts = getattr(m, "ts", None)
        if ts is None:
            rows.append((i, "no-ts"))
            continue
        for q in sorted(ts._plans):
            rows.append((i, q, sorted((k, v) for k, v in ts._plans[q].items())))
    return hashlib.sha256(repr(rows).encode()).hexdigest()[:16]


def assert_same_plans_across_ranks(*models, group=None):
    """Tensor-parallel jobs: the replicated draft / sampler / verifier rely on bit-identical arithmetic on every rank, so
    every rank must have picked the same launch plans.  Call after the warm-up steps (collective; not inside a capture)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    mine = plan_signature(*models)
    sigs = [None] * dist.get_world_size(group)
    dist.all_gather_object(sigs, mine, group=group)
    if len(set(sigs)) != 1:
        raise RuntimeError(f"launch plans differ between tensor-parallel ranks: {sigs}")


# Small draft models (hidden <= 1024, <= 48 rows: every level of the 68m / 160m drafts): the RMSNorm in front of qkv,
# gate_up and lm_head can be computed inside the projection (csrc/draft_fused.hip) and o_proj / down_proj can write the
# residual stream through the "+ residual" epilogue -- 14 launches per 2-layer forward instead of 19.  Measured on MI355X
# (profiles/r03_draft_fused_not_adopted.md): NOT faster -- every kernel of the unfused sequence already sits at the
# ~4.7 us per-kernel floor of a dependent graph node, and a fused kernel's body is the SUM of the two latency chains it
# replaces (norm 1 load + reduce, then weights + MFMA + merge), so 5 fewer launches buy back what the longer bodies cost
# (34-row level: 96 vs 88 us).  Kept as an opt-in (SEQUOIA_DRAFT_FUSED=1), covered by tests/test_draft_fused_gpu.py.
SMALL_FUSED = os.environ.get("SEQUOIA_DRAFT_FUSED", "0") == "1"
SMALL_MAX_ROWS, SMALL_MAX_HIDDEN = 48, 1024


def small_fused_ok(model, ts: "TsLinearSet", q_len: int) -> bool:
    d = model.dims
    return (SMALL_FUSED and hasattr(get_ops().lib, "sq_norm_linear_f16") and q_len <= SMALL_MAX_ROWS and d.hidden_size <= SMALL_MAX_HIDDEN and model.reduce_fn is None
            and model.gather_logits_fn is None and not ts.exclusive and ts.shapes["down"][1] % 32 == 0
            and d.tp_world == 1)


def forward_small_fused(model, ts: "TsLinearSet", ids, q_len, pos, storage_ids, dense, tree, kv_cache, logits_out=None):
    """forward_ts for a small draft: per layer  norm+qkv | RoPE + KV write | tree attention | o_proj + residual |
    norm+gate_up+SwiGLU | down_proj + residual,  then norm+lm_head (optionally straight into `logits_out`, the tree's
    draft_logits rows).  Same rounding points as forward_ts; fp32 summation orders differ (K is not split across
    workgroups here)."""
    from .Llama_modules import attention_core
    ops = get_ops()
    W, dims = model.weights, model.dims
    eps = dims.rms_norm_eps
    dev, dt = W.embed.device, W.embed.dtype
    hidden = dims.hidden_size
    n_qkv = ts.shapes["qkv"][0]
    hd = ts.shapes["o"][1]
    inter = ts.shapes["down"][1]
    vocab = ts.shapes["lm_head"][0]
    ids = ids.contiguous()
    x = torch.empty((q_len, hidden), dtype=dt, device=dev)           # the residual stream
    for li, lw in enumerate(W.layers):
        qkv = torch.empty((q_len, n_qkv), dtype=dt, device=dev)
        if li == 0:
            ops.norm_linear(None, lw.ln1, eps, ts.frag("qkv", li), qkv, q_len, n_qkv, hidden, tiles=n_qkv // 16, ids=ids,
                            embed=W.embed, x_out=x)
        else:
            ops.norm_linear(x, lw.ln1, eps, ts.frag("qkv", li), qkv, q_len, n_qkv, hidden, tiles=n_qkv // 16)
        attn = attention_core(qkv, li, dims, kv_cache, model.cos, model.sin, pos, storage_ids, dense, tree, out_frag=True)
        ops.linear_ts(attn, ts.frag("o", li), q_len, hidden, hd, out=x, res=x, tiles=hidden // 16, splits=1)
        act = torch.empty(ops.frag_shape(q_len, inter), dtype=dt, device=dev)
        ops.norm_linear(x, lw.ln2, eps, ts.frag("gate_up", li), act, q_len, inter, hidden, swiglu=True, tiles=inter // 16)
        ops.linear_ts(act, ts.frag("down", li), q_len, hidden, inter, out=x, res=x, tiles=hidden // 16, splits=1)
    kv_cache.note_written(q_len)
    logits = logits_out if logits_out is not None else torch.empty((q_len, vocab), dtype=dt, device=dev)
    ops.norm_linear(x, W.norm, eps, ts.frag("lm_head"), logits, q_len, vocab, hidden, tiles=(vocab // 16 + 7) // 8)
    return logits.unsqueeze(0)


# Small drafts (heads of 64, hidden 512 / 768 / 1024: the 68m / 160m models), forwards whose rows never see each other (one
# tree level; any one-row forward): qkv projection + RoPE + KV write + tree attention + o_proj of a layer run as ONE launch
# per layer (csrc/draft_block.hip) instead of four -- each of the four sits on the ~5 us floor of a dependent graph node.
# o_proj comes out as per-head fp32 partials that the residual add + RMSNorm sums (splits = heads).
DRAFT_BLOCK = os.environ.get("SEQUOIA_DRAFT_BLOCK", "1") == "1"
BLOCK_MAX_ROWS = int(os.environ.get("SEQUOIA_DRAFT_BLOCK_ROWS", str(MAX_ROWS)))


def block_capable(model, q_len: int) -> bool:
    """The model / row count side of attn_block_ok (what a graph runner needs to know before it captures a variant)."""
    d, ts = model.dims, getattr(model, "ts", None)
    return (DRAFT_BLOCK and ts is not None and q_len <= min(BLOCK_MAX_ROWS, MAX_ROWS) and d.head_dim == 64
            and d.local_heads == d.local_kv_heads and d.local_heads <= 16 and d.hidden_size in (512, 768, 1024)
            and model.reduce_fn is None and d.tp_world == 1 and ts.shapes["o"][1] == d.local_heads * 64
            and ts.shapes["qkv"][0] == 3 * d.local_heads * 64 and ts._slab.numel() >= d.local_heads * q_len * d.hidden_size)


def attn_block_ok(model, ts: "TsLinearSet", q_len: int, tree) -> bool:
    if not (tree is not None and (q_len == 1 or tree.independent_rows) and tree.contiguous_slots
            and block_capable(model, q_len)):
        return False
    # the weight images are made outside any capture (the eager warm-up of a graph comes first)
    have = ("qkv", 0) in ts._frag and ("o", 0) in ts._frag
    return have or not torch.cuda.is_current_stream_capturing()


def forward_ts(model, ts: TsLinearSet, ids, q_len, pos, storage_ids, dense, tree, kv_cache):
    """Decoder forward of <= MAX_ROWS tree tokens on the tall-skinny projections.  ids: int64 [q] token ids.
    Returns logits [1, q, V].  Tensor-parallel shards (model.reduce_fn set): the partial output of the row-parallel
    projections (o_proj, down_proj) is reduced across ranks before the residual add; the vocabulary-parallel logits
    are gathered at the end (model.gather_logits_fn)."""
    stage = tree.stage if tree is not None else None
    # the caller wants the KV rows only (draft forward over the last tree level): stop after the last layer's KV write
    kv_only = tree is not None and not tree.need_logits
    if small_fused_ok(model, ts, q_len) and not kv_only:
        if stage is not None:
            get_ops().stage_tree_inputs(*stage)
        return forward_small_fused(model, ts, ids, q_len, pos, storage_ids, dense, tree, kv_cache)
    from .Llama_modules import attention_core
    ops = get_ops()
    W, dims = model.weights, model.dims
    eps = dims.rms_norm_eps
    plan = ts.plan(q_len)
    dev, dt = W.embed.device, W.embed.dtype
    hidden = dims.hidden_size
    ids = ids.contiguous()
    x = torch.empty((q_len, hidden), dtype=dt, device=dev)           # the residual stream
    inter = ts.shapes["down"][1]
    vocab = ts.shapes["lm_head"][0]
    fs = ops.frag_shape
    slab = ts._slab
    reduce_fn, gather_fn = model.reduce_fn, model.gather_logits_fn

    def reduced(pending):
        """Row-parallel projection under tensor parallelism: this rank's partial product as fp16 rows (split-K slabs
        are summed first), all-reduced over the ranks."""
        if reduce_fn is None:
            return pending
        if pending[0] == "slab":
            rows = torch.empty((q_len, hidden), dtype=dt, device=dev)
            slabs_fn = getattr(model, "reduce_slabs_fn", None)
            if slabs_fn is not None and slabs_fn(slab, pending[1], rows) is not None:
                return ("rows", rows)            # xGMI kernel: slab sum + all-reduce in one launch
            ops.add_rmsnorm_slabs(slab, pending[1], ts._zero_rows[:q_len], rows, None, None, eps)
        else:
            rows = pending[1]
        return ("rows", reduce_fn(rows))

    reduce_norm_fn = getattr(model, "reduce_norm_fn", None) if reduce_fn is not None else None
    use_block = attn_block_ok(model, ts, q_len, tree)
    if use_block and not torch.cuda.is_current_stream_capturing():
        for li in range(len(W.layers)):
            ts.frag("qkv", li); ts.frag("o", li)

    def reduce_norm(partial, weight, want_frag):
        """Tensor-parallel continuation of a row-parallel projection: x += all-reduce(partial); RMSNorm into the next operand.
        One kernel when the xGMI collectives are up (row-aligned all-reduce that finishes the rows it gathers)."""
        if reduce_norm_fn is not None:
            out = torch.empty(fs(q_len, hidden) if want_frag else (q_len, hidden), dtype=dt, device=dev)
            src, splits = (slab, partial[1]) if partial[0] == "slab" else (partial[1], 0)
            if reduce_norm_fn(src, splits, x, weight, out, eps, want_frag) is not None:
                return out
        return norm_into(reduced(partial), weight, want_frag)

    def norm_into(pending, weight, want_frag):
        """Apply the pending branch output (None | ("rows", t) | ("slab", splits)) to the residual stream x,
        then RMSNorm into a row-major or fragment-major buffer."""
        out = torch.empty(fs(q_len, hidden) if want_frag else (q_len, hidden), dtype=dt, device=dev)
        if pending is None:
            (ops.rmsnorm_frag if want_frag else ops.rmsnorm)(x, weight, out, eps)
        elif pending[0] == "slab":
            ops.add_rmsnorm_slabs(slab, pending[1], x, x, weight, out, eps, out_frag=want_frag)
        elif want_frag:
            ops.add_rmsnorm_frag(pending[1], x, x, weight, out, eps)
        else:
            ops.add_rmsnorm(pending[1], x, x, weight, out, eps)
        return out

    def project(name, li, a):
        """One projection (a: fragment-major when the plan uses the kernel, row-major otherwise) with output to rows
        or split-K slabs.  Returns the pending record."""
        p = plan[name]
        n_out, k, _ = ts.shapes[name]
        if p is None:
            return ("rows", F.linear(a, ts._row_major(name, li)))
        tiles, splits = p
        if splits > 1:
            ops.linear_ts(a, ts.frag(name, li), q_len, n_out, k, tiles=tiles, splits=splits, slab=slab)
            return ("slab", splits)
        out = torch.empty((q_len, n_out), dtype=dt, device=dev)
        ops.linear_ts(a, ts.frag(name, li), q_len, n_out, k, out=out, tiles=tiles, splits=1)
        return ("rows", out)

    pending = None
    for li, lw in enumerate(W.layers):
        if li == 0:                                              # embedding lookup + first norm, one launch
            want = use_block or plan["qkv"] is not None
            h = torch.empty(fs(q_len, hidden) if want else (q_len, hidden), dtype=dt, device=dev)
            if stage is not None:      # device-driven step: ids / positions / slots / context staged by the same launch
                ops.embed_stage_rmsnorm(stage, W.embed, lw.ln1, x, h, eps, out_frag=want)
            else:
                ops.embed_rmsnorm(ids, W.embed, lw.ln1, x, h, eps, out_frag=want)
        else:
            h = reduce_norm(pending, lw.ln1, use_block or plan["qkv"] is not None)
        stop = kv_only and li == len(W.layers) - 1
        if use_block:            # qkv + RoPE + KV write + attention + o_proj partials: one launch
            nh = dims.local_heads
            ops.draft_attn_block(h, ts.frag("qkv", li), None if stop else ts.frag("o", li), slab, kv_cache.k_cache[li, 0],
                                 kv_cache.v_cache[li, 0], model.cos, model.sin, pos, storage_ids, q_len, nh, dims.head_dim,
                                 hidden, dims.head_dim ** -0.5, tree.q_slot0, tree.gt, tree.n_tree, tree.bitmask, tree.ctx,
                                 kv_only=stop)
            o_part = ("slab", nh)
        else:
            qkv = project("qkv", li, h)
            if qkv[0] == "slab":
                attn = attention_core(None, li, dims, kv_cache, model.cos, model.sin, pos, storage_ids, dense, tree,
                                      out_frag=plan["o"] is not None, qkv_slab=(slab, qkv[1], q_len), kv_only=stop)
            else:
                attn = attention_core(qkv[1], li, dims, kv_cache, model.cos, model.sin, pos, storage_ids, dense, tree,
                                      out_frag=plan["o"] is not None, kv_only=stop)
            o_part = None if stop else project("o", li, attn)
        if stop:                 # nothing downstream of the last layer's K / V rows is needed
            kv_cache.note_written(q_len)
            return None
        h = reduce_norm(o_part, lw.ln2, plan["gate_up"] is not None)
        down_ts = plan["down"] is not None
        act = torch.empty(fs(q_len, inter) if down_ts else (q_len, inter), dtype=dt, device=dev)
        if plan["gate_up"] is not None:
            tiles, gsplits = plan["gate_up"]
            if gsplits > 1:      # shapes whose activation block outweighs the weights (TP shards): split K, SwiGLU from the slabs
                ops.linear_ts(h, ts.frag("gate_up", li), q_len, 2 * inter, hidden, tiles=tiles, splits=gsplits, slab=slab)
                ops.silu_mul_slabs(slab, gsplits, act, q_len, inter, out_frag=down_ts)
            else:
                ops.linear_ts(h, ts.frag("gate_up", li), q_len, inter, hidden, out=act, silu=True, out_frag=down_ts, tiles=tiles)
        else:
            gu = F.linear(h, lw.w_gate_up)
            if down_ts:
                ops.silu_mul_frag(gu, act, q_len, inter)
            else:
                ops.silu_mul(gu, act)
        pending = project("down", li, act)
    h = reduce_norm(pending, W.norm, plan["lm_head"] is not None)
    kv_cache.note_written(q_len)
    if plan["lm_head"] is not None:
        tiles, _ = plan["lm_head"]
        logits = torch.empty((q_len, vocab), dtype=dt, device=dev)
        ops.linear_ts(h, ts.frag("lm_head"), q_len, vocab, hidden, out=logits, tiles=tiles)
    else:
        logits = F.linear(h, W.lm_head)
    if gather_fn is not None:
        logits = gather_fn(logits)
    return logits.unsqueeze(0)
