"""The reference's evaluation loop (tests/testbed.py:45-95, `simulation_fast`) as a reusable object:
engine construction for the BASELINE.json configurations, the 128-token c4_small prompts, and a
resumable speculation-step iterator.  Shared by bench.py, the growmap tuner and the GPU tests.
"""
from __future__ import annotations

import json
import os
import time

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# whole-step graphs kept in flight by the device-driven loop (the result ring has SQ_RESULT_RING = 4 slots)
PIPE_DEPTH = max(1, min(3, int(os.environ.get("SEQUOIA_PIPE_DEPTH", "2"))))


def _sync(device):
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()

MODELS = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable plumbing case (2-node chain)
    "A": dict(draft="JackFram/llama-68m", target="meta-llama/Llama-2-7b-hf", growmap="2-chain", mode="stochastic", M=384),
    "B": dict(draft="JackFram/llama-68m", target="meta-llama/Llama-2-7b-hf", growmap="A100-CNN-68m-7b-stochastic",
              mode="stochastic", M=384),
    "C": dict(draft="JackFram/llama-68m", target="meta-llama/Llama-2-7b-hf", growmap="8x8-tree", mode="greedy", M=384),
    "D": dict(draft="princeton-nlp/Sheared-LLaMA-1.3B", target="meta-llama/Llama-2-13b-hf",
              growmap="A100-CNN-160m-13b-stochastic", mode="stochastic", M=384),
    # the large-tree path (README.md:47,54: the reference's 256-node growmap, M >= tree + 256): a verify forward of 256 rows,
    # beyond the tall-skinny kernel's 144 -- hipBLASLt projections inside the captured step
    "L": dict(draft="JackFram/llama-68m", target="meta-llama/Llama-2-13b-hf", growmap="A100-CNN-68m-13b-stochastic-S256",
              mode="stochastic", M=512),
    # 70B target sharded tensor-parallel over all launched ranks (replaces the reference's host offload);
    # every rank runs the replicated draft + verifier, so N ranks serve ONE request stream ("strong")
    "E": dict(draft="meta-llama/Llama-2-7b-hf", target="meta-llama/Llama-2-70b-hf", growmap="64x2-tree",
              mode="stochastic", M=1024, tp=True),
}


def tp_capturable(*engines) -> bool:
    """Can the tensor-parallel forward be captured into a hipGraph?  Yes with RCCL (backend "nccl"), without a process
    group, or when the engine runs both collectives on the xGMI kernels (no torch.distributed call in a tree forward);
    not when a gloo collective (through the host) is on the path.  SEQUOIA_TP_GRAPHS=0 forces eager forwards."""
    if os.environ.get("SEQUOIA_TP_GRAPHS", "1") == "0":
        return False
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return True
    if dist.get_backend() == "nccl":
        return True
    # every tensor-parallel engine of the step (the target, and the draft when it is sharded too) must be free of
    # torch.distributed calls; replicated engines have no collective at all
    tp = [e.engine for e in engines if e is not None and hasattr(getattr(e, "engine", None), "group")]
    return bool(tp) and all(getattr(inner, "collectives_capturable", False) for inner in tp)


def load_prompts():
    with open(os.path.join(_PKG, "growmaps", "c4_small_prompts.json")) as f:
        return json.load(f)["prompts"]


def build(cfg, device, pair, seed_d=1, seed_t=2):
    """Draft engine, target engine and growmap of one MODELS entry.  `pair`: "calibrated" (synthetic
    draft/target pair with a realistic acceptance rate, sequoia_amd/synthetic.py) or "random"."""
    from sequoia_amd.Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG
    from sequoia_amd.growmap import GrowMap
    M = cfg["M"]
    if cfg.get("tp") and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from sequoia_amd.Engine import ts_linear
        ts_linear.DETERMINISTIC_PLANS = True       # replicated decisions need identical arithmetic on every rank
    tp_world = int(os.environ.get("WORLD_SIZE", "1")) if cfg.get("tp") else 1
    tp_rank = int(os.environ.get("RANK", "0")) if cfg.get("tp") else 0
    # The 7B draft of configuration E can be sharded like the target (SEQUOIA_TP_DRAFT=1: heads / MLP columns / vocabulary
    # over the ranks, 2 all-reduces per layer on the xGMI kernel, logits gathered; sampler and verifier stay replicated, the
    # gathered logits are bit-identical on every rank).  Replicated it streams 13.5 GB per forward on every GPU (3 forwards
    # per step: about as long as the sharded 70B verify).  Measured per GPU with the tuned shard plans
    # (profiles/r03_ts_linear_tuning_tp_draft7b.json): at TP = 8 a sharded draft layer is 31 us of projections + ~50 us of
    # launch-floor kernels and two all-reduces against 97 us replicated -- a wash that depends on the all-reduce latency of
    # the real links, so the default stays replicated (no collective on the draft path); at TP = 2 sharding wins (the
    # two-ranks-on-one-GPU run: 76.8 -> 70.2 ms / step).
    tp_draft = tp_world > 1 and os.environ.get("SEQUOIA_TP_DRAFT", "0") == "1"
    if pair == "calibrated":
        from sequoia_amd.synthetic import calibrated_pair_specs
        dspec, tspec = calibrated_pair_specs(cfg["draft"], cfg["target"], device, tp_world=tp_world, tp_rank=tp_rank,
                                             draft_tp=tp_draft)
    else:
        dspec, tspec = f"random:{cfg['draft']}:seed={seed_d}", f"random:{cfg['target']}:seed={seed_t}"
    if tp_draft:
        from sequoia_amd.Engine.tp_engine import TPEngine
        draft = TPEngine(max_length=M, model_name_or_path=dspec, dtype=torch.float16, device=device)
    else:
        draft = GraphInferenceEngine(max_length=M, model_name_or_path=dspec, dtype=torch.float16, device=device)
    if cfg.get("tp"):
        from sequoia_amd.Engine.offload_engine import OffloadEngine
        target = OffloadEngine(max_length=M, model_name_or_path=tspec, dtype=torch.float16, device=device)
    else:
        target = GraphInferenceEngineTG(max_length=M, model_name_or_path=tspec, dtype=torch.float16, device=device)
    gm = GrowMap.load(cfg["growmap"])
    return draft, target, gm


class Loop:
    """simulation_fast (tests/testbed.py:45-95) as a resumable step iterator."""

    def __init__(self, cfg, draft, target, gm, device, prompts, use_graphs=True, T=0.6, top_p=1.0, max_new=256, vocab=32000,
                 pipelined=False):
        """pipelined: after a prompt's first (prefill-bearing) step the loop is driven by the device
        (Tree/step_graph.py): whole steps are enqueued as one hipGraph each, up to two in flight, and their result
        records are read one step late -- same tokens as the synchronous loop, no host round trip between steps."""
        from sequoia_amd.Tree.GreedyTree import GreedyTree
        from sequoia_amd.Tree.SpecTree import SpecTree
        from sequoia_amd.Tree.Tree import growmap_on_device
        self.cfg, self.draft, self.target, self.device, self.prompts, self.T = cfg, draft, target, device, prompts, T
        self.top_p, self.max_new, self.vocab = top_p, max_new, vocab
        self.grow_map = gm.to_reference_dict()
        self.gm_obj = gm
        self.cls = SpecTree if cfg["mode"] == "stochastic" else GreedyTree
        M = cfg["M"]
        self.attn_mask = torch.full((M, M), torch.finfo(torch.float16).min, dtype=torch.float16, device=device)
        self.position_ids = torch.zeros(M, dtype=torch.long, device=device)
        g, gdev = growmap_on_device(self.grow_map, device)
        if use_graphs and (not cfg.get("tp") or tp_capturable(target, draft)):
            lens = sorted({lv.total for lv in g.levels} | {1})
            draft.initialize_cuda_graph(lens, tree_bitmask=gdev["bitmask"], n_tree=g.size)
            # (tensor-parallel target: the verify forward is captured WITH its collectives -- the xGMI all-reduce kernel
            # and RCCL calls are stream-ordered and legal inside a capture; a gloo group is not: eager then)
            if hasattr(target, "initialize_cuda_graph") and (not cfg.get("tp") or tp_capturable(target, draft)):
                target.initialize_cuda_graph([g.size], tree_bitmask=gdev["bitmask"], n_tree=g.size)
        self.pi = 0
        self.tree = None
        self.cur_len = 0
        self.prefill_steps = 0       # steps that carried a prompt's target prefill (the first verify of every prompt)
        self.prefill_seconds = 0.0   # wall time of those steps (construct_grow_map + verify; verify ends on a result read)
        self.prefill_tokens = 0      # tokens those steps committed
        self.prompts_done = 0        # prompts generated to the end (max_new tokens, EOS, or a terminal step)
        # device-driven steps under tensor parallelism too: the step block, the result ring and the decisions are per
        # rank and identical on every rank (replicated draft / sampler / verifier, same noise), so every rank replays
        # the same whole-step graph -- collectives included -- without any broadcast
        self.pipelined = bool(pipelined) and str(device).startswith("cuda") and (not cfg.get("tp") or tp_capturable(target, draft))

    def _new_prompt(self):
        self.draft.clear_kv(); self.target.clear_kv()
        p = torch.tensor(self.prompts[self.pi % len(self.prompts)][:128], dtype=torch.long)
        self.pi += 1
        M = self.cfg["M"]
        self.tree = self.cls(prefix=p, device=self.device, temperature=self.T, top_p=self.top_p, draft_kv_len=0,
                             target_kv_len=0, draft_model_engine=self.draft, target_model_engine=self.target,
                             max_length=M, max_target_seq=M, grow_map=self.grow_map, attn_mask=self.attn_mask,
                             sequence=None, new_tokens_buffer=None, parents_buffer=None,
                             position_ids=self.position_ids, residual_graph=None, sampling_callables=None,
                             sample_gather_indices=None, vocab_size=self.vocab,
                             **(dict(step_graph=True) if self.pipelined else {}))
        self.cur_len = len(p)

    def start_fresh_prompt(self):
        """Drop a prompt that run_steps() left half-way: the next step is the prefill-bearing first step of a new prompt
        (bench.py starts its timed window there, so the window holds the steps tests/testbed.py:78-95 times -- the first
        verify of a prompt with its target prefill included -- not only steady ones)."""
        if self.tree is not None:
            if self.pipelined and getattr(self.tree, "_pipe", None) is not None:
                self.tree.end_pipeline()
            self.tree = None

    def run_prompts(self, n_prompts, max_steps=100000):
        """Run until n_prompts more prompts are complete (each from its prefill-bearing first step to max_new tokens / EOS):
        the reference's metric as tests/testbed.py:78-95 computes it -- total_time / tokens over WHOLE prompts, the
        target prefill inside the first verify included.  Returns (seconds, new_tokens, steps)."""
        self.start_fresh_prompt()          # a prompt left half-way by run_steps(): start from a fresh one
        return self.run_steps(max_steps, stop_after_prompts=self.prompts_done + n_prompts)

    def run_steps(self, k_steps, on_step=None, on_accept=None, stop_after_prompts=None):
        """Run exactly k_steps speculation steps; per-prompt setup (tree constructor + draft
        prefill) is outside the timed brackets like the reference (tests/testbed.py:67-79).
        `on_step(tree, terminate)` is called after every verify() (synchronous mode), `on_accept(accept_length)` after
        every step in both modes.  stop_after_prompts: also stop once self.prompts_done reaches it (run_prompts).
        Returns (seconds, new_tokens, steps)."""
        total_t, new_tok, done = 0.0, 0, 0
        more = (lambda: True) if stop_after_prompts is None else (lambda: self.prompts_done < stop_after_prompts)
        while done < k_steps and more():
            if self.tree is None:
                self._new_prompt()
            _sync(self.device)
            t1 = time.perf_counter()
            tree = self.tree
            piped = self.pipelined and tree.state is not None
            while done < k_steps and self.tree is not None:
                if piped and tree._pipe is not None:
                    # device-driven: keep up to two whole-step graphs in flight, read results one step late
                    while (len(tree._pipe["inflight"]) < PIPE_DEPTH and tree.can_enqueue(self.max_new)
                           and done + len(tree._pipe["inflight"]) < k_steps):
                        tree.enqueue_step()
                    if not tree._pipe["inflight"]:
                        tree.end_pipeline()            # out of run-ahead room: finish the prompt synchronously
                        continue
                    a, _, bonus, terminate = tree.collect_step()
                    length = a if terminate else a + 1
                    last = bonus
                else:
                    is_prefill = tree.target_kv_len == 0
                    t_p = time.perf_counter()
                    tree.construct_grow_map()
                    valid, a, _, terminate = tree.verify()
                    length = valid.shape[0]
                    if is_prefill:
                        self.prefill_steps += 1
                        self.prefill_seconds += time.perf_counter() - t_p
                        self.prefill_tokens += length - self.cur_len
                    last = int(valid[-1])              # the reference's EOS test reads the last token (tests/testbed.py:80)
                    if on_step is not None:
                        on_step(tree, terminate)
                    if (piped and not (terminate or length >= self.max_new or last in (0, 2)) and not tree._no_room):
                        tree.begin_pipeline()          # the target is prefilled: hand the loop to the device
                new_tok += length - self.cur_len
                self.cur_len = length
                done += 1
                if on_accept is not None:
                    on_accept(int(a))
                if terminate or self.cur_len >= self.max_new or last in (0, 2):
                    if piped:
                        tree.end_pipeline()
                    self.tree = None
                    self.prompts_done += 1
            if piped and self.tree is not None:
                tree.end_pipeline()                    # k_steps reached mid-prompt: back to host-side state
            _sync(self.device)
            total_t += time.perf_counter() - t1
        return total_t, new_tok, done


class AutoregressiveLoop:
    """The reference's autoregressive baseline (`simulation_baseline`, tests/testbed.py:99-143): the target
    model alone, one token per forward, `softmax(logits / T)` sampled once per step.  The 1-token forward is a
    hipGraph replay with a single-node "tree" (the query sees the committed prefix and itself); the draw is
    sq_sample_wor_f16 with k = 1 — argmax of log(u)/p, an exact sample from p — in place of `p.multinomial(1)`.
    The token is read back every step like the reference's EOS test (:137)."""

    def __init__(self, cfg, target, device, prompts, T=0.6, max_steps=32, use_graphs=True, vocab=32000):
        from sequoia_amd.Engine.Llama_modules import TreeContext
        from sequoia_amd.ops import get_ops
        self.cfg, self.target, self.device, self.prompts, self.T = cfg, target, device, prompts, T
        self.max_steps, self.use_graphs = max_steps, use_graphs
        self.ops = get_ops()
        self.TreeContext = TreeContext
        self.bitmask = torch.ones((1, 1), dtype=torch.int64, device=device)
        if use_graphs and hasattr(target, "initialize_cuda_graph"):
            target.initialize_cuda_graph([1], tree_bitmask=self.bitmask, n_tree=1)
        self.rand = torch.empty((max_steps, vocab), dtype=torch.float16).uniform_().to(device)
        self.row = torch.zeros(1, dtype=torch.int32, device=device)
        self.tok = torch.zeros(1, dtype=torch.long, device=device)
        self.arange = torch.arange(cfg["M"], device=device)
        self.pi = 0

    def _ctx(self, q_slot0, kv_len):
        # gt = kv_len: the last query is tree node 0 (the root), everything before it is committed text
        return self.TreeContext(q_slot0=q_slot0, gt=kv_len, n_tree=1, bitmask=self.bitmask, kv_len=kv_len,
                                contiguous_slots=True)

    @torch.inference_mode()
    def run_prompt(self):
        """One prompt: prefill (untimed) + up to max_steps decode steps.  Returns (seconds, tokens)."""
        tgt = self.target
        tgt.clear_kv()
        p = torch.tensor(self.prompts[self.pi % len(self.prompts)][:128], dtype=torch.long, device=self.device)
        self.pi += 1
        n = len(p)
        logits = tgt.inference(input_ids=p.unsqueeze(0), storage_ids=self.arange[:n], position_ids=self.arange[:n].unsqueeze(0),
                               attn_mask=None, tree=self._ctx(0, n))[0, -1:]
        run = getattr(tgt, "graph_inference", None) if self.use_graphs else None
        kw = dict(borrow=True) if run is not None else {}
        run = run or tgt.inference
        _sync(self.device)
        t0 = time.perf_counter()
        done = 0
        for step in range(self.max_steps):
            self.ops.sample_wor(logits, self.rand[step:step + 1], self.row, 1, self.T, self.tok)
            done += 1
            if int(self.tok[0]) in (0, 2) or n + step + 1 >= self.cfg["M"]:
                break
            s = n + step
            logits = run(input_ids=self.tok.unsqueeze(0), storage_ids=self.arange[s:s + 1],
                         position_ids=self.arange[s:s + 1].unsqueeze(0), attn_mask=None, tree=self._ctx(s, s + 1), **kw)[0]
        _sync(self.device)
        dt = time.perf_counter() - t0
        tgt.clear_kv()
        return dt, done

    def run(self, n_prompts=3):
        secs = toks = 0
        for _ in range(n_prompts):
            dt, k = self.run_prompt()
            secs += dt; toks += k
        return dict(tokens_per_s=toks / secs, ms_per_token=secs / toks * 1e3, tokens=toks, prompts=n_prompts)
