"""The REFERENCE'S OWN harness source executed against (a) the reference's classes and (b) this package's drop-in modules
-- TEST INFRASTRUCTURE ONLY (never imported by sequoia_amd/).

`tests/testbed.py` of the reference cannot be imported (it parses argv, loads a tokenizer and a dataset at import time), so
the three pieces the drop-in claim is about are compiled from the file's AST and executed UNMODIFIED:
  * `def setup_seed` and `def simulation_fast`                                   (tests/testbed.py:35-41, 45-95)
  * the engine / growmap / sampler set-up block, i.e. the `else:` branch of `if args.Mode == 'baseline':`   (:250-285)
  * the call `simulation_fast(target_model=..., ...)` of the `elif args.Mode == 'greedy':` branch          (:296-298)
What is supplied from the OUTSIDE is the environment those lines name, nothing inside them:
  * `args` (a namespace), `dataloader` (a list of {input_ids, labels} batches), `tqdm` (identity);
  * `SpecTree`: a subclass of the class under test (the reference's, or the drop-in's) that forces device = "cpu", logs the
    tokens every `verify()` returns, and pins the one draw the two implementations make from different generators: the bonus
    token (reference: `residual.multinomial(1)`, Tree/SpecTree.py:222 -> the exact inverse CDF at a recorded 24-bit uniform,
    shim 6 of oracle/gen_golden.py; drop-in: the same uniforms through its `bonus_uniforms` argument);
  * `GraphInferenceEngine` / `GraphInferenceEngineTG`: factories that build the engines on the CPU from seeded weights
    instead of `from_pretrained` (no checkpoints offline; shim 4) -- the reference's `initialize_cuda_graph` becomes a
    no-op there (no CUDA graphs on a CPU: `graph_inference` then runs eagerly, Engine/Engine.py:215-218);
  * `cuda_graph_for_residual` / `cuda_graph_for_sampling_without_replacement`: the reference's are replaced by the plain
    functions they capture (shim 5); the drop-in's are its own;
  * a torch-function mode that maps the literal "cuda:0" of the harness lines to the CPU, and `torch.cuda.synchronize` -> no-op.

    python oracle/ref_harness.py reference out.npz [seed [growmap]]   # subprocess side: the reference's classes (needs /root/reference)
    run_dropin(npz)                                         # in-process side: sequoia_amd.dropin + the oracle adapter
"""
from __future__ import annotations

import ast
import contextlib
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("SEQUOIA_REFERENCE", "/root/reference")
GROWMAP = os.environ.get("SEQUOIA_HARNESS_GROWMAP", "L40_growmaps/4x8-tree.pt")     # 33 nodes, 8 draft levels of 4 (the CPU side runs the numpy oracle: a 128-node tree costs 2 s per step)
TINY = (128, 344, 2, 2, 2)           # hidden, intermediate, layers, heads, kv heads (the dims of the live traces)
# the harness constructs SpecTree without `vocab_size` (tests/testbed.py:70-79), i.e. at the class default 32000: the models
# carry the real vocabulary.  Weights are seeded (oracle/seeded_weights.py) so that a record holds tokens, not matrices.
# GAIN: lm_head scale.  At 10 the tiny random pair's distributions are flat (hundreds of tokens survive the nucleus filter, the
# residual relu(p - q) / sum is a difference of nearly equal fp16 numbers): two arithmetics one fp16 ulp apart then disagree on
# a bonus draw in most runs (measured: 5 of 6 seeds).  At 30 the rows are peaked like a language model's (a handful of kept
# tokens) and 5 of 6 UNSCREENED seeds replay token-identical over the whole harness run (tools/harness_seed_rate.py).
VOCAB, M_LEN, TEMP, TOP_P = 32000, 512, 0.6, 0.9    # (M = 384 + headroom: a step that starts at 255 tokens may commit 10 more)
GAIN, SHARE, BRANCH = float(os.environ.get("SEQUOIA_HARNESS_GAIN", "30")), float(os.environ.get("SEQUOIA_HARNESS_SHARE", "0.05")), 0.05
STEPS_PER_PROMPT = 64                # bonus uniforms reserved per prompt


# ---- the reference's lines -------------------------------------------------------------------------------------------
def compile_harness(ref=REF):
    path = os.path.join(ref, "tests", "testbed.py")
    tree = ast.parse(open(path).read())
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("setup_seed", "simulation_fast")]
    assert [d.name for d in defs] == ["setup_seed", "simulation_fast"], [d.name for d in defs]
    setup = next(n for n in tree.body if isinstance(n, ast.If) and ast.unparse(n.test) == "args.Mode == 'baseline'" and n.orelse)
    chain = next(n for n in tree.body if isinstance(n, ast.If) and ast.unparse(n.test) == "args.Mode == 'benchmark'")
    call = None
    while chain is not None:
        if ast.unparse(chain.test) == "args.Mode == 'greedy'":
            call = chain.body
            break
        chain = chain.orelse[0] if chain.orelse and isinstance(chain.orelse[0], ast.If) else None
    assert call is not None and len(call) == 1 and "simulation_fast(" in ast.unparse(call[0])
    # the call statement discards the function's value; keep it (an assignment wrapped AROUND the reference's expression)
    keep = ast.Assign(targets=[ast.Name(id="_harness_result", ctx=ast.Store())], value=call[0].value)
    ast.copy_location(keep, call[0]); ast.fix_missing_locations(keep)
    mod = lambda body: compile(ast.Module(body=body, type_ignores=[]), path, "exec")
    lines = dict(defs=[(d.lineno, d.end_lineno) for d in defs], setup=(setup.orelse[0].lineno, setup.orelse[-1].end_lineno),
                 call=(call[0].lineno, call[0].end_lineno))
    return dict(defs=mod(defs), setup=mod(setup.orelse), call=mod([keep]), lines=lines)


class _CudaToCpu(torch.overrides.TorchFunctionMode):
    """The harness lines say device='cuda:0' / .to('cuda:0'): here that is the CPU."""

    @staticmethod
    def _fix(x):
        if isinstance(x, str) and x.startswith("cuda"):
            return "cpu"
        if isinstance(x, torch.device) and x.type == "cuda":
            return torch.device("cpu")
        return x

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if "device" in kwargs:
            kwargs["device"] = self._fix(kwargs["device"])
        return func(*[self._fix(a) for a in args], **kwargs)


@contextlib.contextmanager
def cpu_environment():
    saved = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        with _CudaToCpu():
            yield
    finally:
        torch.cuda.synchronize = saved


def make_inputs(seed):
    """Three prompts: two that decode up to 256 tokens, one whose labels end in -100 (the harness skips it, :64)."""
    g = torch.Generator().manual_seed(seed)
    lens = [216, 226, 40]
    batches = []
    for i, n in enumerate(lens):
        ids = torch.randint(3, VOCAB, (1, n), generator=g)
        labels = ids.clone()
        if i == 2:
            labels[0, -1] = -100
        batches.append(dict(input_ids=ids, labels=labels))
    u24 = np.random.RandomState(seed + 1).randint(0, 1 << 24, size=len(lens) * STEPS_PER_PROMPT).astype(np.int64)
    return batches, u24


def seeded_pair(seed, gain=None, share=None, branch=None):
    """(draft, target) state dicts + their checksums: the same bits on both sides."""
    from oracle import seeded_weights as SW
    gain, share, branch = GAIN if gain is None else gain, SHARE if share is None else share, BRANCH if branch is None else branch
    sd_t = SW.seeded_state_dict(TINY, VOCAB, 1000 + seed, gain, branch_scale=branch)
    sd_d = SW.seeded_state_dict(TINY, VOCAB, 2000 + seed, gain, branch_scale=branch)
    SW.correlate(sd_d, sd_t, share, 3000 + seed)
    return sd_d, sd_t, [str(SW.checksum(sd_d)), str(SW.checksum(sd_t))]


def noise_seed(seed, prompt):
    """Both SpecTree flavours draw r and rand on the CPU generator inside their constructor (Tree/SpecTree.py:60,84): the
    generator is pinned right in front of it, per prompt, on both sides."""
    return 7000 + 131 * seed + prompt


def harness_args(seed, growmap=None):
    return types.SimpleNamespace(model="draft", target="target", dataset="synthetic", growmap=os.path.join(REF, growmap or GROWMAP), start=0, end=3,
                                 T=TEMP, P=TOP_P, M=M_LEN, seed=seed, Mode="greedy", offloading=False)


def run_harness(code, ns, seed):
    """Execute the compiled reference lines in `ns` (which carries the outside environment).  Returns the harness's value."""
    with cpu_environment():
        exec(code["defs"], ns)
        ns["setup_seed"](ns["args"].seed)
        exec(code["setup"], ns)
        exec(code["call"], ns)
    return ns["_harness_result"]


# ---- (a) the reference's classes ---------------------------------------------------------------------------------------
def run_reference(out_path, seed, growmap=None):
    growmap = growmap or GROWMAP
    sys.path.insert(0, HERE)
    import gen_golden as GG
    from oracle import ops_np
    R = GG.import_reference()
    RU = R["RU"]
    code = compile_harness()
    batches, u24 = make_inputs(seed)
    log, state = [], dict(prompt=-1, step=0)
    engines = {}

    sd_d, sd_t, checks = seeded_pair(seed)

    def engine_factory(outer, inner, model_cls, which, sd):
        def build(max_length, model_name_or_path, dtype=torch.float16, device="cuda:0", **kw):
            assert model_name_or_path == which
            cfg = GG.make_cfg(R, TINY, VOCAB)
            e = GG.make_engine(R, outer, inner, model_cls, cfg, max_length, 1, 1.0)          # shim 4
            missing, unexpected = e.engine.model.load_state_dict(sd, strict=False)
            assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
            e.initialize_cuda_graph = lambda *a, **k: None            # no CUDA graphs on a CPU: graph_inference runs eagerly
            engines[which] = e
            return e
        return build

    class SpySpecTree(R["SpecTree"]):
        def __init__(self, **kw):
            kw["device"] = "cpu"
            state["prompt"] += 1
            state["step"] = 0
            torch.manual_seed(noise_seed(seed, state["prompt"]))
            super().__init__(**kw)

        def verify(self, benchmark=False):
            state["residual"] = None
            out = super().verify(benchmark=benchmark)
            log.append((state["prompt"], out[0].clone().numpy(), state["step"], state["residual"]))
            state["step"] += 1
            return out

    orig_multinomial = torch.Tensor.multinomial

    def fake_multinomial(self, num_samples=1, replacement=False, generator=None):          # shim 6
        p = self.detach().clone().numpy()
        state["residual"] = p
        tok = ops_np.inverse_cdf(p, int(u24[state["prompt"] * STEPS_PER_PROMPT + state["step"]]))
        return torch.tensor([tok], dtype=torch.long)

    ns = dict(torch=torch, time=time, np=np, random=__import__("random"), tqdm=lambda it, total=None: it, print=print,
              DataLoader=object, args=harness_args(seed, growmap), dataloader=batches, SpecTree=SpySpecTree,
              GraphInferenceEngine=engine_factory(R["GIE"], R["IE"], R["MM"].LlamaForCausalLM_FI, "draft", sd_d),
              GraphInferenceEngineTG=engine_factory(R["GIETG"], R["IETG"], R["MM"].LlamaForCausalLM_TG, "target", sd_t),
              OffloadEngine=None,
              cuda_graph_for_residual=lambda *a, **k: RU.get_residual,                                          # shim 5
              cuda_graph_for_sampling_without_replacement=lambda **k: (
                  lambda lg, rnd: RU.sampling_without_replacement(lg, rnd, k["num_samples"], k["temperature"])))
    torch.Tensor.multinomial = fake_multinomial
    try:
        t0 = time.time()
        value = run_harness(code, ns, seed)
        dt = time.time() - t0
    finally:
        torch.Tensor.multinomial = orig_multinomial
    arrays = {}
    for i, b in enumerate(batches):
        arrays[f"prompt{i}/input_ids"] = b["input_ids"].numpy()
        arrays[f"prompt{i}/labels"] = b["labels"].numpy()
    arrays["bonus_u24"] = u24
    # The committed text is append-only, so the tokens verify() call j returned are the first verify_len[j] tokens of its
    # prompt's final sequence -- checked here, then stored that way (a record is a few KB).
    final = {}
    for p, toks, step, res in log:
        assert p not in final or np.array_equal(final[p], toks[:len(final[p])]), "the committed text changed retroactively"
        final[p] = toks
    for p, toks in final.items():
        arrays[f"prompt{p}/final"] = toks
    arrays["verify_prompt"] = np.array([p for p, *_ in log], dtype=np.int64)
    arrays["verify_len"] = np.array([len(t) for _, t, *_ in log], dtype=np.int64)
    arrays["verify_step"] = np.array([st for _, _, st, _ in log], dtype=np.int64)
    # the distribution every bonus token was drawn from, sparse (peaked rows: a handful of tokens carry mass): what proves a
    # boundary draw when another arithmetic picks the neighbouring token
    for j, (p, toks, step, res) in enumerate(log):
        if res is not None:
            nz = np.nonzero(np.nan_to_num(res.astype(np.float32)) > 0)[0]
            arrays[f"verify{j}/res_ids"] = nz.astype(np.int32)
            arrays[f"verify{j}/res_p"] = res[nz].astype(np.float16)
    g = torch.load(os.path.join(REF, growmap), weights_only=False)
    meta = dict(seed=seed, value=float(value), n_verify=len(log), n_prompts=len(batches), dims=list(TINY), vocab=VOCAB, M=M_LEN,
                successors=g["Successors"], gain=GAIN, share=SHARE, branch=BRANCH,
                T=TEMP, top_p=TOP_P, growmap=growmap, lines=code["lines"], seconds=round(dt, 1), torch=torch.__version__,
                weight_checksums=checks, accepted_per_verify=[int(len(t[1])) for t in log])
    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(out_path, **arrays)
    print("reference harness:", meta)


# ---- (b) the drop-in ---------------------------------------------------------------------------------------------------
def load_record(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


def record_tokens(z, j):
    """The tokens verify() call j handed to the harness loop."""
    return z[f"prompt{int(z['verify_prompt'][j])}/final"][:int(z["verify_len"][j])]


def record_residual(z, j, vocab):
    """Dense fp16 distribution the bonus token of call j was drawn from (None when the step was terminal)."""
    if f"verify{j}/res_ids" not in z.files:
        return None
    p = np.zeros(vocab, dtype=np.float16)
    p[z[f"verify{j}/res_ids"]] = z[f"verify{j}/res_p"]
    return p


def classify_run(z, meta, log):
    """Compare a run's verify log [(prompt, tokens), ...] with a record.  Returns ("identical", None), or
    ("boundary", (call, distance)): the runs agree on every token up to ONE bonus draw, whose recorded uniform lies `distance`
    (probability mass under the reference's own residual) from the interval of the token this run drew, or
    ("differs", (call, why)): anything else."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import cdf_interval_distance
    n = meta["n_verify"]
    for j, (p, toks) in enumerate(log):
        if j >= n:
            return "differs", (j, "more verify calls than the reference made")
        ref = record_tokens(z, j)
        if p == int(z["verify_prompt"][j]) and np.array_equal(toks, ref):
            continue
        if p != int(z["verify_prompt"][j]) or len(toks) != len(ref) or not np.array_equal(toks[:-1], ref[:-1]):
            return "differs", (j, "the accepted paths differ")
        res = record_residual(z, j, meta["vocab"])
        if res is None:
            return "differs", (j, "no residual recorded")
        u = int(z["bonus_u24"][p * STEPS_PER_PROMPT + int(z["verify_step"][j])])
        return "boundary", (j, cdf_interval_distance(res, int(toks[-1]), u))
    if len(log) != n:
        return "differs", (len(log), "fewer verify calls than the reference made")
    return "identical", None


def run_dropin(z, meta, device="cpu", growmap=None, code=None):
    """The same compiled lines on sequoia_amd.dropin (oracle adapter on the CPU; the HIP library on a GPU).  `growmap`: the
    growmap dict to hand to `torch.load`'s place when the reference checkout (and its .pt file) is absent."""
    import sequoia_amd.dropin as dropin
    dropin.install(force=True)
    try:
        from Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG
        from Tree.SpecTree import SpecTree
        from utils import cuda_graph_for_residual, cuda_graph_for_sampling_without_replacement
        code = code or compile_harness()
        seed = meta["seed"]
        n_prompts = meta["n_prompts"]
        batches = [dict(input_ids=torch.from_numpy(z[f"prompt{i}/input_ids"]), labels=torch.from_numpy(z[f"prompt{i}/labels"]))
                   for i in range(n_prompts)]
        u24 = z["bonus_u24"]
        log, state = [], dict(prompt=-1)

        sd_d, sd_t, checks = seeded_pair(seed, meta.get("gain"), meta.get("share"), meta.get("branch"))
        assert checks == meta["weight_checksums"], "seeded weights differ from the reference run's (torch CPU generator drift)"
        sds = dict(draft=sd_d, target=sd_t)
        hidden, inter, layers, heads, kv = meta["dims"]
        cfg = dict(vocab_size=meta["vocab"], hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                   num_attention_heads=heads, num_key_value_heads=kv, max_position_embeddings=2048)

        def factory(cls, which):
            def build(max_length, model_name_or_path, dtype=torch.float16, device_="cuda:0", **kw):
                assert model_name_or_path == which
                kw.pop("device", None)
                return cls(max_length=max_length, model_name_or_path=dict(state_dict=sds[which], config=cfg), dtype=dtype,
                           device=device)
            return build

        class SpySpecTree(SpecTree):
            def __init__(self, **kw):
                kw["device"] = device
                state["prompt"] += 1
                p = state["prompt"]
                torch.manual_seed(noise_seed(seed, p))
                super().__init__(bonus_uniforms=[int(x) for x in u24[p * STEPS_PER_PROMPT:(p + 1) * STEPS_PER_PROMPT]],
                                 commit_order="reference", **kw)

            def verify(self, benchmark=False):
                out = super().verify(benchmark=benchmark)
                log.append((state["prompt"], out[0].clone().cpu().numpy()))
                return out

        args = harness_args(seed, meta["growmap"])
        ns = dict(torch=torch, time=time, np=np, random=__import__("random"), tqdm=lambda it, total=None: it, print=print,
                  DataLoader=object, args=args, dataloader=batches, SpecTree=SpySpecTree,
                  GraphInferenceEngine=factory(GraphInferenceEngine, "draft"),
                  GraphInferenceEngineTG=factory(GraphInferenceEngineTG, "target"), OffloadEngine=None,
                  cuda_graph_for_residual=cuda_graph_for_residual,
                  cuda_graph_for_sampling_without_replacement=cuda_graph_for_sampling_without_replacement)
        if growmap is not None:                    # no reference checkout: the growmap file's CONTENT arrives as data
            real_load = torch.load
            torch.load = lambda path, *a, **k: growmap if path == args.growmap else real_load(path, *a, **k)
        try:
            if str(device).startswith("cuda"):
                exec(code["defs"], ns)
                ns["setup_seed"](args.seed)
                exec(code["setup"], ns)
                exec(code["call"], ns)
                value = ns["_harness_result"]
            else:
                value = run_harness(code, ns, seed)
        finally:
            if growmap is not None:
                torch.load = real_load
        return value, log
    finally:
        dropin.uninstall()


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "reference":
        sys.path.insert(0, REPO)
        run_reference(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 11, sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        raise SystemExit(__doc__)
