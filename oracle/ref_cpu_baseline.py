"""The reference's own CPU path beside the port that bench.py times as `cpu_baseline` (test infrastructure; runs only
where /root/reference exists).  Same weights (the synthetic calibrated pair of sequoia_amd/synthetic.py, exported to HF
parameter names), same prompt, same noise seed, same growmap, same number of speculation steps:

    python oracle/ref_cpu_baseline.py [--config B] [--steps 3] [--out profiles/r02_cpu_reference_vs_port.json]

The reference (Tree/SpecTree.py, Engine/*, utils.py imported from /root/reference with the outside shims of
oracle/gen_golden.py; `multinomial` replaced by the recorded-uniform inverse CDF like the traces) is timed per step
exactly like bench.cpu_baseline times the port; the committed JSON states both, the tokens each committed and the
ratio -- the fidelity of the port as a stand-in for the reference on boxes that do not have the reference."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def hf_state_dict(w):
    """LlamaWeights (fused qkv / gate_up) -> HF parameter names of the reference's LlamaForCausalLM_*."""
    d = w.dims
    hd = d.head_dim
    nq, nkv = d.num_attention_heads * hd, d.num_key_value_heads * hd
    inter = w.layers[0].w_down.shape[1]
    sd = {"model.embed_tokens.weight": w.embed, "model.norm.weight": w.norm, "lm_head.weight": w.lm_head}
    for i, lw in enumerate(w.layers):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = lw.wqkv[:nq]
        sd[p + "self_attn.k_proj.weight"] = lw.wqkv[nq:nq + nkv]
        sd[p + "self_attn.v_proj.weight"] = lw.wqkv[nq + nkv:]
        sd[p + "self_attn.o_proj.weight"] = lw.wo
        sd[p + "mlp.gate_proj.weight"] = lw.w_gate_up[:inter]
        sd[p + "mlp.up_proj.weight"] = lw.w_gate_up[inter:]
        sd[p + "mlp.down_proj.weight"] = lw.w_down
        sd[p + "input_layernorm.weight"] = lw.ln1
        sd[p + "post_attention_layernorm.weight"] = lw.ln2
    return sd


def reference_engine(R, outer, inner, model_cls, weights, M):
    """gen_golden.make_engine without the seeded init: parameters are allocated in fp16, uninitialised, and loaded."""
    d = weights.dims
    cfg = R["Cfg"](vocab_size=d.vocab_size, hidden_size=d.hidden_size, intermediate_size=weights.layers[0].w_down.shape[1],
                   num_hidden_layers=d.num_hidden_layers, num_attention_heads=d.num_attention_heads,
                   num_key_value_heads=d.num_key_value_heads, max_position_embeddings=2048)
    cfg.rope_scaling = None
    e = outer.__new__(outer)
    e.device = "cpu"; e.dtype = torch.float16; e.max_length = M; e.callables = {}; e.mempool = None
    n = inner.__new__(inner)
    n.device = "cpu"; n.dtype = torch.float16; n.max_length = M
    # parameters on the meta device (no 7B-parameter initialisation), replaced by the port's tensors; the rotary tables
    # (buffers the modules compute in their constructors) are rebuilt on the CPU afterwards
    with torch.device("meta"):
        model = model_cls(cfg)
    missing, unexpected = model.load_state_dict(hf_state_dict(weights), strict=False, assign=True)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    for mod in list(model.modules()):
        if hasattr(mod, "_init_rope"):
            mod._init_rope()
    model = model.to(torch.float16)        # like gen_golden.make_engine: fp32 tables built, then cast with the model
    for name, buf in model.named_buffers():
        assert buf.device.type == "cpu", f"buffer {name} still on {buf.device}"
    n.model = model.eval(); n.model_config = cfg
    n.kv_cache = R["KV"](config=cfg, max_length=M, device="cpu", dtype=torch.float16)
    e.engine = n
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r02_cpu_reference_vs_port.json"))
    args = ap.parse_args()
    import gen_golden as GG
    from oracle import ops_np
    import bench
    from sequoia_amd.harness import MODELS, build, load_prompts
    cfg = dict(MODELS[args.config])
    M, T = cfg["M"], 0.6
    t0 = time.perf_counter()
    draft_n, target_n, gm = build(cfg, "cpu", "calibrated")
    print(f"weights built in {time.perf_counter() - t0:.0f} s", flush=True)
    wd, wt = draft_n.engine.model.weights, target_n.engine.model.weights

    # ---- the port (what bench.py's cpu_baseline leg runs) ------------------------------------------------------------
    port = bench.cpu_baseline(cfg, args.steps, "calibrated", engines=(draft_n, target_n, gm))
    print("port     :", port["sample"], flush=True)
    # the same loop on the numpy oracle ops (the checker of the GPU tests): slower, but rounding point for rounding point
    # the reference's arithmetic -- its committed tokens must equal the reference's
    draft_n.clear_kv(); target_n.clear_kv()
    port_np = bench.cpu_baseline(cfg, args.steps, "calibrated", engines=(draft_n, target_n, gm), numpy_ops=True)
    print("numpy    :", port_np["step_seconds"], flush=True)
    del draft_n, target_n

    # ---- the reference itself ------------------------------------------------------------------------------------------
    R = GG.import_reference()
    RU = R["RU"]
    draft = reference_engine(R, R["GIE"], R["IE"], R["MM"].LlamaForCausalLM_FI, wd, M)
    target = reference_engine(R, R["GIETG"], R["IETG"], R["MM"].LlamaForCausalLM_TG, wt, M)
    g = gm.to_reference_dict()
    n_levels = len(g["roots"]) - 1
    samp = {i: (lambda k: lambda lg, rnd: RU.sampling_without_replacement(lg, rnd, k, T))(max(g["branches"][i]))
            for i in range(n_levels)}
    gidx = {i: torch.cat([torch.arange(b) + j * max(g["branches"][i]) for j, b in enumerate(g["branches"][i])])
            for i in range(n_levels)}
    prefix = torch.tensor(load_prompts()[0][:128], dtype=torch.long)
    box = {"i": 0, "u": None}
    orig_multinomial = torch.Tensor.multinomial

    def fake_multinomial(self, num_samples=1, replacement=False, generator=None):      # the port's bonus draw
        return torch.tensor([ops_np.inverse_cdf(self.detach().numpy(), int(box["u"][box["i"] % len(box["u"])]))])
    torch.Tensor.multinomial = fake_multinomial
    try:
        torch.manual_seed(17)
        tree = R["SpecTree"](prefix=prefix, device="cpu", temperature=T, top_p=1.0, draft_kv_len=0, target_kv_len=0,
                             draft_model_engine=draft, target_model_engine=target, max_length=M, max_target_seq=M,
                             grow_map=g, attn_mask=torch.full((M, M), torch.finfo(torch.float16).min, dtype=torch.float16),
                             sequence=None, new_tokens_buffer=None, parents_buffer=None, position_ids=torch.zeros(M).long(),
                             residual_graph=RU.get_residual, sampling_callables=samp, sample_gather_indices=gidx,
                             vocab_size=32000)
        # the port draws its bonus uniforms right after the noise, from the same generator (NativeTree.__init__)
        box["u"] = [int(x) for x in torch.randint(0, 1 << 24, (M + 1,))]
        cur, step_s, step_tok = len(prefix), [], []
        for s in range(max(2, args.steps)):
            box["i"] = s
            t1 = time.perf_counter()
            tree.construct_grow_map()
            valid, _, _, term = tree.verify()
            step_s.append(time.perf_counter() - t1)
            step_tok.append(valid.shape[0] - cur)
            cur = valid.shape[0]
            print(f"reference step {s}: {step_s[-1]:.1f} s, +{step_tok[-1]} tokens", flush=True)
            if term:
                break
        ref_tokens = valid[:cur].tolist()
    finally:
        torch.Tensor.multinomial = orig_multinomial
    n_steady, steady_s = len(step_s) - 1, sum(step_s[1:])
    ref = dict(kind="reference", cores=torch.get_num_threads(), prefill_step_s=step_s[0],
               step_seconds=[round(x, 3) for x in step_s], step_tokens=step_tok,
               steps_per_s=n_steady / steady_s if n_steady else None,
               value=sum(step_tok[1:]) / steady_s if n_steady else None, unit="tokens/s")
    def agree(toks, steps_tok):
        """Number of leading speculation steps whose committed tokens equal the reference's."""
        n, ok = len(prefix), 0
        for t_ref, t_p in zip(step_tok, steps_tok):
            if t_ref != t_p or ref_tokens[n:n + t_ref] != toks[n:n + t_ref]:
                break
            n += t_ref; ok += 1
        return ok
    same = agree(port.get("tokens"), port["step_tokens"])
    same_np = agree(port_np.get("tokens"), port_np["step_tokens"])
    rec = dict(config=args.config, workload=f"{cfg['draft']} -> {cfg['target']}, growmap {cfg['growmap']}, T=0.6, prompt 0",
               host=dict(cores=os.cpu_count(), torch_threads=torch.get_num_threads(), torch=torch.__version__),
               reference=ref, port={k: v for k, v in port.items() if k != "tokens"},
               numpy_oracle_port=dict(step_seconds=port_np["step_seconds"], step_tokens=port_np["step_tokens"]),
               steps_with_identical_committed_tokens=dict(numpy_oracle_port=same_np, torch_cpu_port=same, of=len(step_tok)),
               port_over_reference_steady=(port["steps_per_s"] / ref["steps_per_s"]) if n_steady and port["steps_per_s"] else None,
               port_over_reference_prefill_step=ref["prefill_step_s"] / port["prefill_step_s"])
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
