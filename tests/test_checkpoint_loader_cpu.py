"""Checkpoint loading (the reference's engines start from `from_pretrained`: Engine/Engine.py:18,81,
Engine/offload_engine.py:268-300): a HF-style directory -- config.json + two safetensors shards, with and without
`lm_head.weight` (tied embeddings), with and without the index json -- written from a trace's state dict must give the
weights and the logits of the in-memory `state_dict` path, single-process and tensor-parallel at world size 2 (gloo), and
every rank must read only its own shard from disk."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_trace
from helpers import dims_dict, trace_state_dicts

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _weights_equal(a, b):
    assert torch.equal(a.embed, b.embed) and torch.equal(a.norm, b.norm) and torch.equal(a.lm_head, b.lm_head)
    assert len(a.layers) == len(b.layers)
    for la, lb in zip(a.layers, b.layers):
        for f in ("ln1", "wqkv", "wo", "ln2", "w_gate_up", "w_down"):
            assert torch.equal(getattr(la, f), getattr(lb, f)), f


def _tied(sd):
    sd = dict(sd)
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    return sd


@pytest.mark.parametrize("tied,index", [(False, True), (True, True), (False, False)])
def test_directory_equals_state_dict_single_process(tmp_path, tied, index):
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    from sequoia_amd.Engine.checkpoint import save_checkpoint_dir
    from sequoia_amd.Engine.Engine import GraphInferenceEngineTG
    from sequoia_amd.Engine.Llama_model import load_weights
    z, meta = load_trace("E_64x2")                         # GQA 4:1 target, tiny dims, weights stored in the trace
    _, sd = trace_state_dicts(z, meta)
    if tied:
        sd = _tied(sd)
    cfg = dims_dict(meta["target_dims"], meta["vocab"])
    d = save_checkpoint_dir(sd, cfg, str(tmp_path / "ckpt"), n_shards=2, tie_lm_head=tied, index=index)
    files = sorted(os.listdir(d))
    assert sum(f.endswith(".safetensors") for f in files) == 2 and ("model.safetensors.index.json" in files) == index
    w_dir = load_weights(d, torch.float16, "cpu")
    w_sd = load_weights(dict(state_dict=sd, config=cfg), torch.float16, "cpu")
    _weights_equal(w_dir, w_sd)
    # the engines' own path: model_name_or_path = <directory>; logits of a forward equal the state_dict engine's
    ops.set_ops_for_testing(OracleOps())
    try:
        M = meta["M"]
        e_dir = GraphInferenceEngineTG(max_length=M, model_name_or_path=d, dtype=torch.float16, device="cpu")
        e_sd = GraphInferenceEngineTG(max_length=M, model_name_or_path=dict(state_dict=sd, config=cfg), dtype=torch.float16, device="cpu")
        ids = torch.from_numpy(z["prompt"])[None]
        n = ids.shape[1]
        pos = torch.arange(n)
        mask = torch.triu(torch.full((n, n), torch.finfo(torch.float16).min, dtype=torch.float16), 1)[None, None]
        a = e_dir.inference(input_ids=ids, storage_ids=pos, position_ids=pos[None], attn_mask=mask)
        b = e_sd.inference(input_ids=ids, storage_ids=pos, position_ids=pos[None], attn_mask=mask)
        assert torch.equal(a, b) and torch.isfinite(a.float()).all()
    finally:
        ops.set_ops_for_testing(None)


def test_missing_tensor_and_missing_files_fail_loudly(tmp_path):
    from sequoia_amd.Engine.checkpoint import save_checkpoint_dir
    from sequoia_amd.Engine.Llama_model import load_weights
    z, meta = load_trace("A_2chain")
    sd, _ = trace_state_dicts(z, meta)
    cfg = dims_dict(meta["draft_dims"], meta["vocab"])
    bad = {k: v for k, v in sd.items() if k != "model.layers.1.mlp.down_proj.weight"}
    d = save_checkpoint_dir(bad, cfg, str(tmp_path / "bad"), n_shards=2)
    with pytest.raises(KeyError):
        load_weights(d, torch.float16, "cpu")
    os.makedirs(tmp_path / "empty")
    with open(tmp_path / "empty" / "config.json", "w") as f:
        f.write("{}")
    with pytest.raises(FileNotFoundError):
        load_weights(str(tmp_path / "empty"), torch.float16, "cpu")


def _tp_worker(rank, world, port, ckpt, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_trace
    from helpers import dims_dict, trace_state_dicts
    from oracle.ops_adapter import OracleOps
    from sequoia_amd import ops
    from sequoia_amd.Engine.Llama_model import load_weights
    from sequoia_amd.Engine.offload_engine import OffloadEngine
    ops.set_ops_for_testing(OracleOps())
    z, meta = load_trace("E_64x2")
    _, sd = trace_state_dicts(z, meta)
    cfg = dims_dict(meta["target_dims"], meta["vocab"])
    M = meta["M"]
    # shard-aware: this rank's weights from the directory == its shard of the state dict, and it read about 1 / world of the
    # projection bytes from disk (embedding and norms are replicated)
    w_dir = load_weights(ckpt, torch.float16, "cpu", tp_world=world, tp_rank=rank)
    w_sd = load_weights(dict(state_dict=sd, config=cfg), torch.float16, "cpu", tp_world=world, tp_rank=rank)
    for la, lb in zip(w_dir.layers, w_sd.layers):
        for f in ("wqkv", "wo", "w_gate_up", "w_down"):
            assert torch.equal(getattr(la, f), getattr(lb, f)), f
    assert torch.equal(w_dir.lm_head, w_sd.lm_head) and w_dir.lm_head.shape[0] == meta["vocab"] // world
    kv_world = min(world, meta["target_dims"][4])          # (1 KV head here: K / V are replicated, like Llama-2-70b beyond TP = 8)
    want = 0
    for k, v in sd.items():
        if "rotary" in k or "inv_freq" in k:
            continue
        nbytes = int(np.prod(v.shape)) * 2
        if any(t in k for t in ("q_proj", "gate_proj", "up_proj", "o_proj", "down_proj", "lm_head")):
            nbytes //= world
        elif "k_proj" in k or "v_proj" in k:
            nbytes //= kv_world
        want += nbytes
    assert w_dir.checkpoint_bytes_read == want, (w_dir.checkpoint_bytes_read, want)
    # the tensor-parallel engine built from the directory: same logits as from the state dict, on every rank
    e_dir = OffloadEngine(max_length=M, model_name_or_path=ckpt, dtype=torch.float16, device="cpu")
    e_sd = OffloadEngine(max_length=M, model_name_or_path=dict(state_dict=sd, config=cfg), dtype=torch.float16, device="cpu")
    assert e_dir.world == world
    ids = torch.from_numpy(z["prompt"])[None]
    n = ids.shape[1]
    pos = torch.arange(n)
    mask = torch.triu(torch.full((n, n), torch.finfo(torch.float16).min, dtype=torch.float16), 1)[None, None]
    a = e_dir.inference(input_ids=ids, storage_ids=pos, position_ids=pos[None], attn_mask=mask)
    b = e_sd.inference(input_ids=ids, storage_ids=pos, position_ids=pos[None], attn_mask=mask)
    assert torch.equal(a, b) and a.shape[-1] == meta["vocab"]
    np.save(os.path.join(out_dir, f"r{rank}.npy"), a.float().numpy())
    dist.destroy_process_group()


def test_directory_loads_shard_aware_at_world_2(tmp_path):
    from sequoia_amd.Engine.checkpoint import save_checkpoint_dir
    z, meta = load_trace("E_64x2")
    _, sd = trace_state_dicts(z, meta)
    ckpt = save_checkpoint_dir(sd, dims_dict(meta["target_dims"], meta["vocab"]), str(tmp_path / "ckpt"), n_shards=2)
    port = 29500 + ((os.getpid() + 777) % 2000)
    mp.spawn(_tp_worker, args=(2, port, ckpt, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(a, b)                    # gathered logits are identical on both ranks
