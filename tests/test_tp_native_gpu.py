"""Tensor-parallel target on the native kernels, exercised on ONE GPU: a world-size-1 RCCL group with the hooks forced
on (SEQUOIA_TP_FORCE_HOOKS) runs every all-reduce / all-gather call site, the split-K slab -> rows -> all-reduce ->
residual path of forward_ts, the hipGraph capture of the collectives, and the exclusive weight mode (fragment-major
images as the only copy, prefill as row chunks) -- and must reproduce the single-GPU engine bit for bit."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nccl_world1():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    yield
    if created:
        dist.destroy_process_group()


def _engines(monkeypatch, exclusive=False):
    from sequoia_amd.Engine import tp_engine
    from sequoia_amd.Engine.Engine import GraphInferenceEngineTG
    from sequoia_amd.Engine.tp_engine import TPEngine
    spec = "random:JackFram/llama-160m:seed=3:gain=12"
    M = 384
    ref = GraphInferenceEngineTG(max_length=M, model_name_or_path=spec, dtype=torch.float16, device=DEV)
    monkeypatch.setattr(tp_engine, "FORCE_HOOKS", True)
    if exclusive:
        monkeypatch.setenv("SEQUOIA_TS_EXCLUSIVE", "1")
    tp = TPEngine(max_length=M, model_name_or_path=spec, dtype=torch.float16, device=DEV)
    assert tp.engine.model.reduce_fn is not None and tp.engine.model.ts is not None
    assert tp.engine.model.ts.exclusive == exclusive
    return ref, tp


@pytest.mark.parametrize("exclusive", [False, True])
def test_tp_hooks_on_native_path_match_single_gpu(nccl_world1, monkeypatch, exclusive):
    from sequoia_amd.Engine.Llama_modules import TreeContext
    from sequoia_amd.growmap import GrowMap
    ref, tp = _engines(monkeypatch, exclusive)
    g = GrowMap.load("64x2-tree")
    bm = g.device_tensors(DEV)["bitmask"]
    n = g.size
    torch.manual_seed(0)
    ids = torch.randint(3, 32000, (1, 200 + n - 1), device=DEV)
    pos = torch.arange(ids.shape[1], device=DEV)
    outs = []
    for eng in (ref, tp):
        if eng is tp and not exclusive:      # same launch plans (129 rows have no shipped plan: each set would time its own)
            tp.engine.model.ts._plans.update(ref.engine.model.ts._plans)
        # 200-token prefill (general GEMM path, or row chunks in exclusive mode), then the 129-node tree (9 row tiles)
        a = eng.inference(input_ids=ids[:, :199], storage_ids=pos[:199], position_ids=pos[None, :199], attn_mask=None,
                          tree=TreeContext(0, 200, n, bm, 199, contiguous_slots=True))
        b = eng.inference(input_ids=ids[:, 199:], storage_ids=pos[199:], position_ids=pos[None, 199:], attn_mask=None,
                          tree=TreeContext(199, 200, n, bm, 199 + n, contiguous_slots=True))
        outs.append((a.float(), b.float()))
    assert tp.engine.collectives > 0
    if exclusive:      # chunked prefill: different accumulation order in the projections -> a few fp16 ulps
        assert (outs[0][0] - outs[1][0]).abs().max() < 6e-2
        assert (outs[0][1] - outs[1][1]).abs().max() < 6e-2
    else:
        assert torch.equal(outs[0][0], outs[1][0])
        assert torch.equal(outs[0][1], outs[1][1])       # slab -> rows -> all-reduce -> add == fused slab add, bit for bit


def test_tp_forward_replays_from_a_graph_with_its_collectives(nccl_world1, monkeypatch):
    from sequoia_amd.Engine.Llama_modules import TreeContext
    from sequoia_amd.growmap import GrowMap
    ref, tp = _engines(monkeypatch)
    g = GrowMap.load("64x2-tree")
    bm = g.device_tensors(DEV)["bitmask"]
    n = g.size
    tp.initialize_cuda_graph([n], tree_bitmask=bm, n_tree=n)
    torch.manual_seed(1)
    ids = torch.randint(3, 32000, (1, 40 + n - 1), device=DEV)
    pos = torch.arange(ids.shape[1], device=DEV)
    outs = []
    for use_graph in (False, True):
        tp.clear_kv()
        tp.inference(input_ids=ids[:, :39], storage_ids=pos[:39], position_ids=pos[None, :39], attn_mask=None,
                     tree=TreeContext(0, 40, n, bm, 39, contiguous_slots=True))
        run = tp.graph_inference if use_graph else tp.inference
        outs.append(run(input_ids=ids[:, 39:], storage_ids=pos[39:], position_ids=pos[None, 39:], attn_mask=None,
                        tree=TreeContext(39, 40, n, bm, 39 + n, contiguous_slots=True)).clone())
    assert torch.equal(outs[0], outs[1])
