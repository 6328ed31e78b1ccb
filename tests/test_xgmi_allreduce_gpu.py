"""The two-shot all-reduce over peer-mapped buffers (csrc/allreduce.hip, Engine/xgmi_allreduce.py) with two ranks on the
ONE GPU of the test box: the workspaces are hipIpc-exported / -opened across the two processes exactly as they are across
GPUs, every flag and every peer store of the protocol runs, the transport of the setup is gloo.  Checked: sums against
gloo's all-reduce for sizes from 8 elements to the [129, 8192] message of the 70B verify, bit-identical rows on both ranks,
back-to-back calls without host synchronisation and with skewed arrival (epoch protocol), replay from a hipGraph, the
tensor-parallel target engine on it (every step of the reference's trace), and a bounded spin when the peer never shows up.
(N real GPUs: bench.py --gpus N reports `allreduce` for configuration E.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


@pytest.fixture(params=["device", "host"], autouse=True)
def ws_mode(request, monkeypatch):
    """Every test of this file runs twice: with the workspaces in uncached DEVICE memory shared through hipIpc (the
    production form; on this one-GPU box both "peers" then share one HBM and one L2 fabric), and with the workspaces in
    fine-grained HOST memory shared between the processes (SEQUOIA_AR_WS=host): every payload store, flag store and flag
    poll then leaves the device over PCIe, so a missing system-scope fence, a flag overtaking its payload or a cached
    flag read has a real chance to show -- the closest a one-GPU box gets to peer memory behind an xGMI link."""
    monkeypatch.setenv("SEQUOIA_AR_WS", request.param)
    return request.param


def _ar_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, HERE)
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    torch.cuda.set_device(0)
    from sequoia_amd.Engine.xgmi_allreduce import XgmiAllReduce
    ar = XgmiAllReduce.create(None, dev, max_elems=144 * 8192, max_gather_elems=144 * 32000)
    assert ar is not None, "xGMI all-reduce could not be set up on this box (see stderr)"
    res = dict(sizes=[], graph=False, burst=False)
    gen = torch.Generator(device="cpu")
    # 1. sizes, against gloo's sum (fp32 reference of the fp16 inputs), bit-identical across ranks
    for i, n in enumerate([8, 64, 2056, 34 * 768, 128 * 4096, 129 * 8192, 144 * 8192]):
        gen.manual_seed(77 * i + rank)
        x = (torch.randn(n, generator=gen) * 3).half().to(dev)
        want = x.float()
        dist.all_reduce(want)
        got = ar(x.clone())
        torch.cuda.synchronize()
        assert ar.status() == 0
        # one rounding of an fp32 sum: within half an fp16 ulp of the exact sum
        err = (got.float() - want).abs()
        tol = want.abs().clamp(min=1.0) * 2.0 ** -10
        assert bool((err <= tol).all()), f"n = {n}: max err {float(err.max())}"
        bits = got.view(torch.int16).to(torch.int32)
        lo, hi = bits.clone(), bits.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), f"n = {n}: ranks hold different bits"
        res["sizes"].append(n)
    # 1b. all-gather of column-parallel logits: [rows, v] per rank -> [rows, W v], exact, interleaved with all-reduces
    vs = 32000 // world
    for i, (rows, v) in enumerate([(1, vs), (129, vs), (34, 8), (144, vs)]):
        gen.manual_seed(400 + 3 * i + rank)
        sl = torch.randn(rows, v, generator=gen).half().to(dev)
        parts = [torch.empty_like(sl) for _ in range(world)]
        dist.all_gather(parts, sl)
        y = torch.ones(129 * 8192, dtype=torch.float16, device=dev)
        ar(y)                                                   # an all-reduce right before and right after
        got = ar.gather_cols(sl)
        ar(y)
        got2 = ar.gather_cols(sl)                               # back to back: the "read" handshake
        torch.cuda.synchronize()
        assert ar.status() == 0
        assert torch.equal(got, torch.cat(parts, dim=1)) and torch.equal(got2, got), f"gather {rows} x {v}"
        assert float(y[0]) == world * world
    # 1c. the all-reduce fed by split-K partials (fp32 slabs) == fp16 rounding of the slab sum, then the plain all-reduce
    for i, (rows, hid, splits) in enumerate([(129, 8192, 2), (34, 768, 3), (1, 4096, 4)]):
        gen.manual_seed(900 + 7 * i + rank)
        slab = torch.randn(splits, rows * hid, generator=gen).to(dev)
        acc = slab[0].clone()
        for sidx in range(1, splits):
            acc += slab[sidx]
        ref = acc.half()
        ar(ref)
        out = torch.empty(rows * hid, dtype=torch.float16, device=dev)
        ar.reduce_slabs(slab.reshape(-1), splits, out)
        torch.cuda.synchronize()
        assert ar.status() == 0 and torch.equal(out, ref), f"slab all-reduce {rows} x {hid} x {splits}"
    # 1d. all-reduce + skip connection + RMSNorm in one launch (row-aligned two-shot) == all-reduce, then sq_add_rmsnorm_*:
    #     same bits in the residual stream and in the normalised operand (row-major and fragment-major), from fp16 rows and
    #     from split-K slabs, for row counts that do not divide by the world (ranks without rows included), interleaved
    #     with plain all-reduces (shared epochs) and called back to back
    from sequoia_amd.ops import HipOps
    hip = HipOps()
    for i, (rows, hid, splits, frag) in enumerate([(129, 8192, 0, True), (129, 8192, 2, True), (1, 4096, 0, True), (3, 768, 3, False),
                                                    (34, 768, 0, True), (144, 8192, 2, False), (19, 5120, 0, True), (5, 8192, 4, True)]):
        gen.manual_seed(1300 + 11 * i + rank)
        if splits:
            part = torch.randn(splits, rows * hid, generator=gen).to(dev)
            acc = part[0].clone()
            for sidx in range(1, splits):
                acc += part[sidx]
            mine = acc.half().reshape(rows, hid)
        else:
            part = torch.randn(rows, hid, generator=gen).half().to(dev)
            mine = part.clone()
        gen.manual_seed(1300 + 11 * i)                                   # replicated: residual stream and norm weight
        x0 = torch.randn(rows, hid, generator=gen).half().to(dev)
        g = (1.0 + 0.1 * torch.randn(hid, generator=gen)).half().to(dev)
        red = ar(mine.clone().reshape(-1)).reshape(rows, hid)
        x_ref = x0.clone()
        o_ref = torch.zeros(hip.frag_shape(rows, hid) if frag else (rows, hid), dtype=torch.float16, device=dev)
        (hip.add_rmsnorm_frag if frag else hip.add_rmsnorm)(red, x_ref, x_ref, g, o_ref, 1e-5)
        for rep in range(2):
            x = x0.clone()
            out = torch.zeros_like(o_ref)
            assert ar.fits_rows(rows, hid)
            ar.reduce_add_rmsnorm(part.reshape(-1) if splits else part, x, g, out, 1e-5, frag, splits=splits)
            ar(torch.ones(64, dtype=torch.float16, device=dev))
            torch.cuda.synchronize()
            assert ar.status() == 0
            assert torch.equal(x, x_ref), f"residual stream {rows} x {hid} splits {splits}"
            assert torch.equal(out, o_ref), f"normalised rows {rows} x {hid} splits {splits} frag {frag}"
    assert not ar.fits_rows(145 * world, 8192)
    # 2. a burst of back-to-back calls without host synchronisation, arrival skewed (one rank is kept busy / asleep):
    #    call k + 1 of the fast rank must not disturb call k of the slow one (per-block epochs, areas reused every call)
    n = 129 * 8192
    xs, wants = [], []
    for k in range(24):
        gen.manual_seed(5000 + 31 * k + rank)
        x = (torch.randn(n, generator=gen)).half().to(dev)
        w = x.float(); dist.all_reduce(w)
        xs.append(x); wants.append(w)
    torch.cuda.synchronize(); dist.barrier()
    big = torch.randn(4096, 4096, device=dev)
    for k in range(24):
        if k % 5 == rank % 5:
            time.sleep(0.02)                       # host-side skew
        if k % 7 == (3 * rank) % 7:
            big = big @ big * 1e-4                 # device-side skew: this rank's kernel is queued behind a GEMM
        ar(xs[k])
    torch.cuda.synchronize()
    assert ar.status() == 0
    for k in range(24):
        err = (xs[k].float() - wants[k]).abs()
        assert bool((err <= wants[k].abs().clamp(min=1.0) * 2.0 ** -10).all()), f"burst call {k}"
    res["burst"] = True
    # 3. three all-reduces inside one hipGraph, replayed on fresh data (the launch has no per-call argument)
    bufs = [torch.zeros(129 * 1024, dtype=torch.float16, device=dev) for _ in range(3)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for b in bufs:
            ar(b)
        s.synchronize()
    torch.cuda.current_stream().wait_stream(s)
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for b in bufs:
            ar(b)
    for rep in range(4):
        wants = []
        for j, b in enumerate(bufs):
            gen.manual_seed(9000 + 10 * rep + j + 100 * rank)
            v = torch.randn(b.numel(), generator=gen).half().to(dev)
            b.copy_(v)
            w = v.float(); dist.all_reduce(w); wants.append(w)
        torch.cuda.synchronize(); dist.barrier()
        g.replay()
        torch.cuda.synchronize()
        assert ar.status() == 0
        for b, w in zip(bufs, wants):
            assert bool(((b.float() - w).abs() <= w.abs().clamp(min=1.0) * 2.0 ** -10).all()), f"graph replay {rep}"
    res["graph"] = True
    np.save(os.path.join(out_dir, f"ar{rank}.npy"), np.array([len(res["sizes"]), int(res["burst"]), int(res["graph"]), ar.calls]))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_two_shot_allreduce_ranks_on_one_gpu(world, tmp_path):
    """world 4: three peers per rank -- the staggered peer order, the rank-order sum over more than two copies and the
    per-(peer, block) flags of a larger group, still on the one GPU."""
    port = 33100 + (os.getpid() % 1500) + world
    mp.spawn(_ar_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        sizes, burst, graph, calls = np.load(tmp_path / f"ar{r}.npy")
        assert sizes == 7 and burst == 1 and graph == 1 and calls > 40


def _lonely_worker(rank, world, port, out_dir):
    """Rank 1 never launches: rank 0's kernel must give up after its bounded spin and say so."""
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from sequoia_amd.Engine.xgmi_allreduce import XgmiAllReduce
    ar = XgmiAllReduce.create(None, "cuda:0", max_elems=4096, self_check=False)
    assert ar is not None
    if rank == 0:
        from sequoia_amd.Engine import xgmi_allreduce as XA
        x = torch.ones(4096, dtype=torch.float16, device="cuda:0")
        ar(x)
        torch.cuda.synchronize()                         # returns: the spin is bounded
        # ... and the timeout does not pass silently: the kernel stored its bits into the pinned fault word (a plain
        # system-scope store: the word holds the bits of the LAST wait that gave up; any non-zero value raises), which the
        # speculation loop tests at every step (raise_on_fault) -- no device read involved.  The workspace's own status
        # word accumulates every phase that timed out; once it is set the later waits of the job look once, not spin.
        fault = int(ar.fault[0])
        raised = 0
        try:
            XA.raise_on_fault()
        except XA.XgmiCollectiveTimeout:
            raised = 1
        # a caller that never polls raise_on_fault() is stopped by the NEXT collective on the dead workspace
        refused = 0
        try:
            ar(x)
        except XA.XgmiCollectiveTimeout:
            refused = 1
        np.save(os.path.join(out_dir, "lonely.npy"), np.array([ar.status(), fault, raised, refused]))
        ar.clear_fault()
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


def test_missing_peer_times_out_instead_of_hanging(tmp_path, monkeypatch):
    monkeypatch.setenv("SEQUOIA_AR_SPIN_LIMIT", "200000")        # ~0.3 s instead of the production bound of a few seconds
    port = 34700 + (os.getpid() % 1500)
    mp.spawn(_lonely_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    status, fault, raised, refused = (int(v) for v in np.load(tmp_path / "lonely.npy"))
    assert status & 1 == 1 and fault != 0 and raised == 1 and refused == 1


@pytest.mark.parametrize("name,world,fused_norm", [("E_64x2", 2, "1"), ("E_70b_w2", 2, "1"), ("E_64x2", 4, "1"), ("demo4", 4, "1"),
                                                   ("E_64x2", 2, "0")])
def test_tp_on_the_xgmi_allreduce_matches_reference_trace(name, world, fused_norm, tmp_path, monkeypatch):
    """The tensor-parallel target (KV-head split, vocabulary-parallel lm_head) with its row-parallel projections reduced
    by the xGMI kernel: every step of the reference's trace on both ranks (tests/test_tp_gloo_cpu.py::_worker asserts
    identical decisions on both ranks)."""
    from test_tp_gloo_cpu import _worker
    monkeypatch.setenv("SEQUOIA_TP_ALLREDUCE", "xgmi")
    monkeypatch.setenv("SEQUOIA_TP_REQUIRE_XGMI", "1")
    monkeypatch.setenv("SEQUOIA_TP_FUSED_NORM", fused_norm)      # "1": all-reduce + skip + RMSNorm in one kernel; "0": three launches
    port = 36300 + (os.getpid() % 1500) + world
    mp.spawn(_worker, args=(world, port, name, str(tmp_path), "cuda:0"), nprocs=world, join=True)
    from conftest import load_trace
    n_steps = int(load_trace(name)[0]["n_steps"])
    res = [tuple(np.load(tmp_path / f"r{r}.npy")) for r in range(world)]
    assert len(set(res)) == 1, res                           # every rank: the same outcome
    matched, diverged = res[0]
    # all steps token-identical to the reference, or (4-way sharded sums round differently from the reference's unsharded
    # GEMM) a first differing decision that the worker has proven to sit inside one fp16 ulp (assert_replay_complete)
    assert (diverged == -1 and matched == n_steps) or (world > 2 and diverged >= 0), res


def _tp_pipe_worker(rank, world, port, name, out_dir, tp_draft=False):
    """Two tensor-parallel ranks on cuda:0 (gloo for the setup only): the whole speculation step -- replicated draft,
    sharded target with BOTH collectives on the xGMI kernels, verifier, compactions -- captured as one hipGraph per rank
    and driven by the device; must commit the synchronous run's tokens on both ranks."""
    sys.path.insert(0, REPO); sys.path.insert(0, HERE)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SEQUOIA_TP_REQUIRE_XGMI"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = "cuda:0"
    from conftest import load_trace
    from helpers import dims_dict, make_tree, pipelined_run, sync_run, trace_state_dicts
    from sequoia_amd.Engine import ts_linear
    from sequoia_amd.Engine.Engine import GraphInferenceEngine
    from sequoia_amd.Engine.offload_engine import OffloadEngine
    from sequoia_amd.harness import tp_capturable
    ts_linear.DETERMINISTIC_PLANS = True
    z, meta = load_trace(name)
    M = meta["M"]
    sd_d, sd_t = trace_state_dicts(z, meta)
    dspec = dict(state_dict=sd_d, config=dims_dict(meta["draft_dims"], meta["vocab"]))
    tspec = dict(state_dict=sd_t, config=dims_dict(meta["target_dims"], meta["vocab"]))
    if tp_draft:          # the draft sharded like the target (harness.build does this for configuration E)
        from sequoia_amd.Engine.tp_engine import TPEngine
        draft = TPEngine(max_length=M, model_name_or_path=dspec, dtype=torch.float16, device=dev)
        assert draft.engine.xgmi is not None and draft.engine.kv_cache.k_cache.shape[2] == max(1, meta["draft_dims"][4] // world)
    else:
        draft = GraphInferenceEngine(max_length=M, model_name_or_path=dspec, dtype=torch.float16, device=dev)
    target = OffloadEngine(max_length=M, model_name_or_path=tspec, dtype=torch.float16, device=dev)
    assert target.engine.xgmi is not None and tp_capturable(target, draft)
    n_steps = int(z["n_steps"]) + 2
    tree = make_tree(z, meta, draft, target, dev)
    want = sync_run(tree, n_steps)
    draft.clear_kv(); target.clear_kv()
    tree = make_tree(z, meta, draft, target, dev, step_graph=True)
    assert tree.state is not None and tree.state.graph is not None
    got, _ = pipelined_run(tree, max_steps=len(want))
    torch.cuda.synchronize()
    assert target.engine.xgmi.status() == 0
    assert [g[0] for g in got] == [w[0] for w in want], (got, [w[0] for w in want])
    a_end = want[-1][0]
    assert np.array_equal(tree.tokens[:a_end].cpu().numpy(), want[-1][1][:a_end])
    mine = torch.tensor([g[0] for g in got])
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert all(torch.equal(b, mine) for b in both)
    ts_linear.assert_same_plans_across_ranks(draft.engine.model, target.engine.model)
    np.save(os.path.join(out_dir, f"tp{rank}.npy"), np.array([len(got), a_end]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,tp_draft", [("E_64x2", False), ("A_2chain", False), ("E_64x2", True), ("B_seq128", True),
                                           ("E_70b_w2", False), ("E_70b_w2", True)])
def test_tp2_whole_step_graph_device_driven_on_xgmi_collectives(name, tp_draft, tmp_path):
    port = 37900 + (os.getpid() % 1500)
    mp.spawn(_tp_pipe_worker, args=(2, port, name, str(tmp_path), tp_draft), nprocs=2, join=True)
    a = [np.load(tmp_path / f"tp{r}.npy") for r in range(2)]
    assert np.array_equal(a[0], a[1]) and a[0][0] >= 3
