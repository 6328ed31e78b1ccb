"""All-reduce of the tensor-parallel target's row-parallel projections over peer-mapped buffers (csrc/allreduce.hip).

One process per GPU.  Every rank allocates one uncached workspace through the C ABI, exports it as a hipIpc handle,
the 64-byte handles are exchanged once over torch.distributed (any backend: the transport of the SETUP, not of the data),
and every rank maps its peers' workspaces.  After that an all-reduce is one kernel launch per rank -- no RCCL call, no
host synchronisation, legal inside hipGraph capture -- in which each rank stores its 1/W slices straight into its
peers' memory over its own xGMI links (include/sequoia_hip.h, section e).

`XgmiAllReduce.create()` returns None (and says why) when the buffers cannot be set up or the self-check against
`dist.all_reduce` fails; the caller then stays on RCCL.  Slot in the reference: Engine/offload_engine.py:388-451 (the
host-offload engine this tensor-parallel engine replaces).
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from .. import native


# Where the workspaces live: "device" (default) = uncached device memory exported through hipIpc -- across GPUs the peers'
# stores travel over xGMI; "host" = a test rig: POSIX shared memory registered with the HIP runtime, so that on a ONE-GPU box
# every store, flag and poll of the protocol really leaves the device (PCIe) instead of meeting in the same HBM / L2 fabric
# (VERDICT r03 #4a: a missing system-scope fence or a cached flag read cannot hide there).
WS_MODE = os.environ.get("SEQUOIA_AR_WS", "device")
LAST_REFUSAL = None          # why the last create() handed back None (bench.py prints it)
_LIVE = []                   # weak references to every live instance (raise_on_fault)

FAULT_BITS = {1: "all-reduce phase 1 (a peer's partial rows never arrived)", 2: "all-reduce phase 2 (a peer's reduced chunk never arrived)",
              4: "all-gather 'written' flag", 8: "all-gather 'read' flag of the previous gather"}


class XgmiCollectiveTimeout(RuntimeError):
    pass


def raise_on_fault():
    """Called by the speculation loop once per step (Tree/_native_tree.py::verify / collect_step): a collective kernel whose
    bounded spin ran out has continued on stale data -- every token decided after that is wrong, so the job stops here.
    Costs a read of one word of pinned host memory per live instance; nothing when no instance exists."""
    dead = False
    for ref in _LIVE:
        ar = ref()
        if ar is None:
            dead = True
        else:
            ar.check_fault()
    if dead:
        _LIVE[:] = [r for r in _LIVE if r() is not None]


class XgmiAllReduce:
    def __init__(self, lib, rank, world, ws_ptrs, own, opened, max_elems, device, group, max_gather_elems=0, shared=None):
        self.lib, self.rank, self.world = lib, rank, world
        self._own, self._opened = own, opened
        self._shared = shared          # host mode: {"name", "host", "bytes", "peers": [(host_ptr, bytes)]}
        self.max_elems, self.max_gather_elems = max_elems, max_gather_elems
        self.device, self.group = device, group
        self._table = (C.c_void_p * world)(*ws_ptrs)
        self.calls = 0
        # the fault word: pinned host memory the kernels OR their timeout bits into (sq_ar_set_fault_word)
        self.fault = torch.zeros(16, dtype=torch.int32).pin_memory()
        own_ptr = own if isinstance(own, C.c_void_p) else C.c_void_p(own)
        if lib.sq_ar_set_fault_word(own_ptr, C.c_void_p(self.fault.data_ptr())) != native.SQ_OK:
            self.fault = None
        import weakref
        _LIVE.append(weakref.ref(self))

    def check_fault(self):
        if self.fault is None:
            return
        bits = int(self.fault[0])
        if bits:
            what = "; ".join(v for k, v in FAULT_BITS.items() if bits & k)
            raise XgmiCollectiveTimeout(f"rank {self.rank}: an xGMI collective gave up waiting for a peer (status bits {bits}: {what}); "
                                        "its result is invalid -- everything decoded after it would be too")

    def _pre(self):
        """Every collective looks at the pinned fault word before it launches (a host read, no device access, legal inside a
        capture): once a wait of this workspace has given up, every later wait polls ONCE (csrc/allreduce.hip: the sticky
        status word) and the kernels continue on whatever their areas hold -- a faulted workspace is dead, and a caller that
        never polls raise_on_fault() (a direct TPEngine / XgmiAllReduce user) must not get partial sums at full speed
        (ADVICE r05).  The raise comes one call late at the earliest: the word is written by the kernel that timed out."""
        self.check_fault()

    def clear_fault(self):
        """Tests that provoke a timeout on purpose."""
        if self.fault is not None:
            self.fault.zero_()

    # ---- setup ------------------------------------------------------------------------------------------------
    @staticmethod
    def create(group=None, device="cuda:0", max_elems=144 * 8192, self_check=True, max_gather_elems=0):
        """max_elems: the largest tensor (fp16 elements) the job will reduce -- [144 rows, hidden]; max_gather_elems: the
        largest gathered tensor -- [144 rows, vocabulary]."""
        if not (dist.is_available() and dist.is_initialized()) or not str(device).startswith("cuda"):
            return None
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if world == 1 or world > 8:
            return None
        lib = native.load()
        torch.cuda.set_device(device)
        nbytes = int(lib.sq_ar_workspace_bytes(world, max_elems, max_gather_elems))
        if WS_MODE == "host":
            return XgmiAllReduce._create_host(lib, group, device, world, rank, nbytes, max_elems, max_gather_elems, self_check)
        own = C.c_void_p()
        ok = lib.sq_ar_alloc(C.byref(own), nbytes) == native.SQ_OK
        handle = (C.c_ubyte * 64)()
        if ok:
            ok = lib.sq_ar_ipc_export(own, handle) == native.SQ_OK
        # every rank learns whether every rank got its buffer: all or nothing
        infos = [None] * world
        dist.all_gather_object(infos, (bool(ok), bytes(handle), os.getpid()), group=group)
        if not all(i[0] for i in infos):
            if own.value:
                lib.sq_ar_free(own)
            return _refuse(rank, "workspace allocation / hipIpc export failed on rank(s) "
                           + str([r for r, i in enumerate(infos) if not i[0]]))
        ptrs, opened, failed = [], [], False
        for r, (_, h, pid) in enumerate(infos):
            if r == rank:
                ptrs.append(own.value)
                continue
            p = C.c_void_p()
            buf = (C.c_ubyte * 64).from_buffer_copy(h)
            if lib.sq_ar_ipc_open(buf, C.byref(p)) != native.SQ_OK:
                failed = True
                ptrs.append(None)
            else:
                ptrs.append(p.value)
                opened.append(p.value)
        flags = [None] * world
        dist.all_gather_object(flags, failed, group=group)
        ar = XgmiAllReduce(lib, rank, world, ptrs if not failed else [own.value] * world, own, opened, max_elems, device, group,
                           max_gather_elems)
        if any(flags):
            ar.close()
            return _refuse(rank, "hipIpcOpenMemHandle failed on rank(s) " + str([r for r, f in enumerate(flags) if f]))
        if self_check and not ar._self_check():
            ar.close()
            return None
        return ar

    @staticmethod
    def _create_host(lib, group, device, world, rank, nbytes, max_elems, max_gather_elems, self_check):
        """SEQUOIA_AR_WS=host: every rank creates one shared-memory object, the NAMES are exchanged, every rank maps and
        registers every peer's object (sq_ar_shared_host_open)."""
        global _SHM_SEQ
        _SHM_SEQ += 1
        name = f"/sequoia_ar_{os.getpid()}_{rank}_{_SHM_SEQ}".encode()
        host, devp = C.c_void_p(), C.c_void_p()
        ok = lib.sq_ar_shared_host_open(name, nbytes, 1, C.byref(host), C.byref(devp)) == native.SQ_OK
        infos = [None] * world
        dist.all_gather_object(infos, (bool(ok), name), group=group)
        if not all(i[0] for i in infos):
            if ok:
                lib.sq_ar_shared_host_close(name, host, nbytes)
            return _refuse(rank, "host-memory workspace (shm_open / hipHostRegister) failed on rank(s) "
                           + str([r for r, i in enumerate(infos) if not i[0]]))
        ptrs, peers, failed = [], [], False
        for r, (_, pname) in enumerate(infos):
            if r == rank:
                ptrs.append(devp.value)
                continue
            ph, pd = C.c_void_p(), C.c_void_p()
            if lib.sq_ar_shared_host_open(pname, nbytes, 0, C.byref(ph), C.byref(pd)) != native.SQ_OK:
                failed = True
                ptrs.append(devp.value)
            else:
                ptrs.append(pd.value)
                peers.append((ph.value, nbytes))
        flags = [None] * world
        dist.all_gather_object(flags, failed, group=group)      # (also: every rank has mapped every object -> names may go)
        # Every peer holds its mapping now: the NAME can go at once (the mappings stay valid after shm_unlink).  A worker that
        # crashes or is killed later leaves nothing under /dev/shm, and a recycled pid can never collide with a leaked name
        # (the objects used to be unlinked in close() only).
        try:
            import _posixshmem
            _posixshmem.shm_unlink(name.decode())
            name = None
        except (ImportError, OSError):
            pass                                                # close() unlinks it then
        shared = dict(name=name, host=host.value, bytes=nbytes, peers=peers)
        ar = XgmiAllReduce(lib, rank, world, ptrs, devp, [], max_elems, device, group, max_gather_elems, shared=shared)
        if any(flags):
            ar.close()
            return _refuse(rank, "mapping a peer's host-memory workspace failed on rank(s) " + str([r for r, f in enumerate(flags) if f]))
        if self_check and not ar._self_check():
            ar.close()
            return None
        return ar

    def _self_check(self) -> bool:
        """A few reductions of rank-dependent data against dist.all_reduce (which also keeps the ranks in step), the
        all-gather against dist.all_gather, the all-reduce + skip + RMSNorm kernel against the all-reduce followed by
        sq_add_rmsnorm_f16; all ranks must agree that all ranks passed.  Every rank runs every check to the end whatever it
        finds (the checks are collectives: a rank that left early would strand the others), the kernels' spins are
        bounded."""
        ok = True
        gen = torch.Generator(device="cpu")
        for i, n in enumerate((8, 4096, 129 * 1024, min(self.max_elems, 129 * 8192))):
            n = (min(n, self.max_elems) // 8) * 8
            gen.manual_seed(1000 * i + self.rank)
            x = (torch.randn(n, generator=gen) * 2).to(torch.float16).to(self.device)
            want = x.float()
            dist.all_reduce(want, group=self.group)
            got = self(x.clone())
            torch.cuda.synchronize(self.device)
            if self.status() != 0 or not torch.allclose(got.float(), want, rtol=0, atol=2e-2 * self.world):
                ok = False
            # bit-identical on every rank (each element is reduced by exactly one rank)
            same = got.clone().view(torch.int16).to(torch.int32)
            lo, hi = same.clone(), same.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo, hi):
                ok = False
        if self.max_gather_elems >= 8 * self.world:
            v = max(8, (min(4000, self.max_gather_elems // (self.world * 5)) // 8) * 8)
            for rows in (1, 5):
                gen.manual_seed(777 + rows + 10 * self.rank)
                sl = torch.randn(rows, v, generator=gen).to(torch.float16).to(self.device)
                parts = [torch.empty_like(sl) for _ in range(self.world)]
                dist.all_gather(parts, sl, group=self.group)
                got = self.gather_cols(sl)
                torch.cuda.synchronize(self.device)
                if self.status() != 0 or not torch.equal(got, torch.cat(parts, dim=1)):
                    ok = False
        from ..ops import get_ops
        ops = get_ops()
        for rows, hidden in ((1, 256), (5, 1024), (129, 1024)):
            if not self.fits_rows(rows, hidden) or rows * hidden > self.max_elems:
                continue                                          # (same decision on every rank: sizes only)
            gen.manual_seed(4242 + rows + self.rank)
            part = torch.randn(rows, hidden, generator=gen).to(torch.float16).to(self.device)
            gen.manual_seed(4242 + rows)                          # replicated residual stream and weight
            x0 = torch.randn(rows, hidden, generator=gen).to(torch.float16).to(self.device)
            g = (1.0 + 0.1 * torch.randn(hidden, generator=gen)).to(torch.float16).to(self.device)
            red = self(part.clone().reshape(-1)).reshape(rows, hidden)
            x_ref, o_ref = x0.clone(), torch.empty_like(x0)
            ops.add_rmsnorm(red, x_ref, x_ref, g, o_ref, 1e-5)
            x, out = x0.clone(), torch.empty_like(x0)
            self.reduce_add_rmsnorm(part, x, g, out, 1e-5, False)
            torch.cuda.synchronize(self.device)
            if self.status() != 0 or not torch.equal(x, x_ref) or not torch.equal(out, o_ref):
                ok = False
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, ok, group=self.group)
        if not all(verdicts):
            _refuse(self.rank, f"self-check against dist.all_reduce failed on rank(s) {[r for r, v in enumerate(verdicts) if not v]}"
                               f" (status {self.status()})")
        self.clear_fault()       # (a failed self-check may have raised bits: the instance is closed by the caller anyway)
        return all(verdicts)

    # ---- the call ---------------------------------------------------------------------------------------------
    def __call__(self, x: torch.Tensor, blocks: int = 0) -> torch.Tensor:
        """In-place sum over the ranks of a contiguous fp16 tensor (numel % 8 == 0, numel <= max_elems)."""
        self._pre()
        if x.dtype != torch.float16 or not x.is_contiguous() or x.device.type != "cuda":
            raise TypeError("xgmi all-reduce: contiguous fp16 tensor on the device")
        native.check(self.lib.sq_allreduce_sum_f16(x.data_ptr(), x.numel(), self.rank, self.world, self._table, self.max_elems,
                                                   int(blocks), torch.cuda.current_stream().cuda_stream), "sq_allreduce_sum_f16")
        self.calls += 1
        return x

    def reduce_slabs(self, slab: torch.Tensor, splits: int, out: torch.Tensor, blocks: int = 0) -> torch.Tensor:
        """out[n] (fp16) = sum over the ranks of h(sum_s slab[s][n]): the all-reduce of a split-K row-parallel projection
        straight from its fp32 partials (slab: [>= splits * n] fp32, n = out.numel())."""
        self._pre()
        n = out.numel()
        assert slab.dtype == torch.float32 and slab.numel() >= splits * n and out.dtype == torch.float16 and out.is_contiguous()
        native.check(self.lib.sq_allreduce_sum_slabs_f16(slab.data_ptr(), int(splits), out.data_ptr(), n, self.rank, self.world,
                                                         self._table, self.max_elems, int(blocks),
                                                         torch.cuda.current_stream().cuda_stream), "sq_allreduce_sum_slabs_f16")
        self.calls += 1
        return out

    def fits_rows(self, rows: int, hidden: int) -> bool:
        """Row-aligned form: a rank's share of the rows must fit its chunk of the staging areas."""
        chunk_cap = (-(-self.max_elems // self.world) + 7) // 8 * 8
        return rows > 0 and hidden % 32 == 0 and hidden <= 16384 and -(-rows // self.world) * hidden <= chunk_cap

    def reduce_add_rmsnorm(self, partial, x: torch.Tensor, weight: torch.Tensor, out: torch.Tensor, eps: float,
                           out_frag: bool, splits: int = 0) -> torch.Tensor:
        """x <- x + all-reduce(partial); out = RMSNorm(x) * weight (row-major or fragment-major), one launch.
        partial: fp16 rows [rows, hidden] (splits == 0) or the fp32 split-K slab [splits][rows][hidden]."""
        self._pre()
        rows, hidden = x.shape
        assert x.dtype == torch.float16 and x.is_contiguous() and weight.dtype == torch.float16 and out.dtype == torch.float16
        if splits:
            assert partial.dtype == torch.float32 and partial.numel() >= splits * rows * hidden
            slab_p, rows_p = partial.data_ptr(), None
        else:
            assert partial.dtype == torch.float16 and partial.is_contiguous() and partial.numel() == rows * hidden
            slab_p, rows_p = None, partial.data_ptr()
        native.check(self.lib.sq_allreduce_add_rmsnorm_f16(slab_p, int(splits), rows_p, x.data_ptr(), weight.data_ptr(),
                                                           out.data_ptr(), int(bool(out_frag)), rows, hidden, float(eps),
                                                           self.rank, self.world, self._table, self.max_elems,
                                                           torch.cuda.current_stream().cuda_stream),
                     "sq_allreduce_add_rmsnorm_f16")
        self.calls += 1
        return out

    def gather_cols(self, slice_: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """[rows, v] per rank -> [rows, world v] on every rank (rank r's columns at [r v, (r + 1) v))."""
        self._pre()
        rows, v = slice_.shape
        # The per-(peer, block) "read" handshake protects the bytes a block overwrites in the peer's image only while the slice
        # WIDTH is the one of the previous gather (offsets in the image are i W v + R v: another v moves every row).  The product
        # path gathers one width per workspace (the vocabulary shard); a caller that changes it gets a full stop in between
        # instead of a race, and inside a graph capture -- where no stop is possible -- a loud error.
        last = getattr(self, "_gather_v", None)
        if last is not None and last != v:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"all-gather width changed from {last} to {v} inside a graph capture: one width per workspace "
                                   "between synchronisations (include/sequoia_hip.h, sq_allgather_cols_f16)")
            torch.cuda.synchronize(self.device)
            if dist.is_available() and dist.is_initialized():
                dist.barrier(group=self.group)
        self._gather_v = v
        if out is None:
            out = torch.empty((rows, self.world * v), dtype=slice_.dtype, device=slice_.device)
        native.check(self.lib.sq_allgather_cols_f16(slice_.data_ptr(), out.data_ptr(), rows, v, self.rank, self.world, self._table,
                                                    self.max_elems, self.max_gather_elems,
                                                    torch.cuda.current_stream().cuda_stream), "sq_allgather_cols_f16")
        self.calls += 1
        return out

    def fits_gather(self, slice_: torch.Tensor) -> bool:
        return (slice_.dtype == torch.float16 and slice_.is_contiguous() and slice_.dim() == 2 and slice_.shape[1] % 8 == 0
                and 0 < slice_.numel() * self.world <= self.max_gather_elems)

    def fits(self, x: torch.Tensor) -> bool:
        return x.dtype == torch.float16 and x.is_contiguous() and x.numel() % 8 == 0 and 0 < x.numel() <= self.max_elems

    def status(self) -> int:
        st = C.c_int(0)
        self.lib.sq_ar_status(self._own, C.byref(st))
        return int(st.value)

    def close(self):
        _LIVE[:] = [r for r in _LIVE if r() is not None and r() is not self]
        if self._shared is not None:
            sh, self._shared = self._shared, None
            torch.cuda.synchronize(self.device)
            for ph, nb in sh["peers"]:
                self.lib.sq_ar_shared_host_close(None, C.c_void_p(ph), nb)
            self.lib.sq_ar_shared_host_close(sh["name"], C.c_void_p(sh["host"]), sh["bytes"])
            self._own = None
            return
        for p in self._opened:
            self.lib.sq_ar_ipc_close(C.c_void_p(p))
        self._opened = []
        if self._own is not None and self._own.value:
            self.lib.sq_ar_free(self._own)
            self._own = None


_SHM_SEQ = 0


def _refuse(rank, why):
    global LAST_REFUSAL
    LAST_REFUSAL = why
    if rank == 0:
        import sys
        print(f"sequoia_amd: xGMI all-reduce disabled ({why}); staying on RCCL", file=sys.stderr)
    return None
