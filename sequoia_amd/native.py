"""ctypes binding of libsequoia_hip.so (the C ABI declared in include/sequoia_hip.h).

No torch types cross this boundary: callers pass raw device pointers (tensor.data_ptr()) and
sizes.  The library must exist; there is no fallback.  `load()` raises if it is missing.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# (SEQUOIA_LIB: an experiment build of the same ABI, e.g. tools/block_dbg_build.sh -- never a different implementation)
LIB_PATH = os.environ.get("SEQUOIA_LIB") or os.path.join(_PKG, "lib", "libsequoia_hip.so")

SQ_OK, SQ_EINVAL, SQ_EUNSUPPORTED, SQ_ELAUNCH = 0, -1, -2, -3
SQ_MAX_TREE = 512
SQ_MAX_TOPK = 128
SQ_RESULT_INTS = 64
SQ_RES_ACCEPT_LEN, SQ_RES_N_TREE, SQ_RES_BONUS, SQ_RES_TERMINAL = 0, 1, 2, 3
SQ_RES_REASON, SQ_RES_GT, SQ_RES_LAST_NODE, SQ_RES_SLOTS = 4, 5, 6, 8
SQ_STEP_GT, SQ_STEP_NEXT_GT, SQ_STEP_INDEX, SQ_STEP_ACTIVE, SQ_STEP_INTS = 0, 1, 2, 3, 8
SQ_RESULT_RING = 4
SQ_REASON_SKIPPED = 4
SQ_ATT_OUT_FRAG = 0x100
SQ_VERIFY_GATHER_FIRST = 0x80000000

_vp, _i, _i64, _f, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32

# name -> (restype, argtypes); mirrors include/sequoia_hip.h one to one
PROTOTYPES = {
    "sq_version": (_i, []),
    "sq_last_error": (C.c_char_p, []),
    "sq_device_ready": (_i, []),
    "sq_tree_bitmask_from_successors": (_i, [_vp, _vp, _i, _vp, _i]),
    "sq_tree_mask_dense_f16": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "sq_kv_scatter_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sq_kv_compact_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sq_kv_compact2_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "sq_kv_clear_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sq_rope_kv_write_f16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sq_rope_kv_write_slabs_f16": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sq_tree_attention_block_decode": (_i, [_i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "sq_tree_attention_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _i, _i, _i,
                                   _vp, _i, _vp, _vp]),
    "sq_store_i32": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "sq_sample_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "sq_logits_stats_f16": (_i, [_vp, _i64, _vp, _i, _i, _f, _vp, _i, _vp, _i64, _vp]),
    "sq_sample_wor_f16": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sq_topk_f16": (_i, [_vp, _i64, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sq_stage_inputs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "sq_stage_tree_inputs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "sq_verify_workspace_bytes": (C.c_size_t, [_i]),
    "sq_sample_iid_f16": (_i, [_vp, _i64, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "sq_verify_specinfer_f16": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _u32, _vp, _vp, _vp]),
    "sq_sample_wor_f32noise_f16": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "sq_verify_probe_f16": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _u32, _vp, _vp, _vp]),
    "sq_verify_tokens_f16": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sq_verify_stochastic_f16": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _u32, _vp, _vp, _vp, _vp, _i, _vp,
                                      _vp]),
    "sq_top_p_filter_f16": (_i, [_vp, _i64, _i, _i, _f, _f, _vp]),
    "sq_verify_greedy_f16": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sq_rmsnorm_f16": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "sq_silu_mul_f16": (_i, [_vp, _vp, _i, _i, _vp]),
    "sq_add_rmsnorm_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "sq_linear_ts_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "sq_linear_ts_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_size_t, _vp]),
    "sq_repack_linear_weight_f16": (_i, [_vp, _vp, _i, _i, _vp]),
    "sq_repack_rows_frag_f16": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "sq_add_rmsnorm_slabs_f16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sq_embed_rmsnorm_f16": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sq_embed_stage_rmsnorm_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sq_rmsnorm_frag_f16": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "sq_add_rmsnorm_frag_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "sq_silu_mul_frag_f16": (_i, [_vp, _vp, _i, _i, _vp]),
    "sq_silu_mul_slabs_f16": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "sq_draft_attn_block_f16": (_i, [_vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i,
                                     _vp, _i, _vp, _i, _vp]),
    "sq_level_attention_f16": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _i,
                                    _vp, _vp]),
    "sq_ar_workspace_bytes": (C.c_size_t, [_i, C.c_size_t, C.c_size_t]),
    "sq_ar_alloc": (_i, [C.POINTER(_vp), C.c_size_t]),
    "sq_ar_free": (_i, [_vp]),
    "sq_ar_ipc_export": (_i, [_vp, _vp]),
    "sq_ar_ipc_open": (_i, [_vp, C.POINTER(_vp)]),
    "sq_ar_ipc_close": (_i, [_vp]),
    "sq_ar_status": (_i, [_vp, C.POINTER(_i)]),
    "sq_ar_set_fault_word": (_i, [_vp, _vp]),
    "sq_ar_shared_host_open": (_i, [C.c_char_p, C.c_size_t, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    "sq_ar_shared_host_close": (_i, [C.c_char_p, _vp, C.c_size_t]),
    "sq_allreduce_sum_f16": (_i, [_vp, C.c_size_t, _i, _i, C.POINTER(_vp), C.c_size_t, _i, _vp]),
    "sq_allreduce_sum_slabs_f16": (_i, [_vp, _i, _vp, C.c_size_t, _i, _i, C.POINTER(_vp), C.c_size_t, _i, _vp]),
    "sq_allgather_cols_f16": (_i, [_vp, _vp, _i, _i, _i, _i, C.POINTER(_vp), C.c_size_t, C.c_size_t, _vp]),
    "sq_allreduce_add_rmsnorm_f16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _i, C.POINTER(_vp), C.c_size_t, _vp]),
}

_lib = None


class SequoiaNativeError(RuntimeError):
    pass


# measurement aids outside the library's default surface (include/sequoia_hip.h: #ifdef SEQUOIA_BUILD_PROBES)
PROBE_PROTOTYPES = {
    "sq_norm_linear_f16": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sq_linear_ts_prefetch": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
}


def load() -> C.CDLL:
    """Load the shared library (once).  Raises SequoiaNativeError when it is absent: the
    product path never substitutes another implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SequoiaNativeError(
            f"{LIB_PATH} not found: build it with `python -m sequoia_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no non-HIP fallback")
    # The library and PyTorch must share ONE HIP runtime (streams and device pointers cross the boundary): torch first,
    # so that the library's libamdhip64 dependency resolves to the copy torch has already mapped.  Loaded the other way
    # round the process holds two runtimes and sq_device_ready() reports no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # header / library drift
            raise SequoiaNativeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in PROBE_PROTOTYPES.items():      # present only in a SEQUOIA_BUILD_PROBES=1 build
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if os.environ.get("SEQUOIA_TRACE_CALLS", "0") == "1":
        lib = _TracedLib(lib)
    _lib = lib
    return lib


class _TracedLib:
    """Debug aid (SEQUOIA_TRACE_CALLS=1): every C-ABI call is announced on stderr BEFORE it runs and followed by a device
    synchronisation, so that a GPU memory fault (which aborts the process asynchronously) names the entry point it happened
    in.  Orders of magnitude slower; never on by default."""

    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("sq_") or name in ("sq_last_error", "sq_version", "sq_device_ready"):
            return fn
        import sys

        def call(*a):
            print(f"sq-call {name}{tuple(x if isinstance(x, (int, float)) else type(x).__name__ for x in a)}", file=sys.stderr, flush=True)
            rc = fn(*a)
            import torch
            if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()
            return rc
        return call


def check(rc: int, what: str) -> None:
    if rc == SQ_OK:
        return
    msg = {SQ_EINVAL: "invalid argument", SQ_EUNSUPPORTED: "unsupported shape", SQ_ELAUNCH: "HIP launch error"}.get(
        rc, f"error {rc}")
    detail = ""
    if rc == SQ_ELAUNCH and _lib is not None:
        detail = ": " + (_lib.sq_last_error() or b"").decode()
    raise SequoiaNativeError(f"{what}: {msg}{detail}")
