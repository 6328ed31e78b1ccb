"""CPU checks of the C-ABI boundary and the drop-in aliases (no GPU, no compute launches)."""
import os
import re
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from sequoia_amd import native
    lib = native.load()
    header = open(os.path.join(REPO, "include", "sequoia_hip.h")).read()
    # measurement aids live behind SEQUOIA_BUILD_PROBES and are not part of the default surface
    probes = re.findall(r"#ifdef SEQUOIA_BUILD_PROBES.*?#endif", header, flags=re.S)
    probe_names = set(re.findall(r"\b(sq_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", " ", " ".join(probes), flags=re.S)))
    assert probe_names == set(native.PROBE_PROTOTYPES)
    from sequoia_amd.build import PROBES
    if not PROBES:
        for name in probe_names:
            assert not hasattr(lib, name), f"{name} is a measurement aid: it must not be exported by a default build"
    header = re.sub(r"#ifdef SEQUOIA_BUILD_PROBES.*?#endif", " ", header, flags=re.S)
    declared = set(re.findall(r"\b(sq_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/sequoia_hip.h but not exported"
    assert declared == set(native.PROTOTYPES), declared ^ set(native.PROTOTYPES)
    assert lib.sq_version() >= 100
    # the ctypes prototypes carry one argtype per declared parameter (header <-> binding drift)
    text = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    for name, params in re.findall(r"\b(sq_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = params.strip()
        n_params = 0 if params in ("", "void") else params.count(",") + 1
        assert n_params == len(native.PROTOTYPES[name][1]), f"{name}: header declares {n_params} parameters, binding " \
                                                           f"{len(native.PROTOTYPES[name][1])}"


def test_host_helper_bitmask_matches_growmap_masks():
    """sq_tree_bitmask_from_successors (host code of the library) on every bundled growmap ==
    the mask the reference stores in its .pt files (sha recorded by oracle/export_fixtures.py)."""
    import hashlib
    import json
    from sequoia_amd import native
    from sequoia_amd.growmap import BUILTIN_DIR, GrowMap
    lib = native.load()
    for fn in sorted(os.listdir(BUILTIN_DIR)):
        if not fn.endswith(".json") or "prompts" in fn:
            continue
        rec = json.load(open(os.path.join(BUILTIN_DIR, fn)))
        g = GrowMap.from_successors(rec["Successors"])
        out = np.zeros_like(g.bitmask)
        rc = lib.sq_tree_bitmask_from_successors(g.child_off.ctypes.data,
                                                 g.child_ids.ctypes.data if g.size > 1 else None, g.size,
                                                 out.ctypes.data, out.shape[1])
        assert rc == 0 and np.array_equal(out, g.bitmask), fn
        chk = rec.get("check")
        if chk is None:          # growmaps searched here (sequoia_amd.tree_search) carry no reference record
            continue
        assert g.roots == chk["roots"] and g.branches == chk["branches"] and g.depth.tolist() == chk["depth"], fn
        assert hashlib.sha256(g.dense_mask().tobytes()).hexdigest() == chk["mask_sha"], fn
        ref = g.to_reference_dict()
        assert ref["size"] == rec["size"] and ref["Successors"] == rec["Successors"]


def test_argument_validation_returns_error_codes():
    from sequoia_amd import native
    lib = native.load()
    assert lib.sq_tree_bitmask_from_successors(None, None, 4, None, 1) == native.SQ_EINVAL
    assert lib.sq_kv_compact_f16(None, None, 1, 1, 8, 64, None, None, 0, 0, 0, None, None) == native.SQ_EINVAL
    assert lib.sq_sample_wor_f16(None, 0, None, 0, None, 1, 32000, 4, 0.6, None, None, None, None, None, None,
                                 None) == native.SQ_EINVAL
    assert lib.sq_sample_workspace_bytes(34, 32000, 19) >= 34 * 8 * (8 + 19 * 4) and lib.sq_sample_workspace_bytes(0, 32000, 4) == 0
    assert lib.sq_verify_stochastic_f16(None, None, None, 0, None, None, None, 4, 32000, 5, 0.6, 1, None, None, None, None, 0,
                                        None, None) == native.SQ_EINVAL
    assert lib.sq_verify_workspace_bytes(128) > 0 and lib.sq_verify_workspace_bytes(0) == 0
    with pytest.raises(native.SequoiaNativeError):
        native.check(native.SQ_EUNSUPPORTED, "x")


def test_dropin_aliases_resolve_to_this_package():
    import sequoia_amd.dropin as dropin
    for m in list(sys.modules):
        if m in ("utils", "Tree", "Engine") or m.startswith(("Tree.", "Engine.")):
            del sys.modules[m]
    dropin.install()
    try:
        from Engine.Engine import GraphInferenceEngine, GraphInferenceEngineTG  # noqa: F401
        from Engine.offload_engine import OffloadEngine  # noqa: F401
        from Tree.GreedyTree import GreedyTree  # noqa: F401
        from Tree.SpecTree import SpecTree
        from utils import (_make_causal_mask, cuda_graph_for_residual, cuda_graph_for_sampling_argmax,  # noqa: F401
                           cuda_graph_for_sampling_without_replacement, get_sampling_logits)
        assert SpecTree.__module__ == "sequoia_amd.Tree.SpecTree"
        import inspect
        params = list(inspect.signature(SpecTree.__init__).parameters)
        # the reference's constructor keywords, in order (Tree/SpecTree.py:8-28)
        assert params[1:21] == ["draft_model_engine", "target_model_engine", "prefix", "temperature", "top_p",
                                "draft_kv_len", "target_kv_len", "max_length", "device", "max_target_seq", "vocab_size",
                                "grow_map", "attn_mask", "sequence", "new_tokens_buffer", "parents_buffer",
                                "position_ids", "residual_graph", "sampling_callables", "sample_gather_indices"]
    finally:
        dropin.uninstall()


def test_model_spec_errors_are_loud():
    from sequoia_amd.Engine.Llama_model import parse_model_spec
    with pytest.raises(FileNotFoundError):
        parse_model_spec("meta-llama/Llama-2-7b-hf")
    kind, (arch, opts) = parse_model_spec("random:JackFram/llama-68m:seed=3")
    assert kind == "random" and arch == "JackFram/llama-68m" and opts["seed"] == "3"


def test_attention_block_map_covers_every_tile_once_and_keeps_kv_heads_on_few_xcds():
    """sq_tree_attention_block_decode = the kernel's own blockIdx -> (head, query tile) function (att_decode_block, shared by
    host and device code): for every head configuration of the bundled models, tensor-parallel shards (fewer than 8 KV
    heads, down to ONE for the 70B target at TP = 8) and odd test shapes, every (head, tile) is computed by exactly one
    block, and the blocks of one KV head sit on the XCDs (block % 8) the design says: one XCD for >= 8 KV heads, at most
    ceil(items / 32) -- never more than the head's 8 / h_kv share -- below that."""
    import ctypes as C
    from sequoia_amd import native
    lib = native.load()
    cases = [(32, 32), (40, 40), (12, 12), (16, 16), (64, 8), (8, 1), (16, 2), (32, 4), (8, 8), (4, 4), (2, 2), (5, 5), (4, 1), (6, 3), (7, 7),
             (1, 1), (24, 8), (9, 3)]
    for n_heads, h_kv in cases:
        for q_len in (1, 16, 17, 34, 64, 128, 129, 144, 255):
            n_tiles = (q_len + 15) // 16
            nb = C.c_int(0)
            assert lib.sq_tree_attention_block_decode(-1, q_len, n_heads, h_kv, None, None, C.byref(nb)) == 0
            seen, xcds = {}, {}
            for b in range(nb.value):
                h, t = C.c_int(0), C.c_int(0)
                assert lib.sq_tree_attention_block_decode(b, q_len, n_heads, h_kv, C.byref(h), C.byref(t), C.byref(nb)) == 0
                if h.value < 0:
                    continue
                assert 0 <= h.value < n_heads and 0 <= t.value < n_tiles
                assert (h.value, t.value) not in seen, (n_heads, h_kv, q_len, b)
                seen[(h.value, t.value)] = b
                xcds.setdefault(h.value // (n_heads // h_kv), set()).add(b % 8)
            assert len(seen) == n_heads * n_tiles, (n_heads, h_kv, q_len, len(seen))
            items = (n_heads // h_kv) * n_tiles
            for kvh, xs in xcds.items():
                want = 1 if h_kv >= 8 else max(1, min(8 // h_kv, (items + 31) // 32))
                assert len(xs) <= want, (n_heads, h_kv, q_len, kvh, xs)
            assert nb.value <= 8 * ((n_heads * n_tiles + 7) // 8 + n_tiles * max(1, n_heads // h_kv) * 8), "grid far larger than the work"
