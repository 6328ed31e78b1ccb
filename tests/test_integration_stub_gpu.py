"""The ctypes stub INTEGRATION.md shows to a maintainer of the reference (section B), executed as written: raw ctypes
argtypes, torch tensors, no sequoia_amd Python in between -- checked against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ctypes_stub_of_integration_md():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import ops_np as O
    _lib = C.CDLL(os.path.join(REPO, "sequoia_amd", "lib", "libsequoia_hip.so"))
    _vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
    _lib.sq_sample_workspace_bytes.restype = C.c_size_t
    _lib.sq_sample_workspace_bytes.argtypes = [_i, _i, _i]
    _lib.sq_sample_wor_f16.restype = _i
    _lib.sq_sample_wor_f16.argtypes = [_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    _lib.sq_kv_compact_f16.restype = _i
    _lib.sq_kv_compact_f16.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp]

    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def sampling_without_replacement(sampling_logits, rand, num_samples, temperature):
        n, V = sampling_logits.shape
        out = torch.empty(n * num_samples, dtype=torch.long, device=sampling_logits.device)
        ws = torch.empty(_lib.sq_sample_workspace_bytes(n, V, num_samples), dtype=torch.uint8, device=out.device)
        rc = _lib.sq_sample_wor_f16(sampling_logits.data_ptr(), sampling_logits.stride(0), rand.data_ptr(),
                                    rand.stride(0), None, n, V, num_samples, temperature, out.data_ptr(),
                                    None, None, None, None, ws.data_ptr(), _stream())
        assert rc == 0
        return out

    rng = np.random.RandomState(5)
    n, V, k, T = 5, 32000, 8, 0.6
    logits = (rng.randn(n, V) * 2.5).astype(np.float16)
    rand = (rng.randint(0, 2048, size=(n, V)) / 2048.0).astype(np.float16)       # torch's fp16 uniform_ grid
    got = sampling_without_replacement(torch.from_numpy(logits).cuda(), torch.from_numpy(rand).cuda(), k, T)
    want = O.sample_wor(logits, rand, k, T)
    keys = O.sample_keys(logits, rand, T)
    got = got.cpu().numpy().reshape(n, k)
    for r, c in np.argwhere(got != want):       # a differing pick must be an exact-or-adjacent fp16 key (exp / log last ulp)
        a = int(keys[r, got[r, c]].view(np.int16)); b = int(keys[r, want[r, c]].view(np.int16))
        assert abs(a - b) <= 1, (r, c)
    assert (got != want).sum() <= 1

    # Engine/Llama_KV.py:60-68 gather_kv_incremental through sq_kv_compact_f16
    L, H, M, D = 2, 3, 64, 64
    kc = torch.from_numpy(rng.randn(L, 1, H, M, D).astype(np.float16)).cuda()
    vc = torch.from_numpy(rng.randn(L, 1, H, M, D).astype(np.float16)).cuda()
    k0, v0 = kc.clone(), vc.clone()
    indices, offset = [12, 15, 19], 10
    slots = torch.tensor(indices, dtype=torch.int32, device="cuda")
    rc = _lib.sq_kv_compact_f16(kc.data_ptr(), vc.data_ptr(), L, H, M, D, slots.data_ptr(), None, len(indices), offset, M,
                                None, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(kc[..., offset:offset + 3, :], k0[..., indices, :]) and torch.equal(vc[..., offset:offset + 3, :], v0[..., indices, :])
    assert torch.equal(kc[..., :offset, :], k0[..., :offset, :]) and float(kc[..., offset + 3:, :].abs().max()) == 0.0
