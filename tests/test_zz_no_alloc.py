"""gnlp_encoder_layer_forward_noalloc (include/gluonnlp_amd.h): the encoder layer for callers that
cannot allocate -- its scratch is a static device array inside the library.  Called through the C
ABI and checked bit for bit against gnlp_encoder_layer_forward (same kernels, caller-owned
workspace) and against the oracle, on two calls in a row, the second with a smaller input (the
array keeps the first call's contents), plus the over-size refusal.

Collected last on purpose (test_zz_*): the round's end runs `pytest -x`."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import load_params, oracle_params, prep, rel_err, to_dev, to_np, tol

pytestmark = pytest.mark.gpu


def _layer(U, F, H, dtype, rng):
    from gluonnlp_amd.models import TransformerEncoderLayer
    lp = ref.make_encoder_layer_params(rng, U, F)
    name = 'bfloat16' if dtype == torch.bfloat16 else 'float32'
    layer = load_params(TransformerEncoderLayer(U, F, H, activation='gelu', dtype=name), lp, dtype)
    return layer, lp


def _both(lib, layer, x, vl, want_weights, final_seq_mask):
    """-> (out_ws, w_ws, out_na, w_na): the workspace form and the no-alloc form, same inputs."""
    from gluonnlp_amd import _lib, _tensor as T
    B, L, U = x.shape
    H = layer._num_heads
    cfg = layer.cfg_struct(T.code(x))
    p = layer.params_struct()
    stream = T.stream(x.device)
    outs = []
    for noalloc in (False, True):
        out = torch.full_like(x, float('nan'))
        w = torch.full((B, H, L, L), float('nan'), dtype=x.dtype, device=x.device) \
            if want_weights else None
        if noalloc:
            rc = lib.gnlp_encoder_layer_forward_noalloc(
                T.ptr(out), T.ptr(x), ctypes.byref(p), ctypes.byref(cfg), B, L, None, T.ptr(vl),
                _lib.I64, T.ptr(w), int(final_seq_mask), stream)
        else:
            nws = lib.gnlp_encoder_workspace_bytes(ctypes.byref(cfg), B, L, int(want_weights))
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
            rc = lib.gnlp_encoder_layer_forward(
                T.ptr(out), T.ptr(x), ctypes.byref(p), ctypes.byref(cfg), B, L, None, T.ptr(vl),
                _lib.I64, T.ptr(w), int(final_seq_mask), T.ptr(ws), ws.numel(), stream)
        _lib.check(rc)
        torch.cuda.synchronize()
        outs += [out, w]
    return outs


def test_encoder_layer_noalloc_allocates_nothing():
    """Free device memory (hipMemGetInfo through torch) is the same before and after a call of the
    no-alloc entry point, once the device's warm-up call (hipGetSymbolAddress) has been made; a
    stream of another device than the current one is refused (one-GPU box: not exercised)."""
    from gluonnlp_amd import _lib, _tensor as T
    lib = _lib.load()
    rng = np.random.default_rng(1)
    layer, _ = _layer(128, 256, 2, torch.bfloat16, rng)
    x = to_dev(rng.standard_normal((2, 64, 128)), torch.bfloat16)
    vl = torch.tensor([64, 9], device='cuda')
    out = torch.empty_like(x)
    cfg = layer.cfg_struct(T.code(x))
    p = layer.params_struct()
    stream = T.stream(x.device)
    call = lambda: lib.gnlp_encoder_layer_forward_noalloc(
        T.ptr(out), T.ptr(x), ctypes.byref(p), ctypes.byref(cfg), 2, 64, None, T.ptr(vl), _lib.I64,
        None, 1, stream)
    _lib.check(call())                      # warm-up: the address look-up of this device
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        _lib.check(call())
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_encoder_layer_noalloc_matches_workspace_form(dtype):
    from gluonnlp_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    U, F, H = 256, 512, 4          # 64-wide heads: bf16 takes the fused attention kernel
    layer, lp = _layer(U, F, H, dtype, rng)
    t = tol(dtype)
    # two calls in a row on the same static array: a larger input first, then a smaller one
    for (B, L, want_w) in [(6, 200, False), (2, 72, True), (3, 136, False)]:
        x = rng.standard_normal((B, L, U))
        vl_np = np.array([L] + [max(1, (L * (i + 1)) // (B + 1)) for i in range(B - 1)])
        xd = to_dev(x, dtype)
        vl = torch.from_numpy(vl_np).cuda()
        out_ws, w_ws, out_na, w_na = _both(lib, layer, xd, vl, want_w, True)
        assert torch.isfinite(out_na.float()).all()
        assert torch.equal(out_na, out_ws)                      # bit for bit
        if want_w:
            assert torch.equal(w_na, w_ws)
        mask = ref.gen_self_attn_mask((B, L), vl_np)
        exp, expw = ref.transformer_encoder_layer(prep(x, dtype), mask, oracle_params(lp, dtype),
                                                  H, 1e-12, False, 'gelu')
        exp = ref.sequence_mask(exp, vl_np, 1)
        assert rel_err(to_np(out_na), exp) < t
        keep = np.arange(L)[None, :] < vl_np[:, None]
        assert (to_np(out_na)[~keep] == 0).all()
        if want_w:
            assert rel_err(to_np(w_na), expw) < t


def test_encoder_layer_noalloc_refuses_oversize():
    """A layer whose scratch exceeds the static array is refused with a clear error; nothing runs."""
    from gluonnlp_amd import _lib, _tensor as T
    lib = _lib.load()
    rng = np.random.default_rng(78)
    U, F, H = 256, 512, 4
    layer, _ = _layer(U, F, H, torch.bfloat16, rng)
    B, L = 64, 512                                   # 32768 tokens: far beyond 64 MiB of scratch
    cfg = layer.cfg_struct(_lib.BF16)
    need = lib.gnlp_encoder_workspace_bytes(ctypes.byref(cfg), B, L, 0)
    assert need > _lib.NOALLOC_SCRATCH_BYTES
    x = torch.zeros((B, L, U), dtype=torch.bfloat16, device='cuda')
    out = torch.full_like(x, 7.0)
    p = layer.params_struct()
    rc = lib.gnlp_encoder_layer_forward_noalloc(
        T.ptr(out), T.ptr(x), ctypes.byref(p), ctypes.byref(cfg), B, L, None, None, 0, None, 0,
        T.stream(x.device))
    assert rc != 0
    msg = lib.gnlp_last_error().decode()
    assert 'static' in msg and 'scratch' in msg
    torch.cuda.synchronize()
    assert (out == 7.0).all()                        # refused before any launch
