"""The drop-in boundary itself: tensors enter by DLPack without a copy and leave in the
caller's framework; a mask tensor alone defines attention (reference
src/gluonnlp/models/transformer.py:234-265), whatever happens to the valid_length it was built
from; workspace sizing of the stack entry points follows the kernel the dispatcher really takes
(bf16 T5, d_kv = 64, L % 4 != 0)."""
import sys
import types

import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import load_params, oracle_params, prep, rel_err, to_dev, to_np, tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    from gluonnlp_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return gluonnlp_amd


class Box:
    """A minimal non-torch tensor: nothing but the DLPack protocol."""

    def __init__(self, t):
        self._t = t

    def __dlpack__(self, stream=None, **kw):
        return self._t.__dlpack__(stream=stream) if stream is not None else self._t.__dlpack__()

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()


def _fake_framework():
    """A stand-in for mx.np / cupy: a module-level from_dlpack returning its own array type."""
    mod = types.ModuleType('gnlp_fakefw')

    class Array(Box):
        pass

    Array.__module__ = 'gnlp_fakefw'

    def from_dlpack(x):
        return Array(torch.from_dlpack(x))
    mod.Array, mod.from_dlpack = Array, from_dlpack
    sys.modules['gnlp_fakefw'] = mod
    return mod


def test_dlpack_zero_copy_and_return_types(nlp):
    """multi_head_dot_attn, TransformerEncoderLayer.forward and BertTransformer.forward fed through
    objects exposing only __dlpack__ / __dlpack_device__: same device pointers (zero copies), same
    results as with torch tensors, outputs in the caller's framework when it has from_dlpack and
    torch tensors (documented contract, _tensor.like_caller) when it has none."""
    from gluonnlp_amd import _tensor as T
    from gluonnlp_amd.models import BertTransformer, TransformerEncoderLayer
    ac = nlp.attention_cell
    fw = _fake_framework()
    rng = np.random.default_rng(11)
    dtype = torch.bfloat16
    B, L, U, H, F = 3, 40, 128, 2, 256
    x = to_dev(rng.standard_normal((B, L, U)), dtype)
    vl = torch.tensor([L, 17, 1], device='cuda')
    # (0) as_tensor is a view of the same memory
    for box in (Box(x), fw.Array(x)):
        v = T.as_tensor(box)
        assert v.data_ptr() == x.data_ptr() and v.shape == x.shape and v.dtype == x.dtype
    # (1) multi_head_dot_attn
    q, k, v = [to_dev(rng.standard_normal((B, L, H, 64)), dtype) for _ in range(3)]
    mask = ac.gen_self_attn_mask(x, vl)
    want_ctx, (want_s, want_w) = ac.multi_head_dot_attn(q, k, v, mask, query_head_units=64,
                                                        layout='NTK')
    got_ctx, (got_s, got_w) = ac.multi_head_dot_attn(Box(q), Box(k), Box(v), Box(mask),
                                                     query_head_units=64, layout='NTK')
    assert isinstance(got_ctx, torch.Tensor) and isinstance(got_w, torch.Tensor)
    assert torch.equal(got_ctx, want_ctx) and torch.equal(got_w, want_w)
    fctx, (fs, fwts) = ac.multi_head_dot_attn(fw.Array(q), fw.Array(k), fw.Array(v),
                                              fw.Array(mask), query_head_units=64, layout='NTK')
    assert type(fctx) is fw.Array and type(fs) is fw.Array and type(fwts) is fw.Array
    assert torch.equal(torch.from_dlpack(fctx), want_ctx)
    # (2) TransformerEncoderLayer.forward(data, attn_mask)
    lp = ref.make_encoder_layer_params(rng, U, F)
    layer = load_params(TransformerEncoderLayer(U, F, H, activation='gelu', dtype='bfloat16'), lp,
                        dtype)
    want_o, want_aw = layer(x, mask)
    got_o, got_aw = layer(Box(x), Box(mask))
    assert isinstance(got_o, torch.Tensor) and torch.equal(got_o, want_o)
    assert torch.equal(got_aw, want_aw)
    fo, faw = layer(fw.Array(x), fw.Array(mask))
    assert type(fo) is fw.Array and type(faw) is fw.Array
    assert torch.equal(torch.from_dlpack(fo), want_o)
    # against the oracle as well (not only against itself)
    exp, _ = ref.transformer_encoder_layer(prep(to_np(x), dtype), ref.gen_self_attn_mask((B, L),
                                           vl.cpu().numpy()), oracle_params(lp, dtype), H, 1e-12,
                                           False, 'gelu')
    valid = np.arange(L)[None, :] < vl.cpu().numpy()[:, None]
    assert rel_err(to_np(got_o)[valid], exp[valid]) < tol(dtype)
    # (3) BertTransformer.forward(data, valid_length)
    m = BertTransformer(units=U, hidden_size=F, num_layers=2, num_heads=H, dtype='bfloat16')
    for ly in m.all_layers:
        load_params(ly, lp, dtype)
    m.cuda()
    want_b = m(x, vl)[0]
    got_b = m(Box(x), Box(vl))[0]
    assert isinstance(got_b, torch.Tensor) and torch.equal(got_b, want_b)
    fb = m(fw.Array(x), fw.Array(vl))[0]
    assert type(fb) is fw.Array and torch.equal(torch.from_dlpack(fb), want_b)


def test_like_caller_does_not_swallow_errors(nlp):
    from gluonnlp_amd import _tensor as T
    mod = types.ModuleType('gnlp_brokenfw')

    class Arr(Box):
        pass

    Arr.__module__ = 'gnlp_brokenfw'

    def from_dlpack(x):
        raise ValueError('conversion failed')
    mod.from_dlpack = from_dlpack
    sys.modules['gnlp_brokenfw'] = mod
    t = torch.zeros(4, device='cuda')
    with pytest.raises(ValueError, match='conversion failed'):
        T.like_caller(t, Arr(t))
    with pytest.raises(TypeError):
        T.as_tensor(object())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_stale_valid_length_does_not_change_the_mask(nlp, dtype):
    """Build the mask, then edit valid_length IN PLACE (the caller reusing its buffer for the next
    batch): the layer must attend by the mask bytes it was handed, not by the new lengths
    (VERDICT r01 'stale-tag hazard').  bf16 with 64-wide heads takes the in-kernel predicate."""
    from gluonnlp_amd.models import TransformerEncoderLayer
    ac = nlp.attention_cell
    rng = np.random.default_rng(23)
    B, L, U, H, F = 4, 48, 128, 2, 256
    x = rng.standard_normal((B, L, U))
    vl0 = np.array([L - 1, 20, 7, 1])
    lp = ref.make_encoder_layer_params(rng, U, F)
    name = 'bfloat16' if dtype == torch.bfloat16 else 'float32'
    layer = load_params(TransformerEncoderLayer(U, F, H, activation='gelu', dtype=name), lp, dtype)
    xd = to_dev(x, dtype)
    vld = torch.from_numpy(vl0).cuda()
    mask = ac.gen_self_attn_mask(xd, vld)
    mask_bytes = mask.clone()                 # an untagged copy: the dense byte-mask path
    vld.add_(1)                               # stale: the tag must not follow this
    assert torch.equal(mask, mask_bytes)
    out_tagged, _ = layer(xd, mask, _need_weights=False)
    out_dense, _ = layer(xd, mask_bytes, _need_weights=False)
    exp, _ = ref.transformer_encoder_layer(prep(x, dtype), ref.gen_self_attn_mask((B, L), vl0),
                                           oracle_params(lp, dtype), H, 1e-12, False, 'gelu')
    valid = np.arange(L)[None, :] < vl0[:, None]
    t = tol(dtype)
    assert rel_err(to_np(out_tagged)[valid], exp[valid]) < t
    assert rel_err(to_np(out_dense)[valid], exp[valid]) < t
    # the functional form too
    q, k, v = [to_dev(rng.standard_normal((B, L, H, 64)), dtype) for _ in range(3)]
    ctx, _ = ac.multi_head_dot_attn(q, k, v, mask, query_head_units=64, layout='NTK',
                                    _need_scores=False, _need_weights=False)
    ectx, _ = ref.multi_head_dot_attn(prep(to_np(q), dtype), prep(to_np(k), dtype),
                                      prep(to_np(v), dtype), ref.gen_self_attn_mask((B, L), vl0),
                                      query_head_units=64, layout='NTK')
    assert rel_err(to_np(ctx), ectx) < t
    # an in-place edit of the MASK drops the tag: the bytes rule
    mask[:, :, 0] = False
    ctx2, _ = ac.multi_head_dot_attn(q, k, v, mask, query_head_units=64, layout='NTK',
                                     _need_scores=False, _need_weights=False)
    ectx2, _ = ref.multi_head_dot_attn(prep(to_np(q), dtype), prep(to_np(k), dtype),
                                       prep(to_np(v), dtype), mask.cpu().numpy(),
                                       query_head_units=64, layout='NTK')
    assert rel_err(to_np(ctx2), ectx2) < t


@pytest.mark.parametrize('L', [97, 511])
def test_t5_encoder_bf16_d64_odd_length(nlp, L):
    """bf16 T5 with d_kv = 64 and L % 4 != 0, H * L > d_model + d_ff: the workspace the C ABI
    reports must fit the kernel path the dispatcher takes (api.hip t5_ws), and the
    fused kernel reads the last 1-3 bias columns of a row one by one."""
    from gluonnlp_amd.models import T5Encoder
    dtype = torch.bfloat16
    d_model, d_kv, d_ff, H, B = 128, 64, 256, 4, 2
    assert H * L > d_model + d_ff or L == 97
    rng = np.random.default_rng(L)
    lp = [ref.make_t5_block_params(rng, d_model, d_kv, d_ff, H, False) for _ in range(2)]
    relw = (rng.standard_normal((32, H)) * d_model ** -0.5).astype(np.float32)
    fin = {'gamma': (1 + 0.1 * rng.standard_normal(d_model)).astype(np.float32),
           'beta': np.zeros(d_model, np.float32)}
    x = rng.standard_normal((B, L, d_model))
    vl = np.array([L, max(1, L // 3)])
    exp = ref.t5_encoder(prep(x, dtype), vl, prep(relw, dtype),
                         [oracle_params(p, dtype) for p in lp], oracle_params(fin, dtype), H,
                         activation_name='relu')
    enc = T5Encoder(d_model, d_kv, d_ff, num_layers=2, num_heads=H, activation='relu',
                    dtype='bfloat16')
    for i in range(2):
        load_params(enc.layers[i], lp[i], dtype)
    load_params(enc.final_layer_norm, fin, dtype)
    load_params(enc.relative_position_encoder, {'_rel_pos_embed.weight': relw}, dtype)
    enc.cuda()
    got = to_np(enc(to_dev(x, dtype), torch.tensor(vl, device='cuda')))
    valid = np.arange(L)[None, :] < vl[:, None]
    assert rel_err(got[valid], exp[valid]) < tol(dtype)
    # the functional attention with a dense (H, L, L) bias of odd row length, fused path
    ac = nlp.attention_cell
    q, k, v = [to_dev(rng.standard_normal((B, L, H, 64)), dtype) for _ in range(3)]
    bias = to_dev(rng.standard_normal((H, L, L)), dtype)
    ctx, _ = ac.multi_head_dot_attn(q, k, v, None, bias, scaled=False, layout='NTK',
                                    _need_scores=False, _need_weights=False)
    ectx, _ = ref.multi_head_dot_attn(prep(to_np(q), dtype), prep(to_np(k), dtype),
                                      prep(to_np(v), dtype), None, prep(to_np(bias), dtype),
                                      scaled=False, layout='NTK')
    assert rel_err(to_np(ctx), ectx) < tol(dtype)


def test_workspace_is_per_stream(nlp):
    """Two streams never share scratch: the grow-only buffer is keyed by stream."""
    from gluonnlp_amd import _tensor as T
    dev = torch.device('cuda', 0)
    a = T.workspace(dev, 1024)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b = T.workspace(dev, 1024)
    assert a.data_ptr() != b.data_ptr()
    assert T.workspace(dev, 512).data_ptr() == a.data_ptr()
