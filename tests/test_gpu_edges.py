"""GPU edge cases (through the C ABI): empty batches, length-1 sequences, valid_length 0 and
valid_length > L, non-contiguous inputs, C-ABI argument errors."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import load_params, oracle_params, prep, rel_err, to_dev, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    from gluonnlp_amd import _lib
    _lib.load()
    return gluonnlp_amd


def test_empty_and_tiny_shapes(nlp):
    ac = nlp.attention_cell
    from gluonnlp_amd.models import BertTransformer
    vl0 = torch.zeros((0,), dtype=torch.int64, device='cuda')
    m = ac.gen_self_attn_mask(torch.zeros((0, 5, 4), device='cuda'), vl0)
    assert tuple(m.shape) == (0, 5, 5)
    st = BertTransformer(units=64, hidden_size=128, num_layers=1, num_heads=1, dtype='bfloat16').cuda()
    out, _ = st(torch.zeros((0, 7, 64), device='cuda', dtype=torch.bfloat16), vl0)
    assert tuple(out.shape) == (0, 7, 64)
    # L = 1, B = 1
    rng = np.random.default_rng(1)
    lp = ref.make_encoder_layer_params(rng, 64, 128)
    for dtype, name, tol in [(torch.float32, 'float32', 1e-5), (torch.bfloat16, 'bfloat16', 2e-2)]:
        st = BertTransformer(units=64, hidden_size=128, num_layers=1, num_heads=1, dtype=name)
        load_params(st.all_layers[0], lp, dtype)
        x = rng.standard_normal((1, 1, 64))
        out, _ = st(to_dev(x, dtype), torch.tensor([1], device='cuda'))
        exp, _ = ref.bert_transformer(prep(x, dtype), np.array([1]), [oracle_params(lp, dtype)], 1)
        assert rel_err(to_np(out), exp) < tol


@pytest.mark.parametrize('dtype,name,tol', [(torch.float32, 'float32', 1e-5),
                                            (torch.bfloat16, 'bfloat16', 2e-2)])
def test_valid_length_zero_and_beyond_L(nlp, dtype, name, tol):
    """vl = 0 -> every row of the sample is padding (exact zeros after BERT's sequence_mask);
    vl > L behaves like vl = L (the reference's `steps < valid_length` compare)."""
    from gluonnlp_amd.models import BertTransformer
    rng = np.random.default_rng(2)
    U, F, H, B, L = 128, 256, 2, 3, 70
    lp = [ref.make_encoder_layer_params(rng, U, F)]
    st = BertTransformer(units=U, hidden_size=F, num_layers=1, num_heads=H, dtype=name)
    load_params(st.all_layers[0], lp[0], dtype)
    x = rng.standard_normal((B, L, U))
    vl = np.array([0, L + 9, 33])
    out, _ = st(to_dev(x, dtype), torch.tensor(vl, device='cuda'))
    exp, _ = ref.bert_transformer(prep(x, dtype), vl, [oracle_params(lp[0], dtype)], H)
    got = to_np(out)
    assert np.isfinite(got).all()
    assert (got[0] == 0).all()
    assert rel_err(got, exp) < tol
    m = nlp.attention_cell.gen_self_attn_mask(to_dev(x, dtype), torch.tensor(vl, device='cuda'))
    assert np.array_equal(m.cpu().numpy(), ref.gen_self_attn_mask((B, L), vl))


def test_non_contiguous_inputs(nlp):
    ac = nlp.attention_cell
    rng = np.random.default_rng(3)
    B, H, L, d = 2, 2, 96, 64
    big = to_dev(rng.standard_normal((B, L, H, 2 * d)), torch.bfloat16)
    q = big[..., :d]                       # inner stride 1, row stride 2d: a strided view
    k = big[..., d:]
    v = to_dev(rng.standard_normal((B, L, H, d)), torch.bfloat16)
    ctx, _ = ac.multi_head_dot_attn(q, k, v, query_head_units=d, layout='NTK',
                                    _need_scores=False, _need_weights=False)
    e, _ = ref.multi_head_dot_attn(to_np(q), to_np(k), to_np(v), query_head_units=d, layout='NTK')
    assert rel_err(to_np(ctx), e) < 2e-2
    x = to_dev(rng.standard_normal((8, 6, 64)), torch.float32).transpose(0, 1)   # non-contiguous
    ln = nlp.layers.LayerNorm(in_channels=64).cuda()
    assert rel_err(to_np(ln(x)), ref.layer_norm(to_np(x), np.ones(64), np.zeros(64), 1e-5)) < 1e-5


def test_c_abi_argument_errors(nlp):
    from gluonnlp_amd import _lib
    lib = _lib.load()
    x = torch.zeros(4, 8, device='cuda')
    rc = lib.gnlp_activation(x.data_ptr(), x.data_ptr(), 7, 32, 1, None)        # bad dtype
    assert rc != 0 and b'dtype' in lib.gnlp_last_error()
    rc = lib.gnlp_activation(x.data_ptr(), x.data_ptr(), 0, 32, 99, None)       # bad activation
    assert rc != 0 and b'not supported' in lib.gnlp_last_error()
    rc = lib.gnlp_gen_self_attn_mask(x.data_ptr(), None, 0, 1, 2, 5, None)      # bad attn_type
    assert rc != 0
    rc = lib.gnlp_relative_position_bucket(x.data_ptr(), x.data_ptr(), 4, 1, 31, 128, None)
    assert rc != 0 and b'divisible by 2' in lib.gnlp_last_error()
    rc = lib.gnlp_dense(x.data_ptr(), x.data_ptr(), x.data_ptr(), None, None, 0, 0, 0, 4, 8, 8, 4,
                        8, 0, None)                                                # ldx < K
    assert rc != 0 and b'leading dimension' in lib.gnlp_last_error()
    cfg = _lib.EncoderCfg()
    cfg.units, cfg.hidden, cfg.heads, cfg.dtype = 10, 16, 3, 0
    rc = lib.gnlp_encoder_stack_forward(None, x.data_ptr(), None, 1, ctypes.byref(cfg), 1, 1,
                                        None, 0, 0, None, None, 0, None)
    assert rc != 0 and b'divisive' in lib.gnlp_last_error()
    with pytest.raises(_lib.GnlpError):
        _lib.check(rc)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_with_zero_key_units(nlp, dtype):
    """query / key of width 0 (the reference's batch_dot over an empty axis: all scores 0): the
    general MFMA kernel declines (d_k < 1), the unfused path must size its scores workspace itself
    (the workspace query used to answer 0 for it: a NULL scores buffer) -- uniform weights over the
    unmasked keys, context = their mean of V, masked rows exact zeros."""
    ac = nlp.attention_cell
    rng = np.random.default_rng(4)
    B, H, Lq, Lm, dv = 2, 3, 5, 7, 6
    q = torch.zeros(B, H, Lq, 0, dtype=dtype, device='cuda')
    k = torch.zeros(B, H, Lm, 0, dtype=dtype, device='cuda')
    vv = rng.standard_normal((B, H, Lm, dv))
    mask = rng.integers(0, 2, size=(B, Lq, Lm)).astype(np.int32)
    mask[0, 1, :] = 0
    mask[:, 0, 0] = 1
    ctx, (scores, w) = ac.multi_head_dot_attn(q, k, to_dev(vv, dtype), to_dev(mask, torch.int32),
                                              scaled=False, layout='NKT')
    from gluonnlp_amd import _lib
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_UNFUSED
    cnt = mask.sum(-1, keepdims=True)
    want_w = np.where(mask > 0, 1.0 / np.maximum(cnt, 1), 0.0)[:, None].repeat(H, 1)
    vq = to_np(to_dev(vv, dtype)).astype(np.float64)
    want = np.einsum('bhij,bhjd->bihd', want_w, vq).reshape(B, Lq, H * dv)
    t = 1e-5 if dtype == torch.float32 else 2e-2
    assert float(np.abs(to_np(scores)).max()) == 0.0
    assert rel_err(to_np(w), want_w) < t
    assert rel_err(to_np(ctx), want) < t
    assert float(np.abs(to_np(ctx)[0, 1]).max()) == 0.0

