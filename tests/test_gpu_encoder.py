"""GPU parity tests (through the C ABI) for attention, PositionwiseFFN, TransformerEncoderLayer,
the BERT / RoBERTa / ELECTRA / TransformerEncoder stacks and the T5 encoder, against the committed
golden fixtures (generated from the reference's torch sibling) and the CPU oracle.
fp32 within 1e-5, bf16 within 2e-2 (max-abs relative); masks / zeros exact."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import (load_params, oracle_params, prep, rel_err, to_dev, to_np, tol)

pytestmark = pytest.mark.gpu

from gluonnlp_amd import _lib  # noqa: E402

G = os.path.join(os.path.dirname(__file__), 'golden')
DT = {torch.float32: 'float32', torch.bfloat16: 'bfloat16'}


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    from gluonnlp_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return gluonnlp_amd


def _sub(g, prefix):
    n = len(prefix)
    return {k[n:]: g[k] for k in g.files if k.startswith(prefix)}


# ----------------------------------------------------------------------------- attention
@pytest.mark.parametrize('H', [1, 2, 3])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_goldens(nlp, H, dtype):
    """reference tests/test_attention_cell.py:14-151: B=5, Lq=16, Lm=32, d_k=8, d_v=6, random 0/1
    mask, shared / per-head / no edge scores, all three layouts must agree with the golden."""
    ac = nlp.attention_cell
    g = np.load(os.path.join(G, 'attn_h{}.npz'.format(H)))
    q, k, v, mask = g['q'], g['k'], g['v'], g['mask']
    t = tol(dtype)
    for scaled in [True, False]:
        for ename in ['none', 'share', 'head']:
            e = None if ename == 'none' else g['e_' + ename]
            key = 's{}_{}'.format(int(scaled), ename)
            if dtype == torch.float32:
                exp_ctx, exp_s, exp_w = g[key + '_ctx'], g[key + '_scores'], g[key + '_w']
            else:
                exp_ctx, (exp_s, exp_w) = ref.multi_head_dot_attn(
                    prep(q, dtype), prep(k, dtype), prep(v, dtype), mask,
                    None if e is None else prep(e, dtype), scaled=scaled, query_head_units=8)
            for layout, perm, back in [('NKT', None, None), ('NTK', (0, 2, 1, 3), None),
                                       ('TNK', (2, 0, 1, 3), (1, 0, 2))]:
                qq, kk, vv = [to_dev(x if perm is None else np.transpose(x, perm), dtype)
                              for x in (q, k, v)]
                cell = ac.MultiHeadAttentionCell(query_units=8 * H, num_heads=H, scaled=scaled,
                                                 layout=layout)
                ctx, (scores, w) = cell(qq, kk, vv, torch.tensor(mask, device='cuda'),
                                        None if e is None else to_dev(e, dtype))
                # d_k = 8 / d_v = 6, float32 and bf16: the MFMA kernel, not the VALU path
                assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_GENERAL_MFMA
                ctx = to_np(ctx)
                if back is not None:
                    ctx = np.transpose(ctx, back)
                assert rel_err(ctx, exp_ctx) < t
                assert rel_err(to_np(scores), exp_s) < t
                assert rel_err(to_np(w), exp_w) < t
                wn = to_np(w)
                for h in range(H):
                    assert (wn[:, h][mask == 0] == 0).all()   # masked weights are exactly 0
                assert (ctx[1] == 0).all()                      # fully masked sample
    ctx, (_, w) = ac.multi_head_dot_attn(to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype),
                                         query_head_units=8)
    if dtype == torch.float32:
        assert rel_err(to_np(ctx), g['nomask_ctx']) < t
        assert rel_err(to_np(w), g['nomask_w']) < t


def test_attention_errors_and_normalized(nlp):
    ac = nlp.attention_cell
    q = torch.randn(2, 2, 4, 8, device='cuda')
    with pytest.raises(NotImplementedError):
        ac.multi_head_dot_attn(q, q, q, scaled=True, query_head_units=None)
    with pytest.raises(NotImplementedError):
        ac.multi_head_dot_attn(q, q, q, scaled=False, layout='KNT')
    with pytest.raises(AssertionError):
        ac.MultiHeadAttentionCell(query_units=10, num_heads=3)
    ctx, (scores, w) = ac.multi_head_dot_attn(q, q, q, scaled=True, normalized=True,
                                              query_head_units=8, eps=1e-6)
    qo = to_np(q)
    ectx, (es, ew) = ref.multi_head_dot_attn(qo, qo, qo, scaled=True, normalized=True,
                                             query_head_units=8, eps=1e-6)
    assert rel_err(to_np(ctx), ectx) < 1e-5 and rel_err(to_np(scores), es) < 1e-5


@pytest.mark.parametrize('layout', ['NKT', 'NTK', 'TNK'])
@pytest.mark.parametrize('mode', ['vl', 'dense', 'none', 'bias_vl', 'bias_dense'])
@pytest.mark.parametrize('Lq,Lm', [(128, 128), (200, 200), (64, 333), (512, 512)])
def test_fused_attention_bf16_d64(nlp, layout, mode, Lq, Lm):
    """The fused MFMA kernel (bf16, d = 64) against the float64 oracle on bf16-rounded inputs, and
    against the unfused path (which also returns scores / weights)."""
    if mode in ('vl', 'bias_vl') and Lq != Lm:
        pytest.skip('valid_length predicate is a self-attention mask')
    ac = nlp.attention_cell
    rng = np.random.default_rng(Lq * 7 + Lm)
    B, H, d = 3, 2, 64
    q = rng.standard_normal((B, H, Lq, d))
    k = rng.standard_normal((B, H, Lm, d))
    v = rng.standard_normal((B, H, Lm, d))
    # spike one key against one query so that the running max jumps late (online-softmax rescale)
    k[0, 0, Lm - 3] = q[0, 0, 5] * 3.0
    vl = np.array([Lm, 1, max(2, Lm * 3 // 5)])
    mask = None
    kw = {}
    if mode in ('vl', 'bias_vl'):
        mask_np = ref.gen_self_attn_mask((B, Lq), vl)
        data = torch.zeros((B, Lq, 1), device='cuda')
        mask = ac.gen_self_attn_mask(data, torch.tensor(vl, device='cuda'))   # tagged -> predicate
    elif mode in ('dense', 'bias_dense'):
        mask_np = rng.integers(0, 2, size=(B, Lq, Lm)).astype(bool)
        mask_np[0, 3, :] = False
        mask_np[2, :, Lm // 2:] = False
        mask = torch.tensor(mask_np, device='cuda')
    else:
        mask_np = None
    e = rng.standard_normal((H, Lq, Lm)) if mode.startswith('bias') else None
    perm = {'NKT': None, 'NTK': (0, 2, 1, 3), 'TNK': (2, 0, 1, 3)}[layout]
    qq, kk, vv = [to_dev(x if perm is None else np.transpose(x, perm), torch.bfloat16)
                  for x in (q, k, v)]
    ed = None if e is None else to_dev(e, torch.bfloat16)
    ctx, (s_, w_) = ac.multi_head_dot_attn(qq, kk, vv, mask, ed, query_head_units=d, layout=layout,
                                           _need_scores=False, _need_weights=False)
    assert s_ is None and w_ is None
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_FUSED_BF16
    ectx, _ = ref.multi_head_dot_attn(prep(q, torch.bfloat16), prep(k, torch.bfloat16),
                                      prep(v, torch.bfloat16), mask_np,
                                      None if e is None else prep(e, torch.bfloat16),
                                      query_head_units=d)
    got = to_np(ctx)
    if layout == 'TNK':
        got = np.transpose(got, (1, 0, 2))
    assert rel_err(got, ectx) < 2e-2
    if mask_np is not None:
        dead = ~mask_np.any(-1)                     # fully masked query rows give exact zeros
        assert (got[dead] == 0).all()
    ctx2, (s2, w2) = ac.multi_head_dot_attn(qq, kk, vv, mask, ed, query_head_units=d,
                                            layout=layout)
    got2 = to_np(ctx2)
    if layout == 'TNK':
        got2 = np.transpose(got2, (1, 0, 2))
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_GENERAL_MFMA
    assert rel_err(got2, ectx) < 2e-2
    assert s2 is not None and w2 is not None
    _, (es, ew) = ref.multi_head_dot_attn(prep(q, torch.bfloat16), prep(k, torch.bfloat16),
                                          prep(v, torch.bfloat16), mask_np,
                                          None if e is None else prep(e, torch.bfloat16),
                                          query_head_units=d)
    assert rel_err(to_np(s2), es) < 2e-2 and rel_err(to_np(w2), ew) < 2e-2
    # the same tensors in float32 through the same kernel, at the fp32 tolerance
    q3, k3, v3 = [t_.float() for t_ in (qq, kk, vv)]
    ctx3, (s3, w3) = ac.multi_head_dot_attn(q3, k3, v3, mask, None if ed is None else ed.float(),
                                            query_head_units=d, layout=layout)
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_GENERAL_MFMA
    got3 = to_np(ctx3)
    if layout == 'TNK':
        got3 = np.transpose(got3, (1, 0, 2))
    assert rel_err(got3, ectx) < 1e-5
    assert rel_err(to_np(s3), es) < 1e-5 and rel_err(to_np(w3), ew) < 1e-5
    ctx4, _ = ac.multi_head_dot_attn(q3, k3, v3, mask, None if ed is None else ed.float(),
                                     query_head_units=d, layout=layout, _need_scores=False,
                                     _need_weights=False)       # one-pass form
    got4 = to_np(ctx4)
    if layout == 'TNK':
        got4 = np.transpose(got4, (1, 0, 2))
    assert rel_err(got4, ectx) < 1e-5


def test_fused_attention_from_packed_qkv(nlp):
    """q / k / v as strided views of one (B, L, 3U) buffer (how the encoder layer calls it)."""
    ac = nlp.attention_cell
    rng = np.random.default_rng(8)
    B, L, H, d = 2, 160, 4, 64
    U = H * d
    qkv = rng.standard_normal((B, L, 3 * U))
    t = to_dev(qkv, torch.bfloat16)
    q, k, v = [t[..., i * U:(i + 1) * U].reshape(B, L, H, d) for i in range(3)]
    qo, ko, vo = [prep(qkv[..., i * U:(i + 1) * U].reshape(B, L, H, d), torch.bfloat16)
                  for i in range(3)]
    vl = np.array([160, 37])
    ctx, _ = ac.multi_head_dot_attn(q, k, v, None, None, query_head_units=d, layout='NTK',
                                    _need_scores=False, _need_weights=False,
                                    _valid_length=torch.tensor(vl, device='cuda'))
    ectx, _ = ref.multi_head_dot_attn(qo, ko, vo, ref.gen_self_attn_mask((B, L), vl),
                                      query_head_units=d, layout='NTK')
    assert rel_err(to_np(ctx), ectx) < 2e-2
    assert (to_np(ctx)[1, 37:] == 0).all()


# ----------------------------------------------------------------------------- FFN
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('name,kw,okw', [
    ('postnorm_relu', dict(activation='relu', layer_norm_eps=1e-5), {}),
    ('postnorm_gelu', dict(activation='gelu', layer_norm_eps=1e-12), {}),
    ('prenorm_gelu', dict(activation='gelu', pre_norm=True, layer_norm_eps=1e-5), {}),
    ('gated_gelu_tanh', dict(activation='gelu(tanh)', use_gated_activation=True,
                             layer_norm_eps=1e-5), {}),
])
def test_ffn_goldens(nlp, dtype, name, kw, okw):
    g = np.load(os.path.join(G, 'ffn.npz'))
    p = _sub(g, name + '/')
    ffn = nlp.layers.PositionwiseFFN(units=48, hidden_size=64, dtype=DT[dtype], **kw)
    load_params(ffn, p, dtype)
    x = g[name + '_x']
    got = to_np(ffn(to_dev(x, dtype)))
    if dtype == torch.float32:
        exp = g[name + '_y']
    else:
        exp = ref.positionwise_ffn(prep(x, dtype), oracle_params(p, dtype),
                                   activation_name=kw['activation'],
                                   use_gated_activation=kw.get('use_gated_activation', False),
                                   layer_norm_eps=kw['layer_norm_eps'],
                                   pre_norm=kw.get('pre_norm', False))
    assert rel_err(got, exp) < tol(dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_ffn_rms_nobias_gated(nlp, dtype):
    """T5-style FFN: pre-norm RMS, no bias, gated gelu(tanh) (layers.py:151-181,636-638)."""
    rng = np.random.default_rng(12)
    U, F = 128, 512
    p = {k[4:]: v for k, v in ref.make_t5_block_params(rng, U, 64, F, 2, True).items()
         if k.startswith('ffn.')}
    ffn = nlp.layers.PositionwiseFFN(units=U, hidden_size=F, use_bias=False,
                                     activation='gelu(tanh)', use_gated_activation=True,
                                     normalization='rms_norm', pre_norm=True, dtype=DT[dtype],
                                     center=False, scale=True)
    load_params(ffn, p, dtype)
    x = rng.standard_normal((4, 9, U))
    exp = ref.positionwise_ffn(prep(x, dtype), oracle_params(p, dtype), 'gelu(tanh)', True,
                               'rms_norm', 1e-6, True)
    assert rel_err(to_np(ffn(to_dev(x, dtype))), exp) < tol(dtype)


# ----------------------------------------------------------------------------- encoder layer / stacks
CASES = [('enc_tiny_eps1e-12', 48, 64, 2, 1e-12, 'gelu'), ('enc_tiny_eps1e-5', 48, 64, 2, 1e-5, 'gelu'),
         ('enc_tiny_relu_vl1', 32, 64, 4, 1e-5, 'relu'), ('enc_d64', 128, 512, 2, 1e-12, 'gelu')]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('name,U,F,H,eps,act', CASES)
def test_encoder_layer_and_stacks(nlp, dtype, name, U, F, H, eps, act):
    from gluonnlp_amd.models import (BertTransformer, ElectraEncoder, RobertaEncoder,
                                     TransformerEncoderLayer)
    ac = nlp.attention_cell
    g = np.load(os.path.join(G, name + '.npz'))
    x, vl = g['x'], g['valid_length']
    B, L = x.shape[:2]
    t = tol(dtype)
    lp = [_sub(g, 'L0/'), _sub(g, 'L1/')]
    if dtype == torch.float32:
        exp0, expw, exp1, expm = g['out_L0'], g['attn_w_L0'], g['out_L1'], g['out_seqmask']
    else:
        op = [oracle_params(p, dtype) for p in lp]
        xo = prep(x, dtype)
        exp0, expw = ref.transformer_encoder_layer(xo, ref.gen_self_attn_mask(xo, vl), op[0], H,
                                                   eps, False, act)
        exp1, _ = ref.roberta_encoder(xo, vl, op, H, eps, act)
        expm, _ = ref.bert_transformer(xo, vl, op, H, eps, act)
    xd = to_dev(x, dtype)
    vld = torch.tensor(vl, device='cuda')
    # --- single layer through the reference signature forward(data, attn_mask)
    layer = TransformerEncoderLayer(units=U, hidden_size=F, num_heads=H, layer_norm_eps=eps,
                                    activation=act, dtype=DT[dtype])
    load_params(layer, lp[0], dtype)
    mask = ac.gen_self_attn_mask(xd, vld)
    out, w = layer(xd, mask)
    assert rel_err(to_np(out), exp0) < t
    assert rel_err(to_np(w), expw) < t
    # an untagged (dense) copy of the same mask must give the same answer
    out_d, _ = layer(xd, mask.clone(), _need_weights=False)
    assert rel_err(to_np(out_d), exp0) < t
    # --- TN layout == NT transposed (reference tests/test_models_bert.py:51-54)
    layer_tn = TransformerEncoderLayer(units=U, hidden_size=F, num_heads=H, layer_norm_eps=eps,
                                       activation=act, dtype=DT[dtype], layout='TN')
    load_params(layer_tn, lp[0], dtype)
    xt = xd.transpose(0, 1).contiguous()
    out_tn, w_tn = layer_tn(xt, ac.gen_self_attn_mask(xt, vld, layout='TN'))
    assert rel_err(np.swapaxes(to_np(out_tn), 0, 1), exp0) < t
    assert rel_err(to_np(w_tn), expw) < t
    # --- stacks
    for cls, exp, seqm in [(BertTransformer, expm, True), (ElectraEncoder, expm, True),
                           (RobertaEncoder, exp1, False)]:
        for layout in ['NT', 'TN']:
            st = cls(units=U, hidden_size=F, num_layers=2, num_heads=H, layer_norm_eps=eps,
                     activation=act, dtype=DT[dtype], layout=layout)
            attr = 'all_encoder_layers' if cls is ElectraEncoder else 'all_layers'
            for i in range(2):
                load_params(getattr(st, attr)[i], lp[i], dtype)
            xin = xd if layout == 'NT' else xt
            o, extra = st(xin, vld)
            o = to_np(o)
            if layout == 'TN':
                o = np.swapaxes(o, 0, 1)
            assert extra == []
            assert rel_err(o, exp) < t
            if seqm:
                for b, n in enumerate(vl):
                    assert (o[b, n:] == 0).all()       # exact zeros beyond valid_length
    # output_attention / output_all_encodings
    st = BertTransformer(units=U, hidden_size=F, num_layers=2, num_heads=H, layer_norm_eps=eps,
                         activation=act, dtype=DT[dtype], output_attention=True,
                         output_all_encodings=True)
    for i in range(2):
        load_params(st.all_layers[i], lp[i], dtype)
    enc, att = st(xd, vld)
    assert len(enc) == 2 and len(att) == 2
    assert rel_err(to_np(enc[1]), expm) < t
    assert rel_err(to_np(att[0]), expw) < t
    st2 = BertTransformer(units=U, hidden_size=F, num_layers=2, num_heads=H, layer_norm_eps=eps,
                          activation=act, dtype=DT[dtype], output_all_encodings=True)
    for i in range(2):
        load_params(st2.all_layers[i], lp[i], dtype)
    enc2, _ = st2(xd, vld)
    assert rel_err(to_np(enc2[1]), expm) < t
    assert rel_err(to_np(enc2[0]), ref.sequence_mask(np.asarray(exp0), vl, 1)) < t


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('pre_norm', [False, True])
def test_transformer_encoder(nlp, dtype, pre_norm):
    """reference tests/test_models_transformer.py:18-57: units=32, hidden=64, 4 heads, B=8, L=20,
    valid_length in [1, 20); pre_norm follows the MX text (normalised residual)."""
    from gluonnlp_amd.models import TransformerEncoder
    rng = np.random.default_rng(31)
    U, F, H, B, L, NL = 32, 64, 4, 8, 20, 3
    lp = [ref.make_encoder_layer_params(rng, U, F) for _ in range(NL)]
    fin = {'gamma': (1 + 0.1 * rng.standard_normal(U)).astype(np.float32),
           'beta': (0.1 * rng.standard_normal(U)).astype(np.float32)}
    x = rng.standard_normal((B, L, U))
    vl = rng.integers(1, L, size=B)
    enc = TransformerEncoder(num_layers=NL, units=U, hidden_size=F, num_heads=H, pre_norm=pre_norm,
                             layer_norm_eps=1e-5, activation='relu', dtype=DT[dtype])
    for i in range(NL):
        load_params(enc.layers[i], lp[i], dtype)
    if pre_norm:
        load_params(enc.ln_final, fin, dtype)
    enc.cuda()
    exp = ref.transformer_encoder(prep(x, dtype), vl, [oracle_params(p, dtype) for p in lp], H,
                                  1e-5, pre_norm, 'relu', 'NT', ln_final=oracle_params(fin, dtype))
    got = to_np(enc(to_dev(x, dtype), torch.tensor(vl, device='cuda')))
    valid = np.arange(L)[None, :] < vl[:, None]
    assert rel_err(got[valid], exp[valid]) < tol(dtype)


def test_padding_invariance_and_large_layer(nlp):
    """Full-width properties (BERT-large layer shapes, L=512): appending padding leaves the valid
    outputs unchanged; padded outputs are exactly zero; bf16 stays within 2e-2 of the oracle."""
    from gluonnlp_amd.models import BertTransformer
    rng = np.random.default_rng(77)
    U, F, H, B, L = 1024, 4096, 16, 2, 512
    lp = [ref.make_encoder_layer_params(rng, U, F)]
    x = rng.standard_normal((B, L, U)).astype(np.float32)
    vl = np.array([512, 301])
    st = BertTransformer(units=U, hidden_size=F, num_layers=1, num_heads=H, dtype='bfloat16')
    load_params(st.all_layers[0], lp[0], torch.bfloat16)
    xd = to_dev(x, torch.bfloat16)
    out, _ = st(xd, torch.tensor(vl, device='cuda'))
    exp, _ = ref.bert_transformer(prep(x, torch.bfloat16).astype(np.float32), vl,
                                  [{k: prep(v, torch.bfloat16).astype(np.float32)
                                    for k, v in lp[0].items()}], H)
    got = to_np(out)
    assert rel_err(got, exp) < 2e-2
    assert (got[1, 301:] == 0).all()
    x2 = x[:, :384].copy()
    vl2 = np.array([384, 301])
    out2, _ = st(to_dev(x2, torch.bfloat16), torch.tensor(vl2, device='cuda'))
    got2 = to_np(out2)
    assert rel_err(got2[1, :301], got[1, :301]) < 2e-2


# ----------------------------------------------------------------------------- T5
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('act', ['relu', 'gated-gelu'])
@pytest.mark.parametrize('d_model,d_kv,d_ff,H,B,L', [(128, 64, 512, 4, 3, 96), (32, 8, 64, 4, 3, 12)])
def test_t5_encoder(nlp, dtype, act, d_model, d_kv, d_ff, H, B, L):
    """reference tests/test_models_t5.py:19-67 shapes (d_model=128, d_ff=512, H=4, d_kv=64)."""
    from gluonnlp_amd.models import T5Encoder
    rng = np.random.default_rng(d_model + L)
    lp = [ref.make_t5_block_params(rng, d_model, d_kv, d_ff, H, act != 'relu') for _ in range(2)]
    relw = (rng.standard_normal((32, H)) * d_model ** -0.5).astype(np.float32)
    fin = {'gamma': (1 + 0.1 * rng.standard_normal(d_model)).astype(np.float32),
           'beta': np.zeros(d_model, np.float32)}
    x = rng.standard_normal((B, L, d_model))
    vl = np.array([L, max(1, L // 3), 1])
    exp = ref.t5_encoder(prep(x, dtype), vl, prep(relw, dtype),
                         [oracle_params(p, dtype) for p in lp], oracle_params(fin, dtype), H,
                         activation_name=act)
    for layout in ['NT', 'TN']:
        enc = T5Encoder(d_model, d_kv, d_ff, num_layers=2, num_heads=H, activation=act,
                        dtype=DT[dtype], layout=layout)
        for i in range(2):
            load_params(enc.layers[i], lp[i], dtype)
        load_params(enc.final_layer_norm, fin, dtype)
        load_params(enc.relative_position_encoder,
                    {'_rel_pos_embed.weight': relw}, dtype)
        enc.cuda()
        xin = to_dev(x, dtype)
        if layout == 'TN':
            xin = xin.transpose(0, 1).contiguous()
        got = to_np(enc(xin, torch.tensor(vl, device='cuda')))
        if layout == 'TN':
            got = np.swapaxes(got, 0, 1)
        valid = np.arange(L)[None, :] < vl[:, None]
        assert rel_err(got[valid], exp[valid]) < tol(dtype)
    # single block through the reference signature (mask tensor + position embeddings)
    ac = nlp.attention_cell
    blk = enc.layers[0]
    blk_nt = type(blk)(d_model, d_kv, d_ff, False, num_heads=H, activation=act, dtype=DT[dtype])
    load_params(blk_nt, lp[0], dtype)
    xd = to_dev(x, dtype)
    pos = enc.relative_position_encoder(ac.gen_rel_position(xd))
    mask = ac.gen_self_attn_mask(xd, torch.tensor(vl, device='cuda'))
    pe = ref.rel_attention_score_t5(prep(relw, dtype), ref.gen_rel_position((B, L)))
    eb = ref.t5_block(prep(x, dtype), ref.gen_self_attn_mask((B, L), vl), pe,
                      oracle_params(lp[0], dtype), H, activation_name=act)
    valid = np.arange(L)[None, :] < vl[:, None]
    gb = to_np(blk_nt(xd, mask, pos))
    assert rel_err(gb[valid], eb[valid]) < tol(dtype)
    gb2 = to_np(blk_nt(xd, mask.clone(), pos))          # dense-mask path
    assert rel_err(gb2[valid], eb[valid]) < tol(dtype)


def test_rccl_gather_path_single_rank(nlp, tmp_path):
    """The N>1 code path of parallel.BatchShardedEncoder (RCCL all_gather_into_tensor of the CLS
    rows on a side stream) on a one-rank process group: the collective really runs on the GPU."""
    import torch.distributed as dist
    from gluonnlp_amd.models import BertTransformer
    from gluonnlp_amd.parallel import BatchShardedEncoder
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    dist.init_process_group('nccl', init_method='file://' + str(tmp_path / 'rdv'), rank=0,
                            world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(0)
        m = BertTransformer(units=128, hidden_size=256, num_layers=2, num_heads=2,
                            dtype='bfloat16').cuda()
        x = torch.randn(3, 40, 128, device='cuda').bfloat16()
        vl = torch.tensor([40, 7, 21], device='cuda')
        ref_out, ref_cls = BatchShardedEncoder(m).forward_local(x, vl)
        enc = BatchShardedEncoder(m, world_size=1, rank=0, force_collective=True)
        for _ in range(3):   # the side stream is reused across steps
            out, cls_all = enc.forward_local(x, vl)
        torch.cuda.synchronize()
        assert torch.equal(out, ref_out)
        assert torch.equal(cls_all, ref_cls)
        dist.barrier()
    finally:
        dist.destroy_process_group()


# --------------------------------------------------------------------- causal predicate
@pytest.mark.parametrize('dtype,d', [(torch.bfloat16, 64), (torch.float32, 64), (torch.float32, 8),
                                     (torch.bfloat16, 128)])
@pytest.mark.parametrize('L', [200, 512, 31])
@pytest.mark.parametrize('use_vl,use_bias', [(True, False), (False, True), (True, True)])
def test_causal_predicate_matches_mask_bytes(nlp, dtype, d, L, use_vl, use_bias):
    """A mask straight from gen_self_attn_mask(attn_type='causal') is evaluated in-kernel as
    (j <= i) & (i < valid_length) (attention_cell.py:168-180), key tiles above the diagonal are
    skipped: same answers as the same bytes handed over untagged, and as the oracle."""
    from gluonnlp_amd import _lib
    ac = nlp.attention_cell
    rng = np.random.default_rng(L + d)
    B, H = 3, 2
    q, k, v = [rng.standard_normal((B, L, H, d)) for _ in range(3)]
    vl = np.array([L, 1, max(2, L * 3 // 5)]) if use_vl else None
    e = rng.standard_normal((H, L, L)) if use_bias else None
    data = torch.zeros((B, L, 1), device='cuda')
    mask = ac.gen_self_attn_mask(data, None if vl is None else torch.tensor(vl, device='cuda'),
                                 attn_type='causal')
    assert ac.mask_predicate(mask) is None                      # opt-in only
    assert ac.mask_predicate(mask, causal=True)[0] == 'causal'
    dense = mask.clone()
    assert np.array_equal(dense.cpu().numpy(), ref.gen_self_attn_mask((B, L), vl, attn_type='causal'))
    qq, kk, vv = [to_dev(x, dtype) for x in (q, k, v)]
    ee = None if e is None else to_dev(e, dtype)
    kw = dict(scaled=True, query_head_units=d, layout='NTK')
    fused = dtype == torch.bfloat16 and d == 64
    a, _ = ac.multi_head_dot_attn(qq, kk, vv, mask, ee, _need_scores=False, _need_weights=False, **kw)
    assert _lib.load().gnlp_attn_last_kernel() == (
        _lib.ATTN_KERNEL_FUSED_BF16 if fused else _lib.ATTN_KERNEL_GENERAL_MFMA)
    b, _ = ac.multi_head_dot_attn(qq, kk, vv, dense, ee, _need_scores=False, _need_weights=False, **kw)
    assert torch.equal(a, b)
    exp, (es, ew) = ref.multi_head_dot_attn(
        prep(q, dtype), prep(k, dtype), prep(v, dtype), dense.cpu().numpy(),
        None if e is None else prep(e, dtype), scaled=True, query_head_units=d, layout='NTK')
    assert rel_err(to_np(a), exp) < tol(dtype)
    # with scores / weights wanted nothing may be skipped: the whole (B, H, L, L) is written
    c, (s, w) = ac.multi_head_dot_attn(qq, kk, vv, mask, ee, **kw)
    assert rel_err(to_np(c), exp) < tol(dtype)
    assert rel_err(to_np(s), es) < tol(dtype) and rel_err(to_np(w), ew) < tol(dtype)
    assert (to_np(w)[np.broadcast_to(~dense.cpu().numpy()[:, None], w.shape)] == 0).all()


def test_causal_tagged_mask_is_not_ones_for_the_layers(nlp):
    """The encoder layer and the T5 block only know the 'full' predicate: a causal-tagged mask must
    reach them as bytes (GPT-style use of an encoder layer), not as 'no mask'."""
    from gluonnlp_amd.models.transformer import TransformerEncoderLayer
    torch.manual_seed(0)
    B, L, U, H = 2, 24, 64, 4
    layer = TransformerEncoderLayer(units=U, hidden_size=128, num_heads=H, dtype='float32').cuda()
    for p in layer.parameters():
        p.data.normal_(0, 0.1)
    x = torch.randn(B, L, U, device='cuda')
    mask = nlp.attention_cell.gen_self_attn_mask(x, None, attn_type='causal')
    out_tag, _ = layer(x, mask)
    out_bytes, _ = layer(x, mask.clone())
    out_ones, _ = layer(x, torch.ones_like(mask))
    assert torch.equal(out_tag, out_bytes)
    assert not torch.allclose(out_tag, out_ones)


# ------------------------------------------------------------- ragged batches, packed
@pytest.mark.parametrize('layout', ['NT', 'TN'])
@pytest.mark.parametrize('vl_dtype', [torch.int32, torch.int64])
def test_packed_ragged_stack_is_bit_identical(nlp, layout, vl_dtype):
    """A ragged batch through BertTransformer runs on its valid tokens only (gather -> layers on
    sum(vl) rows with the row count read from device memory -> scatter + zeros): bit for bit the
    padded run (GNLP_PACK_RAGGED=0) on every valid row, exact zeros beyond valid_length; lengths
    0, 1, full and odd ones; a second call with other lengths re-uses the workspace."""
    from gluonnlp_amd import _lib
    from gluonnlp_amd.models import BertTransformer
    torch.manual_seed(5)
    B, L, U, H, F = 128, 256, 256, 4, 1024
    m = BertTransformer(units=U, hidden_size=F, num_layers=3, num_heads=H, dtype='bfloat16',
                        layout=layout).cuda()
    for p in m.parameters():
        p.data.normal_(0, 0.05)
    gen = torch.Generator(device='cuda').manual_seed(7)
    shape = (B, L, U) if layout == 'NT' else (L, B, U)
    x = torch.randn(shape, device='cuda', generator=gen).to(torch.bfloat16)
    for trial in range(2):
        vl = torch.randint(1, L + 1, (B,), device='cuda', generator=gen).to(vl_dtype)
        vl[0], vl[1], vl[2], vl[-1] = L, 1, 0, (L if trial == 0 else 3)
        try:
            _lib.set_knob('GNLP_PACK_RAGGED', '0')
            want, _ = m(x, vl)
            assert _lib.load().gnlp_stack_last_packed() == 0
        finally:
            _lib.set_knob('GNLP_PACK_RAGGED', None)
        got, _ = m(x, vl)
        assert _lib.load().gnlp_stack_last_packed() == 1
        assert torch.equal(got, want)
        o = got if layout == 'NT' else got.transpose(0, 1)
        live = torch.arange(L, device='cuda')[None, :] < vl[:, None]
        assert float(o[~live].float().abs().max()) == 0.0
        assert torch.isfinite(o.float()).all()


def test_packed_ragged_stack_matches_oracle(nlp):
    """The packed run against the float64 oracle on a few whole sequences (the padded run is
    covered by the golden / full-size tests; this pins the packed one independently)."""
    from gluonnlp_amd.models import BertTransformer
    rng = np.random.default_rng(21)
    B, L, U, H, F, NL = 128, 256, 256, 4, 1024, 2
    m = BertTransformer(units=U, hidden_size=F, num_layers=NL, num_heads=H, dtype='bfloat16').cuda()
    layers = [ref.make_encoder_layer_params(rng, U, F) for _ in range(NL)]
    for i, lp in enumerate(layers):
        load_params(m.all_layers[i], lp, torch.bfloat16)
    x = rng.standard_normal((B, L, U))
    vl = rng.integers(1, L + 1, size=B)
    vl[:4] = [L, 1, 77, 200]
    got = to_np(m(to_dev(x, torch.bfloat16), torch.tensor(vl, device='cuda'))[0])
    sel = [0, 1, 2, 3, B - 1]
    exp, _ = ref.bert_transformer(prep(x[sel], torch.bfloat16), vl[sel],
                                  [oracle_params(lp, torch.bfloat16) for lp in layers], H)
    assert rel_err(got[sel], exp) < 3e-2      # two bf16 layers end to end (sqrt(2) x 2e-2)


@pytest.mark.parametrize('B,L,case', [(512, 64, 'mixed'), (128, 256, 'zeros'), (128, 256, 'full'),
                                      (300, 128, 'mixed')])
def test_packed_ragged_stack_edge_batches(nlp, B, L, case):
    """More sequences than the offset kernel has threads (B = 512: several per thread), a batch of
    nothing but empty sequences, a full batch, and B * L not a multiple of 256 (B = 300: the
    padded path must run): packed == padded bit for bit, zeros beyond valid_length."""
    from gluonnlp_amd import _lib
    from gluonnlp_amd.models import BertTransformer
    torch.manual_seed(B + L)
    U, H, F = 256, 4, 1024
    m = BertTransformer(units=U, hidden_size=F, num_layers=2, num_heads=H, dtype='bfloat16').cuda()
    for p in m.parameters():
        p.data.normal_(0, 0.05)
    gen = torch.Generator(device='cuda').manual_seed(B)
    x = torch.randn((B, L, U), device='cuda', generator=gen).to(torch.bfloat16)
    if case == 'zeros':
        vl = torch.zeros((B,), dtype=torch.int64, device='cuda')
    elif case == 'full':
        vl = torch.full((B,), L, dtype=torch.int64, device='cuda')
    else:
        vl = torch.randint(0, L + 1, (B,), device='cuda', generator=gen)
    try:
        _lib.set_knob('GNLP_PACK_RAGGED', '0')
        want, _ = m(x, vl)
    finally:
        _lib.set_knob('GNLP_PACK_RAGGED', None)
    got, _ = m(x, vl)
    assert _lib.load().gnlp_stack_last_packed() == (1 if (B * L) % 256 == 0 else 0)
    assert torch.equal(got, want)
    live = torch.arange(L, device='cuda')[None, :] < vl[:, None]
    if bool((~live).any()):
        assert float(got[~live].float().abs().max()) == 0.0
    assert torch.isfinite(got.float()).all()


@pytest.mark.gpu
@pytest.mark.parametrize('attn_type', ['full', 'causal'])
@pytest.mark.parametrize('L', [200, 512])
def test_fused_attention_dense_mask_blocks(nlp, attn_type, L):
    """The dense-mask variant of the fused kernel on masks made of whole blocks: the bytes of a
    gen_self_attn_mask product handed over untagged.  Blocks that are switched on entirely take
    the path without per-element selects, blocks switched off entirely are skipped, the ragged
    ones go element by element -- bit for bit the output of the in-kernel predicate; the same with
    a bias on top."""
    ac = nlp.attention_cell
    B, H, d = 4, 2, 64
    gen = torch.Generator(device='cuda').manual_seed(L)
    q, k, v = [torch.randn((B, L, H, d), device='cuda', generator=gen).to(torch.bfloat16)
               for _ in range(3)]
    e = torch.randn((H, L, L), device='cuda', generator=gen).to(torch.bfloat16)
    vl = torch.tensor([L, L // 2 + 3, 1, 130], device='cuda')
    tagged = ac.gen_self_attn_mask(torch.zeros((B, L, 1), device='cuda'), vl, attn_type=attn_type)
    dense = tagged.clone()
    assert ac.mask_predicate(dense) is None
    kw = dict(scaled=True, query_head_units=d, layout='NTK', _need_scores=False, _need_weights=False)
    for bias in (None, e):
        a, _ = ac.multi_head_dot_attn(q, k, v, tagged, bias, **kw)
        assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_FUSED_BF16
        b, _ = ac.multi_head_dot_attn(q, k, v, dense, bias, **kw)
        assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_FUSED_BF16
        assert torch.equal(a, b)
    # a mask whose key stride is not 1 (a transposed view) goes to the general kernel
    tview = dense.transpose(1, 2)
    c, _ = ac.multi_head_dot_attn(q, k, v, tview, None, **kw)
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_GENERAL_MFMA
    ref_c, _ = ac.multi_head_dot_attn(q, k, v, tview.contiguous(), None, **kw)
    assert rel_err(to_np(c), to_np(ref_c)) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['mask_upper', 'mask_lower', 'spike_upper', 'spike_lower'])
def test_fused_attention_max_over_both_lane_halves(nlp, case):
    """Regression: in the fused kernel two lanes share a query row (keys with key % 8 < 4 and
    >= 4); the row maximum must cover both.  It once covered the lower half only (the compiler
    dropped the maximum behind the lane exchange): a row whose lower-half keys were all masked, or
    whose largest un-scaled score sat on an upper-half key by more than 2^127, came out NaN."""
    ac = nlp.attention_cell
    B, H, L, d = 2, 2, 128, 64
    rng = np.random.default_rng(7)
    q, k, v = [rng.standard_normal((B, H, L, d)) for _ in range(3)]
    mask_np = np.ones((B, L, L), dtype=bool)
    scaled = True
    if case.startswith('mask'):
        mask_np[:] = False
        lo = 60 if case == 'mask_upper' else 56        # keys 60..63 (upper half) / 56..59 (lower)
        mask_np[:, :, lo:lo + 4] = True
        mask_np[1, :, 68:72 if case == 'mask_upper' else 64:68] = True
    else:
        scaled = False                                 # T5: raw scores, no temperature
        j = 13 if case == 'spike_upper' else 9         # 13 % 8 = 5 (upper half), 9 % 8 = 1 (lower)
        k[:, :, j] = q[:, :, 5] * 8.0                  # score ~ 8 |q|^2 ~ 500 against |s| < 40
    qq, kk, vv = [to_dev(np.transpose(x, (0, 2, 1, 3)), torch.bfloat16) for x in (q, k, v)]
    mask = torch.tensor(mask_np, device='cuda')
    ctx, _ = ac.multi_head_dot_attn(qq, kk, vv, mask, None, scaled=scaled, query_head_units=d,
                                    layout='NTK', _need_scores=False, _need_weights=False)
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_FUSED_BF16
    got = to_np(ctx)
    assert np.isfinite(got).all()
    ectx, _ = ref.multi_head_dot_attn(prep(q, torch.bfloat16), prep(k, torch.bfloat16),
                                      prep(v, torch.bfloat16), mask_np, None, scaled=scaled,
                                      query_head_units=d)
    assert rel_err(got, ectx) < 2e-2
    # the same through the in-kernel predicate (no dense mask): spikes only
    if case.startswith('spike'):
        vl = torch.tensor([L, L], device='cuda')
        tagged = ac.gen_self_attn_mask(torch.zeros((B, L, 1), device='cuda'), vl)
        ctx2, _ = ac.multi_head_dot_attn(qq, kk, vv, tagged, None, scaled=scaled, query_head_units=d,
                                         layout='NTK', _need_scores=False, _need_weights=False)
        assert np.isfinite(to_np(ctx2)).all() and rel_err(to_np(ctx2), ectx) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize('with_data_vl', [False, True])
@pytest.mark.parametrize('Lq,Lm', [(70, 200), (256, 512), (1, 33)])
def test_fused_attention_memory_mask_predicate(nlp, with_data_vl, Lq, Lm):
    """Decoder cross-attention: a mask straight from gen_mem_attn_mask is evaluated in the fused
    kernel from its two length vectors ((j < mem_vl[b]) & (i < data_vl[b]), attention_cell.py:
    186-274) -- bit for bit the output of the same bytes handed over untagged, and the oracle's;
    in-place edits of the mask or of the callers' length tensors do not change what is computed
    from the bytes."""
    ac = nlp.attention_cell
    B, H, d = 3, 2, 64
    rng = np.random.default_rng(Lq + Lm)
    q = rng.standard_normal((B, H, Lq, d))
    k, v = rng.standard_normal((B, H, Lm, d)), rng.standard_normal((B, H, Lm, d))
    mem_vl = np.array([Lm, 1, max(1, Lm * 2 // 3)])
    data_vl = np.array([Lq, max(1, Lq // 2), 1]) if with_data_vl else None
    qq, kk, vv = [to_dev(np.transpose(x, (0, 2, 1, 3)), torch.bfloat16) for x in (q, k, v)]
    mvl_t = torch.tensor(mem_vl, device='cuda')
    dvl_t = torch.tensor(data_vl, device='cuda') if with_data_vl else None
    mem = torch.zeros((B, Lm, 1), device='cuda')
    data = torch.zeros((B, Lq, 1), device='cuda')
    tagged = ac.gen_mem_attn_mask(mem, mvl_t, data, dvl_t)
    mask_np = ref.gen_mem_attn_mask((B, Lm), mem_vl, (B, Lq), data_vl)
    assert np.array_equal(to_np(tagged).astype(bool), mask_np.astype(bool))
    assert ac.mask_predicate(tagged) is None and ac.mask_predicate(tagged, mem=True)[0] == 'mem'
    dense = tagged.clone()
    kw = dict(scaled=True, query_head_units=d, layout='NTK', _need_scores=False, _need_weights=False)
    a, _ = ac.multi_head_dot_attn(qq, kk, vv, tagged, None, **kw)
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_FUSED_BF16
    b, _ = ac.multi_head_dot_attn(qq, kk, vv, dense, None, **kw)
    assert torch.equal(a, b)
    ectx, _ = ref.multi_head_dot_attn(prep(q, torch.bfloat16), prep(k, torch.bfloat16),
                                      prep(v, torch.bfloat16), mask_np, None, query_head_units=d)
    assert rel_err(to_np(a), ectx) < 2e-2
    # the callers' length tensors edited in place afterwards: the tag holds its own copy
    mvl_t.fill_(1)
    c, _ = ac.multi_head_dot_attn(qq, kk, vv, tagged, None, **kw)
    assert torch.equal(a, c)
    # the mask edited in place: the tag is void, the bytes rule
    tagged[:, :, 0] = False
    assert ac.mask_predicate(tagged, mem=True) is None
    e, _ = ac.multi_head_dot_attn(qq, kk, vv, tagged, None, **kw)
    f, _ = ac.multi_head_dot_attn(qq, kk, vv, tagged.clone(), None, **kw)
    assert torch.equal(e, f)
    # the C ABI refuses the pair-of-lengths predicate where no kernel evaluates it (head dims
    # beyond 128), and with a mask or the causal flag beside it
    lib = _lib.load()
    dd = _lib.AttnDesc()
    dd.B, dd.H, dd.Lq, dd.Lm, dd.dk, dd.dv = 1, 1, 4, 4, 256, 256
    dd.q_stride[:] = dd.k_stride[:] = dd.v_stride[:] = dd.o_stride[:] = [1024, 256, 256]
    dd.mem_lengths = 1
    x = torch.zeros(4096, device='cuda')
    two = torch.tensor([4, 4], device='cuda', dtype=torch.int32)
    args = lambda m, ms: (x.data_ptr(), None, None, x.data_ptr(), x.data_ptr(), x.data_ptr(),
                          ctypes.byref(dd), _lib.F32, m, ms, two.data_ptr(), _lib.I32, None, None,
                          1.0, None, 0, None)
    assert lib.gnlp_multi_head_dot_attn(*args(None, None)) == -2
    assert b'memory-mask' in lib.gnlp_last_error()
    dd.dk = dd.dv = 8
    dd.causal = 1
    assert lib.gnlp_multi_head_dot_attn(*args(None, None)) == -1
    dd.causal = 0
    mb = torch.ones(16, dtype=torch.bool, device='cuda')
    assert lib.gnlp_multi_head_dot_attn(*args(mb.data_ptr(), (ctypes.c_int64 * 3)(16, 4, 1))) == -1


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NKT', 'NTK', 'TNK'])
def test_general_attention_memory_mask_predicate(nlp, layout, dtype):
    """The same predicate on the general MFMA kernel (fp32, other head dims, scores / weights
    wanted): the tagged gen_mem_attn_mask product against its untagged bytes and the oracle, with
    the weights of switched-off keys / rows exactly zero."""
    ac = nlp.attention_cell
    B, H, Lq, Lm, dk, dv = 3, 2, 45, 70, 24, 40
    rng = np.random.default_rng(5)
    q, k, v = (rng.standard_normal((B, H, Lq, dk)), rng.standard_normal((B, H, Lm, dk)),
               rng.standard_normal((B, H, Lm, dv)))
    e = rng.standard_normal((H, Lq, Lm))
    perm = {'NKT': None, 'NTK': (0, 2, 1, 3), 'TNK': (2, 0, 1, 3)}[layout]
    qq, kk, vv = [to_dev(x if perm is None else np.transpose(x, perm), dtype) for x in (q, k, v)]
    mem_vl, data_vl = np.array([Lm, 1, 37]), np.array([11, Lq, 1])
    tagged = ac.gen_mem_attn_mask(torch.zeros((B, Lm, 1), device='cuda'), torch.tensor(mem_vl, device='cuda'),
                                  torch.zeros((B, Lq, 1), device='cuda'), torch.tensor(data_vl, device='cuda'))
    mask_np = ref.gen_mem_attn_mask((B, Lm), mem_vl, (B, Lq), data_vl)
    kw = dict(scaled=True, query_head_units=dk, layout=layout)
    a, (sa, wa) = ac.multi_head_dot_attn(qq, kk, vv, tagged, to_dev(e, dtype), **kw)
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_GENERAL_MFMA
    b, (sb, wb) = ac.multi_head_dot_attn(qq, kk, vv, tagged.clone(), to_dev(e, dtype), **kw)
    assert torch.equal(a, b) and torch.equal(wa, wb) and torch.equal(sa, sb)
    off = ~torch.from_numpy(mask_np.astype(bool)).cuda()
    assert (wa[off[:, None].expand_as(wa)] == 0).all()
    ectx, (_, ew) = ref.multi_head_dot_attn(prep(q, dtype), prep(k, dtype), prep(v, dtype), mask_np,
                                            prep(e, dtype), query_head_units=dk)
    got = to_np(a) if layout != 'TNK' else np.transpose(to_np(a), (1, 0, 2))
    assert rel_err(got, ectx) < tol(dtype)
    assert rel_err(to_np(wa), ew) < tol(dtype)
