"""Pin the CPU oracle (oracle/ref.py, oracle/bucket_ref.c) against the golden vectors generated
from the reference's PyTorch sibling (oracle/gen_golden.py) and against the truth tables /
properties the reference's own tests hold.  CPU only."""
import ctypes
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from oracle import ref

G = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(G, name))


# ------------------------------------------------------------------ masks
def test_mask_goldens_bit_exact():
    g = load('mask_self.npz')
    for tag, B, L in [('t', 4, 8), ('doc', 2, 8)]:
        vl = g[tag + '_vl']
        for layout in ['NT', 'TN']:
            shape = (B, L, 5) if layout == 'NT' else (L, B, 5)
            for attn_type in ['full', 'causal']:
                m = ref.gen_self_attn_mask(np.zeros(shape), vl, attn_type=attn_type, layout=layout)
                assert m.dtype == np.bool_
                assert_array_equal(m, g['{}_{}_{}'.format(tag, attn_type, layout)])
            m = ref.gen_self_attn_mask(np.zeros(shape), None, attn_type='causal', layout=layout)
            assert_array_equal(m, g['{}_causal_novl_{}'.format(tag, layout)])


def test_mask_truth_tables():
    """Exact expectations of reference tests/test_attention_cell.py:225-269 and the docstring
    tables attention_cell.py:63-121."""
    rng = np.random.default_rng(0)
    B, Lq, Lm = 4, 8, 6
    vl = rng.integers(1, Lq + 1, size=B)
    mvl = rng.integers(1, Lm + 1, size=B)
    full = ref.gen_self_attn_mask((B, Lq, 5), vl, attn_type='full')
    causal = ref.gen_self_attn_mask((B, Lq, 5), vl, attn_type='causal')
    mem = ref.gen_mem_attn_mask((B, Lm, 5), mvl, (B, Lq, 5), vl)
    for b in range(B):
        for i in range(Lq):
            if i < vl[b]:
                assert full[b, i, :vl[b]].all() and not full[b, i, vl[b]:].any()
                assert causal[b, i, :i + 1].all() and not causal[b, i, i + 1:].any()
                assert mem[b, i, :mvl[b]].all() and not mem[b, i, mvl[b]:].any()
            else:
                assert not full[b, i].any() and not causal[b, i].any() and not mem[b, i].any()
    # docstring example, vl = [8, 6], second sample causal row 'you' = 1 1 1 1 1 1 0 0
    c = ref.gen_self_attn_mask((2, 8, 1), np.array([8, 6]), attn_type='causal')
    assert_array_equal(c[1, 5], [1, 1, 1, 1, 1, 1, 0, 0])
    assert_array_equal(c[1, 6], [0] * 8)
    # no data_valid_length: every query row sees the memory mask
    mem2 = ref.gen_mem_attn_mask((B, Lm, 5), mvl, (B, Lq, 5), None)
    for b in range(B):
        assert mem2[b, :, :mvl[b]].all() and not mem2[b, :, mvl[b]:].any()
    # TN == NT
    assert_array_equal(ref.gen_self_attn_mask((Lq, B, 5), vl, layout='TN'), full)


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize('H', [1, 2, 3])
def test_attention_goldens(H):
    g = load('attn_h{}.npz'.format(H))
    q, k, v = g['q'].astype(np.float64), g['k'].astype(np.float64), g['v'].astype(np.float64)
    mask = g['mask']
    for scaled in [True, False]:
        for ename in ['none', 'share', 'head']:
            e = None if ename == 'none' else g['e_' + ename].astype(np.float64)
            key = 's{}_{}'.format(int(scaled), ename)
            for layout, perm, back in [('NKT', None, None), ('NTK', (0, 2, 1, 3), None),
                                       ('TNK', (2, 0, 1, 3), (1, 0, 2))]:
                qq, kk, vv = [x if perm is None else np.transpose(x, perm) for x in (q, k, v)]
                ctx, (scores, w) = ref.multi_head_dot_attn(qq, kk, vv, mask, e, scaled=scaled,
                                                           query_head_units=q.shape[-1],
                                                           layout=layout)
                if back is not None:
                    ctx = np.transpose(ctx, back)
                assert_allclose(ctx, g[key + '_ctx'], rtol=1e-10, atol=1e-12)
                assert_allclose(scores, g[key + '_scores'], rtol=1e-10, atol=1e-12)
                assert_allclose(w, g[key + '_w'], rtol=1e-10, atol=1e-12)
                # reference test property: masked weights are exactly zero
                for h in range(H):
                    assert (w[:, h][mask == 0] == 0).all()
                # torch-sibling property: row sums in {0, 1}
                rs = w.sum(-1)
                full_masked = (mask.sum(-1) == 0)[:, None, :].repeat(H, 1)
                assert_allclose(rs[~full_masked], 1.0, atol=1e-12)
                assert (rs[full_masked] == 0).all()
    ctx, (_, w) = ref.multi_head_dot_attn(q, k, v, None, None, scaled=True,
                                          query_head_units=q.shape[-1])
    assert_allclose(ctx, g['nomask_ctx'], rtol=1e-10, atol=1e-12)
    assert_allclose(w, g['nomask_w'], rtol=1e-10, atol=1e-12)


def test_attention_errors_and_normalized():
    rng = np.random.default_rng(3)
    q = rng.standard_normal((2, 2, 4, 8))
    with pytest.raises(NotImplementedError):
        ref.multi_head_dot_attn(q, q, q, scaled=True, query_head_units=None)
    with pytest.raises(NotImplementedError):
        ref.multi_head_dot_attn(q, q, q, scaled=False, layout='KNT')
    # normalized: x / (||x|| + eps)  (op.py:306) -- score of a vector with itself is < 1
    _, (scores, _) = ref.multi_head_dot_attn(q, q, q, scaled=False, normalized=True, eps=1e-6)
    n = np.linalg.norm(q, axis=-1)
    expect = (n / (n + 1e-6)) ** 2
    assert_allclose(np.diagonal(scores, axis1=-2, axis2=-1), expect, rtol=1e-12)


# ------------------------------------------------------------------ FFN / layer / stacks
def _params(g, prefix):
    n = len(prefix)
    return {k[n:]: g[k].astype(np.float64) for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize('name,kw', [
    ('postnorm_relu', dict(activation_name='relu', layer_norm_eps=1e-5)),
    ('postnorm_gelu', dict(activation_name='gelu', layer_norm_eps=1e-12)),
    ('prenorm_gelu', dict(activation_name='gelu', pre_norm=True, layer_norm_eps=1e-5)),
    ('gated_gelu_tanh', dict(activation_name='gelu(tanh)', use_gated_activation=True,
                             layer_norm_eps=1e-5)),
])
def test_ffn_goldens(name, kw):
    g = load('ffn.npz')
    y = ref.positionwise_ffn(g[name + '_x'].astype(np.float64), _params(g, name + '/'), **kw)
    assert_allclose(y, g[name + '_y'], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize('name,H,eps,act', [
    ('enc_tiny_eps1e-12', 2, 1e-12, 'gelu'), ('enc_tiny_eps1e-5', 2, 1e-5, 'gelu'),
    ('enc_tiny_relu_vl1', 4, 1e-5, 'relu'), ('enc_d64', 2, 1e-12, 'gelu')])
def test_encoder_layer_and_stack_goldens(name, H, eps, act):
    g = load(name + '.npz')
    x = g['x'].astype(np.float64)
    vl = g['valid_length']
    layers = [_params(g, 'L0/'), _params(g, 'L1/')]
    mask = ref.gen_self_attn_mask(x, vl)
    h, w = ref.transformer_encoder_layer(x, mask, layers[0], H, eps, False, act, 'NT')
    assert_allclose(h, g['out_L0'], rtol=1e-8, atol=1e-10)
    assert_allclose(w, g['attn_w_L0'], rtol=1e-8, atol=1e-12)
    out, _ = ref.bert_transformer(x, vl, layers, H, eps, act, 'NT')
    assert_allclose(out, g['out_seqmask'], rtol=1e-8, atol=1e-10)
    out_r, _ = ref.roberta_encoder(x, vl, layers, H, eps, act, 'NT')
    assert_allclose(out_r, g['out_L1'], rtol=1e-8, atol=1e-10)
    # NT vs TN (reference tests/test_models_bert.py:51-54, 1e-4; exact here up to rounding)
    out_tn, _ = ref.bert_transformer(np.swapaxes(x, 0, 1), vl, layers, H, eps, act, 'TN')
    assert_allclose(np.swapaxes(out_tn, 0, 1), out, rtol=1e-10, atol=1e-12)
    # outputs beyond valid_length are exactly 0 (tests/test_models_albert.py:65-84)
    for b, n in enumerate(vl):
        assert (out[b, n:] == 0).all()
    # output_all_encodings masks every layer and feeds the masked tensor on (bert.py:300-304)
    allenc, _ = ref.bert_transformer(x, vl, layers, H, eps, act, 'NT', output_all_encodings=True)
    assert len(allenc) == 2
    assert_allclose(allenc[-1], out, rtol=1e-8, atol=1e-10)


def test_padding_invariance():
    """Appending padding does not change valid outputs (tests/test_models_albert.py:65-84)."""
    rng = np.random.default_rng(5)
    U, F, H, B, L = 32, 64, 4, 3, 10
    layers = [ref.make_encoder_layer_params(rng, U, F, np.float64) for _ in range(2)]
    x = rng.standard_normal((B, L, U))
    vl = np.array([10, 4, 7])
    out, _ = ref.bert_transformer(x, vl, layers, H)
    xp = np.concatenate([x, rng.standard_normal((B, 6, U))], axis=1)
    outp, _ = ref.bert_transformer(xp, vl, layers, H)
    assert_allclose(outp[:, :L], out, rtol=1e-9, atol=1e-11)
    assert (outp[:, L:] == 0).all()


def test_pre_norm_residual_follows_mx_text():
    """MX adds the *normalised* input (models/transformer.py:259-268); torch sibling the raw."""
    rng = np.random.default_rng(6)
    U, F, H = 16, 32, 2
    p = ref.make_encoder_layer_params(rng, U, F, np.float64)
    x = rng.standard_normal((2, 5, U))
    mask = ref.gen_self_attn_mask(x, np.array([5, 3]))
    a, _ = ref.transformer_encoder_layer(x, mask, p, H, 1e-5, True, 'relu', 'NT', True)
    b, _ = ref.transformer_encoder_layer(x, mask, p, H, 1e-5, True, 'relu', 'NT', False)
    assert not np.allclose(a, b)


# ------------------------------------------------------------------ T5 pieces
def _c_bucket(rel, bidirectional, num_buckets, max_distance):
    so = os.path.join(os.path.dirname(G), '..', 'oracle', '_build', 'libgnlp_oracle.so')
    so = os.path.abspath(so)
    if not os.path.exists(so):
        pytest.skip('oracle C library not built (run __graft_entry__.build())')
    lib = ctypes.CDLL(so)
    rel = np.ascontiguousarray(rel, dtype=np.int32)
    out = np.empty_like(rel)
    lib.oracle_relative_position_bucket(
        rel.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int64(rel.size), int(bidirectional), int(num_buckets), int(max_distance))
    return out


@pytest.mark.parametrize('bidirectional', [True, False])
@pytest.mark.parametrize('num_buckets,max_distance', [(32, 128), (64, 256), (32, 256)])
def test_bucket_numpy_vs_c_vs_hf(bidirectional, num_buckets, max_distance):
    rel = np.concatenate([np.arange(-10000, 10001), np.arange(-10, 11)]).astype(np.int32)
    a = ref.relative_position_bucket(rel, bidirectional, num_buckets, max_distance)
    b = _c_bucket(rel, bidirectional, num_buckets, max_distance)
    assert a.dtype == np.int32
    assert_array_equal(a, b)
    # independent pin: HuggingFace T5's bucket function implements the same mesh-tf algorithm
    try:
        import torch
        from transformers.models.t5.modeling_t5 import T5Attention
    except Exception:
        return
    c = T5Attention._relative_position_bucket(torch.from_numpy(rel.astype(np.int64)),
                                              bidirectional=bidirectional,
                                              num_buckets=num_buckets,
                                              max_distance=max_distance).numpy()
    assert_array_equal(a, c.astype(np.int32))


def test_bucket_properties():
    """reference tests/test_layers.py:200-224."""
    for nb, md in [(32, 128), (64, 256)]:
        rel = np.arange(-1000, 1001).astype(np.int32)
        b = ref.relative_position_bucket(rel, True, nb, md)
        assert (b[rel > 0] >= nb // 2).all()           # :220-221
        assert (b[rel <= 0] < nb // 2).all()
        assert (b[rel >= md] == nb - 1).all()          # saturation
        assert (b[rel <= -md] == nb // 2 - 1).all()
        assert b.min() == 0 and b.max() == nb - 1
    g = ref.gen_rel_position((2, 7, 3))
    assert g.dtype == np.int32 and g.shape == (7, 7)
    assert g[2, 5] == 3 and g[5, 2] == -3              # mem_j - query_i
    gp = ref.gen_rel_position((2, 1, 3), past_data=(2, 6, 4, 8))       # :644-652
    assert_array_equal(gp, g)
    assert_array_equal(ref.gen_rel_position((1, 2, 3), past_data=(6, 2, 4, 8), layout='TN'), g)
    w = np.random.default_rng(0).standard_normal((32, 4))
    s = ref.rel_attention_score_t5(w, g)
    assert s.shape == (4, 7, 7)
    assert_array_equal(s[:, 2, 5], w[ref.relative_position_bucket(np.int32(3))])


def test_t5_block_prefix_consistency():
    """tests/test_models_t5.py: shapes + NT/TN agreement for the T5 encoder restatement."""
    rng = np.random.default_rng(9)
    d_model, d_kv, d_ff, H, B, L = 32, 8, 64, 4, 3, 12
    for act in ['relu', 'gated-gelu']:
        layers = [ref.make_t5_block_params(rng, d_model, d_kv, d_ff, H, act != 'relu', np.float64)
                  for _ in range(2)]
        relw = rng.standard_normal((32, H))
        fin = {'gamma': np.ones(d_model), 'beta': np.zeros(d_model)}
        x = rng.standard_normal((B, L, d_model))
        vl = np.array([12, 5, 1])
        y = ref.t5_encoder(x, vl, relw, layers, fin, H, activation_name=act)
        y_tn = ref.t5_encoder(np.swapaxes(x, 0, 1), vl, relw, layers, fin, H, activation_name=act,
                              layout='TN')
        assert_allclose(np.swapaxes(y_tn, 0, 1), y, rtol=1e-10, atol=1e-12)
        # padding invariance on valid tokens
        xp = np.concatenate([x, rng.standard_normal((B, 4, d_model))], axis=1)
        yp = ref.t5_encoder(xp, vl, relw, layers, fin, H, activation_name=act)
        for b, n in enumerate(vl):
            assert_allclose(yp[b, :n], y[b, :n], rtol=1e-9, atol=1e-11)


def test_softmax_and_norm_semantics():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 3, 4, 5))
    m = rng.integers(0, 2, size=(2, 1, 4, 5)).astype(bool)
    m[0, 0, 1] = False
    w = ref.masked_softmax(x, m, temperature=2.0)
    neg = np.where(m, x, -1e18) / 2.0     # utils/tvm_utils.py:59-75 restatement
    e = np.exp(neg - neg.max(-1, keepdims=True))
    assert_allclose(w, e / e.sum(-1, keepdims=True) * m, rtol=1e-12, atol=1e-15)
    assert (w[0, :, 1] == 0).all()
    ls = ref.masked_logsoftmax(x, m)
    assert np.isneginf(ls[~np.broadcast_to(m, x.shape)]).all()
    g = rng.standard_normal(5)
    b = rng.standard_normal(5)
    y = ref.rms_norm(x, g, b, 1e-6)
    assert_allclose(y, x / np.sqrt((x ** 2).mean(-1, keepdims=True) + 1e-6) * g + b, rtol=1e-12)
    with pytest.raises(NotImplementedError):
        ref.activation(x, 'swish')
    with pytest.raises(ValueError):
        ref.gelu(x, 'foo')
    with pytest.raises(NotImplementedError):
        ref.norm(x, {}, 'group_norm')


@pytest.mark.parametrize('layout', ['NT', 'TN'])
def test_embedding_front_end_against_torch_ops(layout):
    """SURVEY 8f rank 1.  The torch sibling has no importable BertModel (its models/bert.py pulls
    in mxnet), so the restatement of get_initial_embedding / apply_pooling
    (models/bert.py:458-529) is pinned against the same op sequence written with torch's own CPU
    ops (F.embedding, F.layer_norm, F.linear + tanh) in float64, plus the properties the reference
    states: token_types None == zeros, position ids = arange over the time axis broadcast over the
    batch axis, np.take(mode='clip') beyond max_length."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(7)
    V, NT, ML, U, B, L = 50, 3, 12, 16, 3, 10
    word, typ, pos = (rng.standard_normal((n, U)) for n in (V, NT, ML))
    g, b = rng.standard_normal(U), rng.standard_normal(U)
    shape = (B, L) if layout == 'NT' else (L, B)
    ids = rng.integers(0, V, shape)
    tts = rng.integers(0, NT, shape)
    got = ref.initial_embedding(ids, tts, word, typ, pos, {'gamma': g, 'beta': b}, 1e-12, layout)
    t = lambda a: torch.from_numpy(np.asarray(a))
    e = F.embedding(t(ids), t(word)) + F.embedding(t(tts), t(typ))
    p = F.embedding(torch.arange(L), t(pos))
    e = e + (p[None] if layout == 'NT' else p[:, None])
    want = F.layer_norm(e, (U,), t(g), t(b), 1e-12).numpy()
    assert np.abs(got - want).max() < 1e-12
    # token_types None == zeros (bert.py:490-491); no type table / no position table / no LN
    z = ref.initial_embedding(ids, None, word, typ, pos, {'gamma': g, 'beta': b}, 1e-12, layout)
    z2 = ref.initial_embedding(ids, np.zeros_like(ids), word, typ, pos, {'gamma': g, 'beta': b},
                               1e-12, layout)
    assert np.array_equal(z, z2)
    assert np.array_equal(ref.initial_embedding(ids, None, word), word[ids])
    # positions beyond the table are clipped (layers.py:481-482, mode='clip')
    long_ids = rng.integers(0, V, (2, ML + 5) if layout == 'NT' else (ML + 5, 2))
    lo = ref.initial_embedding(long_ids, None, word, None, pos, None, 0.0, layout)
    last = lo[:, -1] if layout == 'NT' else lo[-1]
    lid = long_ids[:, -1] if layout == 'NT' else long_ids[-1]
    assert np.allclose(last, word[lid] + pos[ML - 1])
    # pooler: CLS row -> Dense + tanh
    seq = rng.standard_normal(shape + (U,))
    w, bb = rng.standard_normal((U, U)) * 0.2, rng.standard_normal(U)
    cls = seq[:, 0] if layout == 'NT' else seq[0]
    assert np.abs(ref.apply_pooling(seq, w, bb, 'tanh', layout) -
                  torch.tanh(F.linear(t(cls), t(w), t(bb))).numpy()).max() < 1e-12
    assert np.array_equal(ref.apply_pooling(seq, None, None, 'tanh', layout), cls)


@pytest.mark.parametrize('name,H,act', [('dec_tiny', 2, 'relu'), ('dec_d64', 2, 'gelu')])
def test_decoder_layer_goldens(name, H, act):
    """the restated TransformerDecoderLayer (models/transformer.py:513-598)
    against the torch sibling's decoder layer (post-norm, where the two texts agree), two layers,
    ragged target and memory lengths incl. 1; and the decoder's masks against the stored ones."""
    g = np.load(os.path.join(G, name + '.npz'))
    x, mem, vl, mvl = g['x'], g['mem'], g['valid_length'], g['mem_valid_length']
    causal = ref.gen_self_attn_mask(x, vl, attn_type='causal')
    mm = ref.gen_mem_attn_mask(mem, mvl, x, vl)
    assert np.array_equal(causal, g['self_causal_mask'])
    assert np.array_equal(mm, g['mem_attn_mask'])
    h = x.astype(np.float64)
    layers = []
    for li in range(2):
        p = {k[len('L{}/'.format(li)):]: g[k].astype(np.float64) for k in g.files
             if k.startswith('L{}/'.format(li))}
        layers.append(p)
        h = ref.transformer_decoder_layer(h, mem.astype(np.float64), causal, mm, p, H, 1e-5, False,
                                          act)
        assert np.abs(h - g['out_L{}'.format(li)]).max() < 1e-9
    full = ref.transformer_decoder(x.astype(np.float64), vl, mem.astype(np.float64), mvl, layers,
                                   H, 1e-5, False, act)
    assert np.abs(full - g['out_L1']).max() < 1e-9
    # TN layout gives the transposed result
    tn = ref.transformer_decoder(np.swapaxes(x, 0, 1).astype(np.float64), vl,
                                 np.swapaxes(mem, 0, 1).astype(np.float64), mvl, layers, H, 1e-5,
                                 False, act, layout='TN')
    assert np.abs(np.swapaxes(tn, 0, 1) - full).max() < 1e-9
    # pre_norm: the MX text adds the NORMALISED input (transformer.py:569-570,582-583); the
    # sibling adds the raw one -- the two must differ, and the restatement follows the MX text
    a = ref.transformer_decoder_layer(h, mem.astype(np.float64), causal, mm, layers[0], H, 1e-5,
                                      True, act, residual_normed=True)
    b = ref.transformer_decoder_layer(h, mem.astype(np.float64), causal, mm, layers[0], H, 1e-5,
                                      True, act, residual_normed=False)
    assert np.abs(a - b).max() > 1e-3


@pytest.mark.parametrize('name', ['t5_hf_relu', 't5_hf_gated'])
def test_t5_encoder_decoder_against_hf_goldens(name):
    """The whole T5 path of the oracle (bidirectional / unidirectional buckets, un-scaled
    attention + bias, causal + memory masks, RMSNorm pre-norm blocks, relu and gated-gelu FFN)
    against HuggingFace's T5Model in float64, with the weights renamed the way the reference's
    converter renames them -- the reference's own equivalence check
    (scripts/conversion_toolkits/convert_t5.py:203-232, 1e-3 there, 5e-6 here: HF's T5LayerNorm
    takes its mean of squares in float32 whatever the model dtype), on the valid target positions as
    that check does (HF leaves padded query rows unmasked)."""
    g = load(name + '.npz')
    p = {k[2:]: g[k] for k in g.files if k.startswith('p:')}
    svl, tvl = g['src_valid_length'], g['tgt_valid_length']
    enc, dec = ref.t5_model(g['src'], svl, g['tgt'], tvl, p, int(g['num_layers']),
                            int(g['num_heads']),
                            activation_name='gated-gelu' if int(g['gated']) else 'relu')
    for b in range(len(svl)):
        assert_allclose(enc[b, :svl[b]], g['enc_out'][b, :svl[b]], rtol=0, atol=5e-6)
        assert_allclose(dec[b, :tvl[b]], g['dec_out'][b, :tvl[b]], rtol=0, atol=5e-6)


def bert_from_ids(p, ids, token_types, valid_length, num_layers, num_heads):
    """ids -> (sequence, pooled) with the oracle's pieces, parameters under the reference's names
    (models/bert.py:404-456)."""
    emb = ref.initial_embedding(ids, token_types, p['word_embed.weight'],
                                p['token_type_embed.weight'], p['token_pos_embed._embed.weight'],
                                {'gamma': p['embed_layer_norm.gamma'],
                                 'beta': p['embed_layer_norm.beta']}, eps=1e-12)
    layers = [ref._sub(p, 'encoder.all_layers.{}.'.format(i)) for i in range(num_layers)]
    seq, _ = ref.bert_transformer(emb, valid_length, layers, num_heads, 1e-12, 'gelu')
    return seq, ref.apply_pooling(seq, p['pooler.weight'], p['pooler.bias'])


@pytest.mark.parametrize('name', ['bert_hf_d64', 'bert_hf_h4'])
def test_bert_backbone_against_hf_goldens(name):
    """ids -> contextual embeddings -> pooled output of the oracle against HuggingFace's BertModel
    in float64 (oracle/gen_golden_bert_hf.py: the reference converter's renaming of the TF layout
    composed with HF's).  Pins the embedding front-end, the packed-QKV convention, erf-GELU, both
    LayerNorms and the pooler in one known-answer check.  Valid positions only: HF masks keys, the
    reference also masks padded query rows and zeroes them (models/bert.py:311-312)."""
    g = load(name + '.npz')
    p = {k[2:]: g[k] for k in g.files if k.startswith('p:')}
    vl = g['valid_length']
    seq, pooled = bert_from_ids(p, g['ids'], g['token_types'], vl, int(g['num_layers']),
                                int(g['num_heads']))
    for b in range(len(vl)):
        assert_allclose(seq[b, :vl[b]], g['sequence'][b, :vl[b]], rtol=0, atol=1e-10)
        assert (seq[b, vl[b]:] == 0).all()
    assert_allclose(pooled, g['pooled'], rtol=0, atol=1e-10)


@pytest.mark.parametrize('method,bidirectional,max_distance', [
    ('transformer_xl', False, None), ('transformer_xl', True, 6), ('shaw', False, 20),
    ('shaw', True, 5)])
def test_rel_attention_score_restatement(method, bidirectional, max_distance):
    """'transformer_xl' / 'shaw' scores of the oracle (the reference's unique + matmul + gather
    text) against an independent per-element formulation R[b,h,i,j] = <Q[b,h,i], E[rel(i,j)][h]>
    (the docstring formula, attention_cell.py:678-689), plus the reference test's properties:
    permutation equivariance and 3-way layout agreement (tests/test_attention_cell.py:332-389).
    No known-answer vectors exist for these methods in the reference: parity unpinned beyond that."""
    rng = np.random.default_rng(5)
    B, H, Lq, Lm, C = 2, 3, 7, 9, 4
    q = rng.standard_normal((B, H, Lq, C))
    rel = np.arange(Lq)[:, None] - np.arange(Lm)[None, :]      # as the reference test builds it
    kw = dict(method=method, num_heads=H, max_distance=max_distance, bidirectional=bidirectional)
    if method == 'transformer_xl':
        kw['rel_proj_weight'] = rng.standard_normal((H * C, H * C))
    else:
        vocab = (2 * max_distance + 1) if bidirectional else (max_distance + 1)
        kw['shaw_weight'] = rng.standard_normal((vocab, H * C))
    got = ref.rel_attention_score(q, rel, layout='NKT', **kw)
    assert got.shape == (B, H, Lq, Lm)
    want = np.zeros_like(got)
    for i in range(Lq):
        for j in range(Lm):
            r = rel[i, j]
            if max_distance is not None:
                r = int(np.clip(r, -max_distance if bidirectional else 0, max_distance))
            if method == 'transformer_xl':
                e = ref.sinusoidal_positional_embedding(np.array([r]), H * C)[0] @ kw['rel_proj_weight'].T
            else:
                e = kw['shaw_weight'][r % kw['shaw_weight'].shape[0]]
            want[:, :, i, j] = np.einsum('bhc,hc->bh', q[:, :, i, :], e.reshape(H, C))
    assert_allclose(got, want, rtol=0, atol=1e-12)
    qp, mp = rng.permutation(Lq), rng.permutation(Lm)
    perm = ref.rel_attention_score(q[:, :, qp, :], rel[qp][:, mp], layout='NKT', **kw)
    assert_allclose(perm, got[:, :, qp][:, :, :, mp], rtol=0, atol=1e-12)
    assert_allclose(ref.rel_attention_score(q.transpose(0, 2, 1, 3), rel, layout='NTK', **kw), got,
                    rtol=0, atol=1e-12)
    assert_allclose(ref.rel_attention_score(q.transpose(2, 0, 1, 3), rel, layout='TNK', **kw), got,
                    rtol=0, atol=1e-12)


def test_sinusoidal_embedding_restatement():
    """layers.py:405-450 against torch's float64 sin / cos of the same float32 arguments; odd
    units get the zero column."""
    import torch
    pos = np.array([0, 1, 2, 17, 511, -3])
    for units in (8, 9):
        got = ref.sinusoidal_positional_embedding(pos, units)
        half = units // 2
        base = np.exp(np.arange(half, dtype=np.float32) * -(np.log(10000) / (half - 1))).astype(np.float32)
        arg = torch.from_numpy((pos.astype(np.float32)[:, None] * base).astype(np.float32)).double()
        want = torch.cat([torch.sin(arg), torch.cos(arg)] +
                         ([torch.zeros(len(pos), 1, dtype=torch.float64)] if units % 2 else []), -1)
        assert got.shape == (len(pos), units)
        assert_allclose(got, want.numpy(), rtol=0, atol=1e-15)


# ---------------------------------------------------------------------------------------------
# backward pass: oracle gradients pinned to autograd through the reference's
# PyTorch sibling (oracle/gen_golden_grad.py -> tests/golden/grad.npz)
# ---------------------------------------------------------------------------------------------
def _grad_cases(g):
    for tag, biases in (('small', ('none', 'share', 'head')), ('d64', ('none', 'head'))):
        for scaled in (1, 0):
            for ename in biases:
                yield tag, scaled, ename


@pytest.mark.parametrize('layout', ['NKT', 'NTK', 'TNK'])
def test_attention_backward_goldens(layout):
    g = load('grad.npz')

    def lay(x):    # goldens are stored NKT
        return {'NKT': x, 'NTK': x.transpose(0, 2, 1, 3), 'TNK': x.transpose(2, 0, 1, 3)}[layout]

    for tag, scaled, ename in _grad_cases(g):
        q, k, v = [g['{}_{}'.format(tag, n)].astype(np.float64) for n in 'qkv']
        e = None if ename == 'none' else g['{}_e_{}'.format(tag, ename)].astype(np.float64)
        gc = g[tag + '_g_ctx'].astype(np.float64)
        if layout == 'TNK':
            gc = gc.transpose(1, 0, 2)
        key = '{}_s{}_{}'.format(tag, scaled, ename)
        ctx, _ = ref.multi_head_dot_attn(lay(q), lay(k), lay(v), g[tag + '_mask'], e,
                                         scaled=bool(scaled), query_head_units=q.shape[-1],
                                         layout=layout)
        want_ctx = g[key + '_ctx']
        assert_allclose(ctx if layout != 'TNK' else ctx.transpose(1, 0, 2), want_ctx,
                        rtol=1e-12, atol=1e-12)
        dq, dk, dv, de = ref.multi_head_dot_attn_backward(
            gc, lay(q), lay(k), lay(v), g[tag + '_mask'], e, scaled=bool(scaled),
            query_head_units=q.shape[-1], layout=layout)
        for got, name in ((dq, 'dq'), (dk, 'dk'), (dv, 'dv')):
            assert got.shape == lay(g[key + '_' + name]).shape
            assert_allclose(got, lay(g[key + '_' + name]), rtol=1e-11, atol=1e-12, err_msg=key)
        if e is None:
            assert de is None
        else:
            assert de.shape == e.shape
            assert_allclose(de, g[key + '_de'], rtol=1e-11, atol=1e-12, err_msg=key)
    # the fully masked query row of the small case gets no gradient and gives none
    assert not g['small_mask'][0, 3].any()
    assert np.all(g['small_s1_head_dq'][0, :, 3] == 0)


@pytest.mark.parametrize('tag', ['ln48', 'ln1024'])
def test_layer_norm_backward_goldens(tag):
    g = load('grad.npz')
    x, gam, bet, go = [g['{}_{}'.format(tag, n)].astype(np.float64)
                       for n in ('x', 'gamma', 'beta', 'g')]
    eps = float(g[tag + '_eps'])
    assert_allclose(ref.layer_norm(x, gam, bet, eps), g[tag + '_y'], rtol=1e-12, atol=1e-12)
    dx, dgam, dbet = ref.layer_norm_backward(go, x, gam, eps)
    assert_allclose(dx, g[tag + '_dx'], rtol=1e-10, atol=1e-12)
    assert_allclose(dgam, g[tag + '_dgamma'], rtol=1e-10, atol=1e-12)
    assert_allclose(dbet, g[tag + '_dbeta'], rtol=1e-10, atol=1e-12)


def test_rms_norm_backward_by_finite_differences():
    """No rms_norm gradient golden (the sibling has no RMSNorm layer): central differences of
    the oracle's own forward, float64 -- parity unpinned beyond that."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 10))
    gam = 1.0 + 0.1 * rng.standard_normal(10)
    go = rng.standard_normal((3, 10))
    dx, dgam, dbet = ref.rms_norm_backward(go, x, gam, 1e-6)
    h = 1e-6
    num = np.zeros_like(x)
    for i in range(x.shape[0]):
        for j in range(x.shape[1]):
            xp, xm = x.copy(), x.copy()
            xp[i, j] += h
            xm[i, j] -= h
            num[i, j] = ((ref.rms_norm(xp, gam, 0.0) - ref.rms_norm(xm, gam, 0.0)) * go).sum() / (2 * h)
    assert_allclose(dx, num, rtol=1e-6, atol=1e-8)
    assert_allclose(dgam, (go * x / np.sqrt((x * x).mean(-1, keepdims=True) + 1e-6)).sum(0))
    assert_allclose(dbet, go.sum(0))


def test_l2_normalize_backward_by_finite_differences():
    """No golden for this gradient: central differences of the oracle's own forward."""
    rng = np.random.default_rng(6)
    x = rng.standard_normal((4, 7))
    go = rng.standard_normal((4, 7))
    dx = ref.l2_normalize_backward(go, x, -1, 1e-6)
    h = 1e-6
    num = np.zeros_like(x)
    for i in range(x.shape[0]):
        for j in range(x.shape[1]):
            xp, xm = x.copy(), x.copy()
            xp[i, j] += h
            xm[i, j] -= h
            num[i, j] = ((ref.l2_normalize(xp, -1, 1e-6) - ref.l2_normalize(xm, -1, 1e-6)) * go).sum() / (2 * h)
    assert_allclose(dx, num, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize('method,bidirectional,max_distance', [
    ('shaw', True, 3), ('shaw', False, 4), ('transformer_xl', False, None),
    ('transformer_xl', True, 3)])
def test_rel_attention_score_backward_by_finite_differences(method, bidirectional, max_distance):
    """The reference holds no known answers for these gradients (its test compares gradient norms
    across layouts, tests/test_attention_cell.py:313-391): central differences of the oracle's own
    forward -- parity unpinned beyond that."""
    rng = np.random.default_rng(0)
    B, H, Lq, Lm, C = 2, 2, 4, 6, 3
    q = rng.standard_normal((B, Lq, H, C))
    rel = (np.arange(Lm)[None, :] - np.arange(Lq)[:, None]).astype(np.int32)
    if method == 'shaw' and not bidirectional:
        rel = np.abs(rel)
    g = rng.standard_normal((B, H, Lq, Lm))
    kw = dict(method=method, num_heads=H, max_distance=max_distance, bidirectional=bidirectional,
              layout='NTK')
    if method == 'shaw':
        pname = 'shaw_weight'
        w = rng.standard_normal(((2 * max_distance + 1) if bidirectional else max_distance + 1,
                                 H * C))
    else:
        pname = 'rel_proj_weight'
        w = rng.standard_normal((H * C, 6))
    kw[pname] = w
    dq, dp = ref.rel_attention_score_backward(g, q, rel, **kw)

    def f(q_, w_):
        return (ref.rel_attention_score(q_, rel, **{**kw, pname: w_}) * g).sum()
    h = 1e-6
    for arr, grad, which in ((q, dq, 0), (w, dp, 1)):
        num = np.zeros_like(arr)
        for idx in np.ndindex(arr.shape):
            ap, am = arr.copy(), arr.copy()
            ap[idx] += h
            am[idx] -= h
            num[idx] = ((f(ap, w) - f(am, w)) if which == 0 else (f(q, ap) - f(q, am))) / (2 * h)
        assert_allclose(grad, num, rtol=1e-6, atol=1e-7)
