"""GPU parity tests (through the C ABI) of the embedding front-end and the pooler -- SURVEY 8f
rank 1: BertModel / RobertaModel / ElectraModel.get_initial_embedding and apply_pooling
(models/bert.py:458-529, roberta.py:337-398, electra.py:403-510) -- against the CPU oracle.
fp32 within 1e-5, bf16 within 2e-2; gathers with clipped indices."""
import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import load_params, prep, rel_err, to_dev, to_np, tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    from gluonnlp_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return gluonnlp_amd


def _tables(rng, V, NT, ML, U):
    word = rng.standard_normal((V, U)) * 0.5
    typ = rng.standard_normal((NT, U)) * 0.5
    pos = rng.standard_normal((ML, U)) * 0.5
    g = 1.0 + 0.1 * rng.standard_normal(U)
    b = 0.1 * rng.standard_normal(U)
    return word, typ, pos, g, b


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NT', 'TN'])
@pytest.mark.parametrize('id_dtype', [torch.int32, torch.int64])
def test_embedding_forward(nlp, dtype, layout, id_dtype):
    from gluonnlp_amd.layers import embedding_forward
    rng = np.random.default_rng(11)
    V, NT, ML, U, B, L = 97, 3, 40, 64, 5, 33
    word, typ, pos, g, b = _tables(rng, V, NT, ML, U)
    shape = (B, L) if layout == 'NT' else (L, B)
    ids = rng.integers(0, V, shape)
    tts = rng.integers(0, NT, shape)
    d = lambda a: to_dev(a, dtype)
    o = lambda a: prep(a, dtype)
    i_d = torch.from_numpy(ids).cuda().to(id_dtype)
    t_d = torch.from_numpy(tts).cuda().to(id_dtype)
    t = tol(dtype)
    # BERT / ELECTRA: three tables + LayerNorm
    got = embedding_forward(i_d, t_d, d(word), d(typ), d(pos), d(g), d(b), 1e-12, layout)
    want = ref.initial_embedding(ids, tts, o(word), o(typ), o(pos),
                                 {'gamma': o(g), 'beta': o(b)}, 1e-12, layout)
    assert got.shape == shape + (U,) and got.dtype == dtype
    assert rel_err(to_np(got), want) < t
    # token_types None == zeros, bit for bit
    z1 = embedding_forward(i_d, None, d(word), d(typ), d(pos), d(g), d(b), 1e-12, layout)
    z2 = embedding_forward(i_d, torch.zeros_like(i_d), d(word), d(typ), d(pos), d(g), d(b), 1e-12,
                           layout)
    assert torch.equal(z1, z2)
    # RoBERTa: no token types; with and without the embedding LayerNorm
    got = embedding_forward(i_d, None, d(word), None, d(pos), d(g), d(b), 1e-5, layout)
    want = ref.initial_embedding(ids, None, o(word), None, o(pos), {'gamma': o(g), 'beta': o(b)},
                                 1e-5, layout)
    assert rel_err(to_np(got), want) < t
    got = embedding_forward(i_d, None, d(word), None, d(pos), None, None, 0.0, layout)
    want = ref.initial_embedding(ids, None, o(word), None, o(pos), None, 0.0, layout)
    assert rel_err(to_np(got), want) < t
    # a bare gather is exact
    got = embedding_forward(i_d, None, d(word), None, None, None, None, 0.0, layout)
    assert torch.equal(got, d(word)[i_d.long()])


def test_embedding_clipping_and_errors(nlp):
    from gluonnlp_amd._lib import GnlpError
    from gluonnlp_amd.layers import embedding_forward
    rng = np.random.default_rng(12)
    V, NT, ML, U = 10, 2, 6, 32
    word, typ, pos, g, b = _tables(rng, V, NT, ML, U)
    d = lambda a: to_dev(a, torch.float32)
    # ids / types out of range and positions beyond max_length are clipped (np.take mode='clip')
    ids = np.array([[0, 9, 12, -3, 5, 5, 5, 5, 5]])
    tts = np.array([[0, 1, 7, -1, 0, 0, 0, 0, 1]])
    got = embedding_forward(torch.from_numpy(ids).cuda(), torch.from_numpy(tts).cuda(), d(word),
                            d(typ), d(pos), d(g), d(b), 1e-12, 'NT')
    want = ref.initial_embedding(ids, tts, word, typ, pos, {'gamma': g, 'beta': b}, 1e-12, 'NT')
    assert rel_err(to_np(got), want) < 1e-5
    # empty batch
    e = embedding_forward(torch.zeros((0, 4), dtype=torch.int64, device='cuda'), None, d(word),
                          None, None, None, None, 0.0, 'NT')
    assert e.shape == (0, 4, U)
    with pytest.raises(TypeError):
        embedding_forward(torch.zeros((2, 2), device='cuda'), None, d(word), None, None, None, None,
                          0.0, 'NT')
    with pytest.raises(RuntimeError):
        embedding_forward(torch.zeros((2, 2), dtype=torch.int64), None, d(word), None, None, None,
                          None, 0.0, 'NT')
    with pytest.raises(GnlpError):   # units not a multiple of the vector width
        embedding_forward(torch.zeros((2, 2), dtype=torch.int64, device='cuda'), None,
                          d(rng.standard_normal((V, 30))), None, None, None, None, 0.0, 'NT')


def _backbone_params(rng, V, NT, ML, U, F, NL, embed=None):
    E = U if embed is None else embed
    p = {'word_embed.weight': rng.standard_normal((V, E)) * 0.5,
         'token_type_embed.weight': rng.standard_normal((NT, E)) * 0.5,
         'token_pos_embed._embed.weight': rng.standard_normal((ML, E)) * 0.5,
         'embed_layer_norm.gamma': 1.0 + 0.1 * rng.standard_normal(E),
         'embed_layer_norm.beta': 0.1 * rng.standard_normal(E)}
    layers = [ref.make_encoder_layer_params(rng, U, F) for _ in range(NL)]
    return p, layers


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout,compute_layout', [('NT', 'auto'), ('TN', 'auto'), ('NT', 'TN')])
def test_bert_model_ids_to_pooled(nlp, dtype, layout, compute_layout):
    """BertModel.forward (models/bert.py:404-456): ids -> embeddings -> stack -> pooled CLS."""
    from gluonnlp_amd.models import BertModel
    rng = np.random.default_rng(21)
    V, NT, ML, U, F, H, NL, B, L = 120, 2, 48, 64, 128, 4, 2, 4, 24
    name = 'float32' if dtype == torch.float32 else 'bfloat16'
    m = BertModel(vocab_size=V, units=U, hidden_size=F, num_layers=NL, num_heads=H, max_length=ML,
                  num_token_types=NT, dtype=name, layout=layout, compute_layout=compute_layout)
    p, layers = _backbone_params(rng, V, NT, ML, U, F, NL)
    p['pooler.weight'] = rng.standard_normal((U, U)) * 0.1
    p['pooler.bias'] = 0.1 * rng.standard_normal(U)
    full = dict(p)
    for i, lp in enumerate(layers):
        full.update({'encoder.all_layers.{}.{}'.format(i, k): v for k, v in lp.items()})
    load_params(m, full, dtype)
    shape = (B, L) if layout == 'NT' else (L, B)
    ids, tts = rng.integers(0, V, shape), rng.integers(0, NT, shape)
    vl = np.array([L, 1, 7, L - 3])
    ctx, pooled = m(torch.from_numpy(ids).cuda(), torch.from_numpy(tts).cuda(),
                    torch.from_numpy(vl).cuda())
    o = lambda a: prep(a, dtype)
    emb = ref.initial_embedding(ids, tts, o(p['word_embed.weight']),
                                o(p['token_type_embed.weight']),
                                o(p['token_pos_embed._embed.weight']),
                                {'gamma': o(p['embed_layer_norm.gamma']),
                                 'beta': o(p['embed_layer_norm.beta'])}, 1e-12, layout)
    olayers = [{k: o(v) for k, v in lp.items()} for lp in layers]
    want_ctx, _ = ref.bert_transformer(emb, vl, olayers, H, 1e-12, 'gelu', layout)
    want_pool = ref.apply_pooling(want_ctx, o(p['pooler.weight']), o(p['pooler.bias']), 'tanh',
                                  layout)
    t = tol(dtype) * (1 if dtype == torch.float32 else 2)   # two layers of bf16 rounding + pooler
    assert rel_err(to_np(ctx), want_ctx) < t
    assert rel_err(to_np(pooled), want_pool) < t
    assert pooled.shape == (B, U)
    # padded tokens are exactly zero (npx.sequence_mask, bert.py:311-312)
    c = to_np(ctx) if layout == 'NT' else np.swapaxes(to_np(ctx), 0, 1)
    for bi in range(B):
        assert np.all(c[bi, vl[bi]:] == 0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_roberta_and_electra_models(nlp, dtype):
    from gluonnlp_amd.models import ElectraModel, RobertaModel
    rng = np.random.default_rng(22)
    V, NT, ML, U, F, H, NL, B, L, E = 90, 2, 40, 64, 128, 4, 2, 3, 20, 32
    name = 'float32' if dtype == torch.float32 else 'bfloat16'
    o = lambda a: prep(a, dtype)
    t = tol(dtype) * (1 if dtype == torch.float32 else 2)
    ids, tts = rng.integers(0, V, (B, L)), rng.integers(0, NT, (B, L))
    vl = np.array([L, 5, 13])
    dv = lambda a: torch.from_numpy(a).cuda()
    # ---- RoBERTa: word + position, embed_ln, pooler only with classifier_activation
    p, layers = _backbone_params(rng, V, NT, ML, U, F, NL)
    rp = {'word_embed.weight': p['word_embed.weight'],
          'pos_embed._embed.weight': p['token_pos_embed._embed.weight'],
          'embed_ln.gamma': p['embed_layer_norm.gamma'], 'embed_ln.beta': p['embed_layer_norm.beta'],
          'pooler.weight': rng.standard_normal((U, U)) * 0.1,
          'pooler.bias': 0.1 * rng.standard_normal(U)}
    full = dict(rp)
    for i, lp in enumerate(layers):
        full.update({'encoder.all_layers.{}.{}'.format(i, k): v for k, v in lp.items()})
    m = RobertaModel(vocab_size=V, units=U, hidden_size=F, num_layers=NL, num_heads=H,
                     max_length=ML, dtype=name, classifier_activation=True)
    load_params(m, full, dtype)
    ctx, pooled = m(dv(ids), dv(vl))
    emb = ref.initial_embedding(ids, None, o(rp['word_embed.weight']), None,
                                o(rp['pos_embed._embed.weight']),
                                {'gamma': o(rp['embed_ln.gamma']), 'beta': o(rp['embed_ln.beta'])},
                                1e-5)
    olayers = [{k: o(v) for k, v in lp.items()} for lp in layers]
    want_ctx, _ = ref.roberta_encoder(emb, vl, olayers, H, 1e-5, 'gelu')
    assert rel_err(to_np(ctx), want_ctx) < t
    assert rel_err(to_np(pooled), ref.apply_pooling(want_ctx, o(rp['pooler.weight']),
                                                    o(rp['pooler.bias']), 'tanh')) < t
    m2 = RobertaModel(vocab_size=V, units=U, hidden_size=F, num_layers=NL, num_heads=H,
                      max_length=ML, dtype=name)     # default: pooled = the CLS rows themselves
    load_params(m2, {k: v for k, v in full.items() if not k.startswith('pooler')}, dtype)
    ctx2, pooled2 = m2(dv(ids), dv(vl))
    assert torch.equal(pooled2, ctx2[:, 0, :])
    # ---- ELECTRA: factorised embedding (embed_size != units), pooled = CLS rows
    p, layers = _backbone_params(rng, V, NT, ML, U, F, NL, embed=E)
    p['embed_factorized_proj.weight'] = rng.standard_normal((U, E)) * 0.2
    p['embed_factorized_proj.bias'] = 0.1 * rng.standard_normal(U)
    full = dict(p)
    for i, lp in enumerate(layers):
        full.update({'encoder.all_encoder_layers.{}.{}'.format(i, k): v for k, v in lp.items()})
    m = ElectraModel(vocab_size=V, units=U, hidden_size=F, embed_size=E, num_layers=NL,
                     num_heads=H, max_length=ML, num_token_types=NT, dtype=name)
    load_params(m, full, dtype)
    ctx, pooled = m(dv(ids), dv(tts), dv(vl))
    emb = ref.initial_embedding(ids, tts, o(p['word_embed.weight']),
                                o(p['token_type_embed.weight']),
                                o(p['token_pos_embed._embed.weight']),
                                {'gamma': o(p['embed_layer_norm.gamma']),
                                 'beta': o(p['embed_layer_norm.beta'])}, 1e-12)
    if dtype == torch.bfloat16:
        emb = prep(emb, dtype)      # the device holds the embedding in bf16 before the projection
    proj = ref.dense(emb, o(p['embed_factorized_proj.weight']), o(p['embed_factorized_proj.bias']))
    olayers = [{k: o(v) for k, v in lp.items()} for lp in layers]
    want_ctx, _ = ref.electra_encoder(proj, vl, olayers, H, 1e-12, 'gelu')
    assert rel_err(to_np(ctx), want_ctx) < t
    assert torch.equal(pooled, ctx[:, 0, :])


# ------------------------------------------------------------------ decoder
import os  # noqa: E402

G = os.path.join(os.path.dirname(__file__), 'golden')


def _dec_params(g, li):
    pre = 'L{}/'.format(li)
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('name,H,act', [('dec_tiny', 2, 'relu'), ('dec_d64', 2, 'gelu')])
def test_decoder_layer_goldens(nlp, dtype, name, H, act):
    """TransformerDecoderLayer.forward (models/transformer.py:513-598) against the goldens of
    the torch sibling (fp32) / the oracle on bf16-rounded inputs (bf16); dec_d64 runs the fused
    attention kernel with causal and memory byte masks and Lq != Lm."""
    from gluonnlp_amd.models import TransformerDecoderLayer
    g = np.load(os.path.join(G, name + '.npz'))
    x, mem = g['x'], g['mem']
    U, MU = x.shape[-1], mem.shape[-1]
    F = g['L0/ffn.ffn_1.weight'].shape[0]
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    causal = torch.from_numpy(g['self_causal_mask']).cuda()
    mm = torch.from_numpy(g['mem_attn_mask']).cuda()
    h = to_dev(x, dtype)
    md = to_dev(mem, dtype)
    ho = prep(x, dtype)
    t = tol(dtype) * (1 if dtype == torch.float32 else 2)
    for li in range(2):
        p = _dec_params(g, li)
        layer = TransformerDecoderLayer(units=U, mem_units=MU, hidden_size=F, num_heads=H,
                                        layer_norm_eps=1e-5, activation=act, dtype=dn)
        load_params(layer, p, dtype)
        h = layer(h, md, causal, mm)
        if dtype == torch.float32:
            want = g['out_L{}'.format(li)]
        else:
            ho = ref.transformer_decoder_layer(ho, prep(mem, dtype), g['self_causal_mask'],
                                               g['mem_attn_mask'],
                                               {k: prep(v, dtype) for k, v in p.items()}, H, 1e-5,
                                               False, act)
            want = ho
        assert rel_err(to_np(h), want) < t, li


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout,pre_norm,recurrent', [('NT', False, False), ('TN', True, False),
                                                       ('NT', True, True)])
def test_transformer_decoder(nlp, dtype, layout, pre_norm, recurrent):
    """TransformerDecoder.forward (models/transformer.py:789-840): masks from the valid lengths,
    data_norm, pre_norm with the MX residual, recurrent weight sharing, both layouts."""
    from gluonnlp_amd.models import TransformerDecoder
    rng = np.random.default_rng(41)
    U, MU, F, H, NL, B, L, LM = 64, 48, 128, 4, 2, 3, 18, 26
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    m = TransformerDecoder(num_layers=NL, recurrent=recurrent, units=U, mem_units=MU,
                           hidden_size=F, num_heads=H, data_norm=True, pre_norm=pre_norm,
                           activation='gelu', dtype=dn, layout=layout)

    def layer_params():
        lin = lambda n, k: ref.trunc_norm(rng, (n, k), 0.05)
        vec = lambda n, c=0.0: (c + 0.1 * rng.standard_normal(n)).astype(np.float32)
        p = {}
        for nm, n, k in [('attn_in_qkv', 3 * U, U), ('proj_in', U, U), ('attn_inter_q', U, U),
                         ('attn_inter_k', U, MU), ('attn_inter_v', U, MU), ('proj_inter', U, U),
                         ('ffn.ffn_1', F, U), ('ffn.ffn_2', U, F)]:
            p[nm + '.weight'], p[nm + '.bias'] = lin(n, k), vec(n)
        for nm in ['ln_in', 'ln_inter', 'ffn.layer_norm']:
            p[nm + '.gamma'], p[nm + '.beta'] = vec(U, 1.0), vec(U)
        return p
    layers = [layer_params() for _ in range(1 if recurrent else NL)]
    ln_data = {'gamma': (1 + 0.1 * rng.standard_normal(U)).astype(np.float32),
               'beta': (0.1 * rng.standard_normal(U)).astype(np.float32)}
    ln_final = {'gamma': (1 + 0.1 * rng.standard_normal(U)).astype(np.float32),
                'beta': (0.1 * rng.standard_normal(U)).astype(np.float32)}
    full = {'ln_data.gamma': ln_data['gamma'], 'ln_data.beta': ln_data['beta']}
    if pre_norm:
        full.update({'ln_final.gamma': ln_final['gamma'], 'ln_final.beta': ln_final['beta']})
    for i, lp in enumerate(layers):
        full.update({'layers.{}.{}'.format(i, k): v for k, v in lp.items()})
    load_params(m, full, dtype)
    shape = (B, L, U) if layout == 'NT' else (L, B, U)
    mshape = (B, LM, MU) if layout == 'NT' else (LM, B, MU)
    x, mem = rng.standard_normal(shape), rng.standard_normal(mshape)
    vl, mvl = np.array([L, 1, 11]), np.array([LM, 9, 1])
    got = m(to_dev(x, dtype), torch.from_numpy(vl).cuda(), to_dev(mem, dtype),
            torch.from_numpy(mvl).cuda())
    o = lambda a: prep(a, dtype)
    want = ref.transformer_decoder(
        o(x), vl, o(mem), mvl, [{k: o(v) for k, v in lp.items()} for lp in layers], H, 1e-5,
        pre_norm, 'gelu', layout, recurrent=recurrent, num_layers=NL,
        ln_data={k: o(v) for k, v in ln_data.items()},
        ln_final={k: o(v) for k, v in ln_final.items()})
    t = tol(dtype) * (1 if dtype == torch.float32 else 3)   # NL layers x 3 sub-blocks of bf16
    assert rel_err(to_np(got), want) < t
    # incremental decoding (models/transformer.py:884-959): feeding the target one position at a
    # time, with the key / value caches, reproduces the full forward on every valid position
    xd, memd = to_dev(x, dtype), to_dev(mem, dtype)
    mvld = torch.from_numpy(mvl).cuda()
    states = m.init_states(B, 'cuda', dn)
    assert m.state_batch_axis == [(0, 0) if layout == 'NT' else (1, 1)] * NL
    tax = 1 if layout == 'NT' else 0
    steps = []
    for i in range(L):
        step_in = xd[:, i] if layout == 'NT' else xd[i]
        out_i, states = m.incremental_decode(step_in, states, memd, mvld)
        assert tuple(out_i.shape) == (B, U)
        assert states[0][0].shape[tax] == i + 1
        steps.append(to_np(out_i))
    inc = np.stack(steps, axis=tax)
    full = to_np(got)
    for b in range(B):
        sl = (b, slice(0, vl[b])) if layout == 'NT' else (slice(0, vl[b]), b)
        assert rel_err(inc[sl], full[sl]) < (1e-5 if dtype == torch.float32 else 2e-2), b


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NT', 'TN'])
@pytest.mark.parametrize('name', ['bert_hf_d64', 'bert_hf_h4'])
def test_bert_model_against_hf_goldens(nlp, name, layout, dtype):
    """BertModel (ids -> contextual embeddings, pooled) against HuggingFace's BertModel outputs
    (tests/golden/bert_hf_*.npz, oracle/gen_golden_bert_hf.py) in fp32; in bf16 against the oracle
    -- pinned to the same goldens at 1e-10 -- on bf16-rounded parameters."""
    from gluonnlp_amd.models import BertModel
    from tests.test_oracle_golden import bert_from_ids
    g = np.load(os.path.join(G, name + '.npz'))
    p = {k[2:]: g[k] for k in g.files if k.startswith('p:')}
    V, U = p['word_embed.weight'].shape
    NL, H = int(g['num_layers']), int(g['num_heads'])
    F = p['encoder.all_layers.0.ffn.ffn_1.weight'].shape[0]
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    m = BertModel(vocab_size=V, units=U, hidden_size=F, num_layers=NL, num_heads=H,
                  max_length=p['token_pos_embed._embed.weight'].shape[0], dtype=dn, layout=layout)
    load_params(m, p, dtype)
    vl = g['valid_length']
    ids, tt = g['ids'], g['token_types']
    if layout == 'TN':
        ids, tt = ids.T, tt.T
    seq, pooled = m(torch.from_numpy(np.ascontiguousarray(ids)).cuda(),
                    torch.from_numpy(np.ascontiguousarray(tt)).cuda(),
                    torch.from_numpy(vl).cuda())
    seq_n = to_np(seq)
    if layout == 'TN':
        seq_n = seq_n.transpose(1, 0, 2)
    if dtype == torch.float32:
        want_seq, want_pooled, t = g['sequence'], g['pooled'], 1e-5
    else:
        want_seq, want_pooled = bert_from_ids({k: prep(v, dtype) for k, v in p.items()}, g['ids'],
                                              g['token_types'], vl, NL, H)
        t = 2e-2 * 1.5
    for b in range(len(vl)):
        assert rel_err(seq_n[b, :vl[b]], want_seq[b, :vl[b]]) < t, b
        assert (seq_n[b, vl[b]:] == 0).all()
    assert rel_err(to_np(pooled), want_pooled) < t


@pytest.mark.parametrize('suffix', ['.params', '.state_dict.pt'])
def test_bert_model_from_reference_layout_parameter_file(nlp, suffix):
    """a LOCAL parameter file in the reference's on-disk layout
    (tests/golden/bert_hf_h4.params: MXNet 2.x save_parameters' npz container, float32, structural
    names; and the torch state_dict form -- oracle/gen_params_fixture.py) is read by
    utils.load_parameters into BertModel, and token ids -> (contextual embeddings, pooled output)
    then match HuggingFace's outputs for those weights (fp32, 1e-5) and the oracle on
    bf16-rounded parameters (bf16)."""
    from gluonnlp_amd.models import BertModel
    from gluonnlp_amd.utils import load_parameters
    from tests.test_oracle_golden import bert_from_ids
    g = np.load(os.path.join(G, 'bert_hf_h4.npz'))
    p = {k[2:]: g[k] for k in g.files if k.startswith('p:')}
    V, U = p['word_embed.weight'].shape
    NL, H = int(g['num_layers']), int(g['num_heads'])
    F = p['encoder.all_layers.0.ffn.ffn_1.weight'].shape[0]
    vl, ids, tt = g['valid_length'], g['ids'], g['token_types']
    for dtype, dn in ((torch.float32, 'float32'), (torch.bfloat16, 'bfloat16')):
        m = BertModel(vocab_size=V, units=U, hidden_size=F, num_layers=NL, num_heads=H,
                      max_length=p['token_pos_embed._embed.weight'].shape[0], dtype=dn).cuda()
        load_parameters(m, os.path.join(G, 'bert_hf_h4' + suffix))
        seq, pooled = m(torch.from_numpy(ids).cuda(), torch.from_numpy(tt).cuda(),
                        torch.from_numpy(vl).cuda())
        seq_n = to_np(seq)
        if dtype == torch.float32:
            want_seq, want_pooled, t = g['sequence'], g['pooled'], 1e-5
        else:
            f32 = {k: v.astype(np.float32) for k, v in p.items()}      # what the file holds
            want_seq, want_pooled = bert_from_ids({k: prep(v, dtype) for k, v in f32.items()}, ids,
                                                  tt, vl, NL, H)
            t = 2e-2 * 1.5
        for b in range(len(vl)):
            assert rel_err(seq_n[b, :vl[b]], want_seq[b, :vl[b]]) < t, (dn, b)
            assert (seq_n[b, vl[b]:] == 0).all()
        assert rel_err(to_np(pooled), want_pooled) < t


@pytest.mark.parametrize('kind', ['transformer', 't5'])
def test_incremental_decode_keeps_memory_projections(nlp, kind):
    """During incremental decoding the memory's key / value projections are computed once and
    travel with the decoding state instead of being recomputed per step (the reference recomputes
    them, models/transformer.py:716-724, models/t5.py:462-470: same values).  The kept pair is
    used only for the very same memory tensor object with unchanged contents and parameters, from
    the state the previous step returned: an in-place edit of the memory, an edited weight,
    another tensor or a reordered state recompute -- and every step agrees bit for bit with
    layers that never keep anything."""
    from gluonnlp_amd import layers as L
    from gluonnlp_amd._tensor import as_tensor
    from gluonnlp_amd.models import T5Decoder, TransformerDecoder
    import gluonnlp_amd.models.t5 as m5
    import gluonnlp_amd.models.transformer as mt
    torch.manual_seed(3)
    B, LM, U = 2, 9, 64
    if kind == 'transformer':
        m = TransformerDecoder(num_layers=2, units=U, mem_units=U, hidden_size=128, num_heads=2,
                               dtype='bfloat16', layout='NT').cuda()
        kd = m.layers[0].attn_inter_k
        init = lambda: m.init_states(B, 'cuda', 'bfloat16')
        step = lambda i, x, st, mem, mvl: m.incremental_decode(x[:, 0], st, mem, mvl)
    else:
        m = T5Decoder(d_model=U, d_kv=32, d_ff=128, num_layers=2, num_heads=2, dtype='bfloat16',
                      layout='NT').cuda()
        kd = m.layers[0].cross_attn_k
        init = lambda: m._init_key_values(B, 'cuda', 'bfloat16')
        step = lambda i, x, st, mem, mvl: m.incremental_decode(
            x, torch.full((B,), i, dtype=torch.int32, device='cuda'), st, mem, mvl)
    for p in m.parameters():
        p.data.normal_(0, 0.1)
    mem = torch.randn(B, LM, U, device='cuda').to(torch.bfloat16)
    mvl = torch.tensor([LM, 4], device='cuda')
    xs = [torch.randn(B, 1, U, device='cuda').to(torch.bfloat16) for _ in range(6)]
    calls = []
    hook = kd.register_forward_hook(lambda *a: calls.append(1))
    real = L.memory_kv
    fresh = lambda prev, mem_, k, v: (as_tensor(k(mem_)), as_tensor(v(mem_)), None)

    def decode(keep, edits):
        """six steps; edits: step index -> callable run before that step (may return new states)"""
        mt.memory_kv = m5.memory_kv = real if keep else fresh
        try:
            st, outs, counts = init(), [], []
            for i, x in enumerate(xs):
                if i in edits:
                    st = edits[i](st) or st
                o, st = step(i, x, st, mem, mvl)
                outs.append(o.clone())
                counts.append(len(calls))
            return outs, counts
        finally:
            mt.memory_kv = m5.memory_kv = real

    def edit_mem(st):
        mem[:, 0] += 1.0                     # in place: same object, new version

    def edit_weight(st):
        with torch.no_grad():
            kd.weight.mul_(0.5)

    def reorder(st):                         # what a beam search does to the states
        perm = torch.tensor([1, 0], device='cuda')
        return [tuple(t.index_select(0, perm) for t in pair) for pair in st]
    edits = {2: edit_mem, 3: edit_weight, 5: reorder}
    try:
        snap = {k: v.clone() for k, v in m.state_dict().items()}
        mem0 = mem.clone()
        calls.clear()
        kept, counts = decode(True, edits)
        # computed at steps 0 (first), 2 (memory edited), 3 (weight edited), 5 (reordered state)
        assert [b - a for a, b in zip([0] + counts, counts)] == [1, 0, 1, 1, 0, 1]
        m.load_state_dict(snap)
        mem.copy_(mem0)
        calls.clear()
        ref_outs, counts = decode(False, edits)
        assert counts[-1] == 6
        for a, b in zip(kept, ref_outs):
            assert torch.equal(a, b)
    finally:
        hook.remove()


def test_t5_packed_qkv_follows_parameter_edits(nlp):
    """The encoder blocks' packed [q; k; v] weight buffer is the parameters' own storage, not a
    copy: whatever way the parameters are edited after a first forward (load_state_dict,
    p.data.copy_, in-place arithmetic, .to()), the next forward uses the new values."""
    from gluonnlp_amd.models import T5Encoder
    torch.manual_seed(5)
    mk = lambda: T5Encoder(64, 32, 128, num_layers=2, num_heads=2, activation='relu',
                           dtype='bfloat16', layout='NT').cuda()
    a, b = mk(), mk()
    for mod in (a, b):
        for p in mod.parameters():
            p.data.normal_(0, 0.1)
    x = torch.randn(2, 10, 64, device='cuda').to(torch.bfloat16)
    vl = torch.tensor([10, 6], device='cuda')
    ya0, yb0 = a(x, vl).clone(), b(x, vl).clone()          # both have packed their weights now
    assert not torch.equal(ya0, yb0)
    q = a.layers[0].self_attn_q.weight
    assert q.data_ptr() == a.layers[0].packed_qkv().data_ptr()
    a.load_state_dict(b.state_dict())                      # param.copy_ under no_grad
    assert torch.equal(a(x, vl), yb0)
    for pa in a.parameters():
        pa.data.mul_(0.5)                                  # invisible to version counters
    for pb in b.parameters():
        pb.data.mul_(0.5)
    yb1 = b(x, vl).clone()
    assert torch.equal(a(x, vl), yb1) and not torch.equal(yb1, yb0)
    a.layers[0].self_attn_k.weight.data = b.layers[1].self_attn_k.weight.data.clone()   # re-seated
    b.layers[0].self_attn_k.weight.data.copy_(b.layers[1].self_attn_k.weight.data)
    assert torch.equal(a(x, vl), b(x, vl))
    sd = {k: v.clone() for k, v in a.state_dict().items()}   # round trip through a state dict
    c = mk()
    c.load_state_dict(sd)
    assert torch.equal(c(x, vl), a(x, vl))


def test_memory_projections_and_dot_data_edits(nlp):
    """The documented limit of the kept memory projections (layers.memory_kv): an in-place write
    through ``.data`` bumps no version counter, so the next step still serves the projections of
    the OLD weights -- until drop_memory_kv(states); pinned so that the behaviour cannot change
    silently."""
    from gluonnlp_amd.layers import drop_memory_kv
    from gluonnlp_amd.models import TransformerDecoder
    torch.manual_seed(5)
    B, LM, U = 2, 6, 64
    m = TransformerDecoder(num_layers=1, units=U, mem_units=U, hidden_size=128, num_heads=2,
                           dtype='bfloat16', layout='NT').cuda()
    for p in m.parameters():
        p.data.normal_(0, 0.1)
    mem = torch.randn(B, LM, U, device='cuda').to(torch.bfloat16)
    mvl = torch.tensor([LM, 3], device='cuda')
    x = torch.randn(B, U, device='cuda').to(torch.bfloat16)
    st = m.init_states(B, 'cuda', 'bfloat16')
    _, st = m.incremental_decode(x, st, mem, mvl)
    m.layers[0].attn_inter_k.weight.data.mul_(-3.0)          # in place through .data
    stale, st_stale = m.incremental_decode(x, st, mem, mvl)
    drop_memory_kv(st)
    fresh, _ = m.incremental_decode(x, st, mem, mvl)
    ref_, _ = m.incremental_decode(x, [type(s_)(s_) if isinstance(s_, (list, tuple)) else s_.clone()
                                       for s_ in st], mem, mvl)   # cloned states carry nothing
    assert torch.equal(fresh, ref_)
    assert not torch.equal(stale, fresh)

