"""T5 decoder / T5Model on the MI355X (models/t5.py:715-1189) through the C ABI:
against the HuggingFace-derived goldens (the reference's own conversion check,
scripts/conversion_toolkits/convert_t5.py:203-232), against the oracle at the fused-kernel head
size, and incremental decoding against the full forward."""
import os

import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import load_params, prep, rel_err, to_dev, to_np, tol

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    from gluonnlp_amd import _lib
    _lib.load()
    return gluonnlp_amd


def _full_params(p):
    """Reference names; the RMSNorm betas (zero, t5.py:166-172) are not in a converted file."""
    return dict(p)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NT', 'TN'])
@pytest.mark.parametrize('name', ['t5_hf_relu', 't5_hf_gated'])
def test_t5_model_against_hf_goldens(nlp, name, layout, dtype):
    from gluonnlp_amd.models import T5Model
    g = np.load(os.path.join(G, name + '.npz'))
    p = {k[2:]: g[k] for k in g.files if k.startswith('p:')}
    V, D = p['input_embedding_layer.weight'].shape
    NL, H, gated = int(g['num_layers']), int(g['num_heads']), int(g['gated'])
    inner = p['encoder.layers.0.self_attn_q.weight'].shape[0]
    F = p['encoder.layers.0.ffn.ffn_1.weight'].shape[0]
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    m = T5Model(vocab_size=V, d_model=D, d_kv=inner // H, d_ff=F, num_layers=NL, num_heads=H,
                activation='gated-gelu' if gated else 'relu', layout=layout, dtype=dn)
    for q in m.parameters():       # betas of the RMSNorms are zero in the reference (center=False)
        q.data.zero_()
    load_params(m, p, dtype)
    svl, tvl = g['src_valid_length'], g['tgt_valid_length']
    src, tgt = g['src'], g['tgt']
    if layout == 'TN':
        src, tgt = src.T, tgt.T
    sd, td = torch.from_numpy(np.ascontiguousarray(src)).cuda(), torch.from_numpy(
        np.ascontiguousarray(tgt)).cuda()
    svd, tvd = torch.from_numpy(svl).cuda(), torch.from_numpy(tvl).cuda()
    enc = m.encode(sd, svd)
    dec = m(sd, svd, td, tvd)
    enc_n, dec_n = to_np(enc), to_np(dec)
    if layout == 'TN':
        enc_n, dec_n = enc_n.transpose(1, 0, 2), dec_n.transpose(1, 0, 2)
    if dtype == torch.float32:
        want_enc, want_dec = g['enc_out'], g['dec_out']
        t_enc = t_dec = 1e-5
    else:   # bf16: the oracle (pinned to these goldens at 5e-6) on bf16-rounded parameters
        want_enc, want_dec = ref.t5_model(g['src'], svl, g['tgt'], tvl,
                                          {k: prep(v, dtype) for k, v in p.items()}, NL, H,
                                          activation_name='gated-gelu' if gated else 'relu')
        t_enc, t_dec = 2e-2 * 2, 2e-2 * 3          # 2 / 4 stacked pre-norm stacks of bf16 hand-offs
    for b in range(len(svl)):
        assert rel_err(enc_n[b, :svl[b]], want_enc[b, :svl[b]]) < t_enc, ('enc', b)
        assert rel_err(dec_n[b, :tvl[b]], want_dec[b, :tvl[b]]) < t_dec, ('dec', b)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NT', 'TN'])
def test_t5_decoder_d64_and_incremental(nlp, layout, dtype):
    """d_kv = 64 (the fused attention kernel in bf16: causal byte mask + unidirectional bucket
    bias, memory mask with Lq != Lm) against the oracle; then incremental_decode, one position at
    a time with the key / value caches, against the full forward on the valid positions
    (models/t5.py:812-893)."""
    from gluonnlp_amd.models import T5Decoder
    rng = np.random.default_rng(77)
    D, DKV, F, H, NL, B, L, LM = 128, 64, 256, 2, 2, 3, 40, 72
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    m = T5Decoder(d_model=D, d_kv=DKV, d_ff=F, num_layers=NL, num_heads=H, activation='gated-gelu',
                  layout=layout, dtype=dn)
    layers = [ref.make_t5_block_params(rng, D, DKV, F, H, gated=True, decoder=True)
              for _ in range(NL)]
    relw = (0.5 * rng.standard_normal((32, H))).astype(np.float32)
    fin = {'gamma': (1 + 0.1 * rng.standard_normal(D)).astype(np.float32),
           'beta': np.zeros(D, np.float32)}
    full = {'relative_position_encoder._rel_pos_embed.weight': relw,
            'final_layer_norm.gamma': fin['gamma'], 'final_layer_norm.beta': fin['beta']}
    for i, lp in enumerate(layers):
        full.update({'layers.{}.{}'.format(i, k): v for k, v in lp.items()})
    load_params(m, full, dtype)
    shape = (B, L, D) if layout == 'NT' else (L, B, D)
    mshape = (B, LM, D) if layout == 'NT' else (LM, B, D)
    x, mem = rng.standard_normal(shape), rng.standard_normal(mshape)
    vl, mvl = np.array([L, 1, 23]), np.array([LM, 33, 1])
    xd, memd = to_dev(x, dtype), to_dev(mem, dtype)
    vld, mvld = torch.from_numpy(vl).cuda(), torch.from_numpy(mvl).cuda()
    got = m(xd, vld, memd, mvld)
    o = lambda a: prep(a, dtype)
    want = ref.t5_decoder(o(x), vl, o(mem), mvl, o(relw),
                          [{k: o(v) for k, v in lp.items()} for lp in layers],
                          {k: o(v) for k, v in fin.items()}, H, activation_name='gated-gelu',
                          layout=layout)
    full_out = to_np(got)
    tax = 1 if layout == 'NT' else 0
    sel = lambda a, b: a[b, :vl[b]] if layout == 'NT' else a[:vl[b], b]
    t = tol(dtype) * (1 if dtype == torch.float32 else 3)
    for b in range(B):
        assert rel_err(sel(full_out, b), sel(want, b)) < t, b
    # incremental decoding
    past = m._init_key_values(B, 'cuda', dn)
    assert m.state_batch_axis == [(0, 0) if layout == 'NT' else (1, 1)] * NL
    steps = []
    for i in range(L):
        step = xd[:, i:i + 1] if layout == 'NT' else xd[i:i + 1]
        pos = torch.full((B,), i, dtype=torch.int32, device='cuda')
        out_i, past = m.incremental_decode(step, pos, past, memd, mvld)
        assert past[0][0].shape[tax] == i + 1
        steps.append(to_np(out_i))
    inc = np.concatenate(steps, axis=tax)
    for b in range(B):
        assert rel_err(sel(inc, b), sel(full_out, b)) < (1e-5 if dtype == torch.float32 else 2e-2), b


@pytest.mark.parametrize('bidirectional', [True, False])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_t5_bias_table_is_the_toeplitz_form_of_the_bias(nlp, dtype, bidirectional):
    """gnlp_t5_rel_bias_table: bias[h, i, j] == table[h, (Lq - 1) + j - i], bit for bit, for the
    (H, Lq, Lm) bias of RelAttentionScoreCell('t5') (attention_cell.py:876-878)."""
    from gluonnlp_amd.attention_cell import RelAttentionScoreCell
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    H, Lq, Lm = 3, 150, 211
    cell = RelAttentionScoreCell(query_units=H * 8, num_heads=H, method='t5',
                                 bidirectional=bidirectional, dtype=dn).cuda()
    cell._rel_pos_embed.weight.data = torch.randn(32, H, device='cuda').to(dtype)
    dense = cell.self_attn_bias(Lq, Lm)
    table = cell.self_attn_bias_table(Lq, Lm)
    assert tuple(table.shape) == (H, Lq + Lm - 1)
    idx = (Lq - 1) + torch.arange(Lm, device='cuda')[None, :] - torch.arange(Lq, device='cuda')[:, None]
    assert torch.equal(table[:, idx], dense)
