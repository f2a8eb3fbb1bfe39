"""GPU tests of the stacks run with their LayerNorms FOLDED into the GEMMs around them
(gnlp_encoder_stack_forward_folded; the reference runs nn.LayerNorm as its own operator,
models/transformer.py:268-270, layers.py:642-646 -- the same arithmetic regrouped): the helper
kernels against numpy, the folded stack against the float64 oracle and against the plain stack
(GNLP_LN_FOLD=0), rows whose mean dwarfs their spread, packed ragged batches, and the lifetime of
the folded parameters.  bf16 within 2e-2 per layer (sqrt(layers) x 2e-2 end to end, as the other
stack tests)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import load_params, oracle_params, prep, rel_err, to_dev, to_np

pytestmark = pytest.mark.gpu

from gluonnlp_amd import _lib  # noqa: E402


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    _lib.load()
    assert torch.cuda.is_available()
    return gluonnlp_amd


@pytest.fixture
def tiles256():
    """The folded path needs the persistent GEMM, which the dispatcher gives to GEMMs of at least
    128 tiles of 256 x 256; the developer knob hands it the small shapes of these tests too."""
    _lib.set_knob('GNLP_GEMM_TILE', '256')
    yield
    _lib.set_knob('GNLP_GEMM_TILE', None)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_fold_norm_dense_kernel(nlp):
    """wf = bf16(gamma * w [* scale]) bit for bit, c = the row sums of the ROUNDED wf, bf = beta . w^T
    + b; with and without bias / beta, a scaled leading block of rows, K not a multiple of 512."""
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(3)
    for N, K, has_b, has_beta, n_scaled, scale in [(3072, 1024, True, True, 0, 1.0),
                                                   (100, 264, False, True, 0, 1.0),
                                                   (64, 768, True, False, 0, 1.0),
                                                   (96, 512, True, True, 32, 0.1803368801)]:
        w = (torch.randn(N, K, device='cuda', generator=g) * 0.05).to(torch.bfloat16)
        b = (torch.randn(N, device='cuda', generator=g) * 0.1).to(torch.bfloat16) if has_b else None
        gamma = (1 + 0.2 * torch.randn(K, device='cuda', generator=g)).to(torch.bfloat16)
        beta = (0.2 * torch.randn(K, device='cuda', generator=g)).to(torch.bfloat16) if has_beta else None
        wf = torch.empty_like(w)
        c = torch.empty(N, dtype=torch.float32, device='cuda')
        bf = torch.empty(N, dtype=torch.float32, device='cuda')
        _lib.check(lib.gnlp_fold_norm_dense(
            wf.data_ptr(), c.data_ptr(), bf.data_ptr(), w.data_ptr(),
            b.data_ptr() if has_b else None, gamma.data_ptr(),
            beta.data_ptr() if has_beta else None, _lib.BF16, N, K, n_scaled, scale, _stream()))
        sc = torch.ones(N, 1, device='cuda')
        sc[:n_scaled] = scale
        want_wf = (gamma.float()[None, :] * w.float() * sc).to(torch.bfloat16)
        assert torch.equal(wf, want_wf)
        want_c = wf.double().sum(1)
        assert float((c.double() - want_c).abs().max()) <= 1e-5 * float(want_c.abs().max()) + 1e-6
        want_bf = torch.zeros(N, dtype=torch.float64, device='cuda')
        if has_beta:
            want_bf += w.double() @ beta.double()
        if has_b:
            want_bf += b.double()
        want_bf *= sc[:, 0].double()
        assert float((bf.double() - want_bf).abs().max()) <= 1e-5 * float(want_bf.abs().max()) + 1e-6
    with pytest.raises(_lib.GnlpError):
        _lib.check(lib.gnlp_fold_norm_dense(wf.data_ptr(), c.data_ptr(), bf.data_ptr(), w.data_ptr(),
                                            None, gamma.data_ptr(), None, _lib.F32, 8, 8, 0, 1.0,
                                            _stream()))


def test_row_stats_kernel(nlp):
    """(r, r * mean) per row, two-pass statistics: rows of 256 / 768 / 1024 / 2048, an offset 30x
    the spread, eps 1e-12 and 1e-5; wider or unaligned rows are refused."""
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(4)
    for C, eps, off in [(256, 1e-12, 0.0), (768, 1e-5, 3.0), (1024, 1e-12, 30.0), (2048, 1e-5, -7.0)]:
        rows = 777
        x = (torch.randn(rows, C, device='cuda', generator=g) + off).to(torch.bfloat16)
        rs = torch.empty(rows, 2, dtype=torch.float32, device='cuda')
        _lib.check(lib.gnlp_row_stats(rs.data_ptr(), x.data_ptr(), _lib.BF16, eps, rows, C, _stream()))
        xd = x.double()
        mean = xd.mean(1)
        r = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + eps)
        assert float(((rs[:, 0].double() - r) / r).abs().max()) < 2e-6
        assert float((rs[:, 1].double() - r * mean).abs().max()) < 2e-6 * float((r * mean).abs().max()) + 1e-6
    x = torch.zeros(4, 4096, dtype=torch.bfloat16, device='cuda')
    rs = torch.empty(4, 2, dtype=torch.float32, device='cuda')
    with pytest.raises(_lib.GnlpError):
        _lib.check(lib.gnlp_row_stats(rs.data_ptr(), x.data_ptr(), _lib.BF16, 1e-5, 4, 4096, _stream()))


def _bert(U, F, H, NL, rng, layout='NT', stdev=0.02):
    from gluonnlp_amd.models import BertTransformer
    m = BertTransformer(units=U, hidden_size=F, num_layers=NL, num_heads=H, dtype='bfloat16',
                        layout=layout).cuda()
    layers = [ref.make_encoder_layer_params(rng, U, F, stdev=stdev) for _ in range(NL)]
    for i, lp in enumerate(layers):
        load_params(m.all_layers[i], lp, torch.bfloat16)
    return m, layers


def _plain(m, *args):
    try:
        _lib.set_knob('GNLP_LN_FOLD', '0')
        out = m(*args)[0]
        assert _lib.load().gnlp_stack_last_folded() == 0
    finally:
        _lib.set_knob('GNLP_LN_FOLD', None)
    return out


@pytest.mark.parametrize('U,F,H,layout', [(256, 1024, 4, 'NT'), (768, 3072, 12, 'NT'),
                                          (1024, 4096, 16, 'TN')])
def test_folded_stack_matches_oracle_and_plain_stack(nlp, tiles256, U, F, H, layout):
    """BertTransformer (3 layers, random gamma / beta / biases, ragged lengths, final sequence
    mask) with the folded path: it ran (gnlp_stack_last_folded), every layer's worth of error stays
    within the stack tolerance against the float64 oracle, and it is no further from the oracle
    than the plain stack is (+ a quarter of the tolerance)."""
    rng = np.random.default_rng(U)
    B, L, NL = 4, 128, 3
    m, layers = _bert(U, F, H, NL, rng, layout)
    x = rng.standard_normal((B, L, U))
    vl = np.array([L, 1, 77, 100])
    xs = x if layout == 'NT' else np.transpose(x, (1, 0, 2))
    xd, vd = to_dev(xs, torch.bfloat16), torch.tensor(vl, device='cuda')
    got = m(xd, vd)[0]
    assert _lib.load().gnlp_stack_last_folded() == 1
    plain = _plain(m, xd, vd)
    exp, _ = ref.bert_transformer(prep(xs, torch.bfloat16), vl,
                                  [oracle_params(lp, torch.bfloat16) for lp in layers], H,
                                  layout=layout)
    e_fold, e_plain = rel_err(to_np(got), exp), rel_err(to_np(plain), exp)
    print('[fold] U=%d %s: folded %.3e, plain %.3e against the oracle' % (U, layout, e_fold, e_plain))
    assert e_fold < np.sqrt(NL) * 2e-2
    assert e_fold < e_plain + 5e-3
    o = to_np(got) if layout == 'NT' else np.transpose(to_np(got), (1, 0, 2))
    live = np.arange(L)[None, :] < vl[:, None]
    assert np.all(o[~live] == 0.0)


@pytest.mark.parametrize('offset', [3.0, 10.0])
def test_folded_stack_rows_with_large_mean(nlp, tiles256, offset):
    """The cancellation case of the fold: r (s . W') - (r mu) c with |mu| = 3 and 10 sigma (the
    input rows carry that offset, so the first layer's pre-norm sum does too).  What bounds the
    result there is the bf16 STORAGE of the sum, which both paths and a bf16 run of the reference
    share (half an ulp of 10 is 3 % of the row's spread): at 3 sigma the stack tolerance holds, at
    10 sigma neither path meets it (measured 5.3e-2 folded, 5.4e-2 plain) and the folded run is
    asserted against the plain one -- the fold itself (fp32 accumulation of s . W', c summed over
    the rounded W') adds nothing."""
    rng = np.random.default_rng(99)
    B, L, U, F, H, NL = 2, 128, 512, 2048, 8, 2
    m, layers = _bert(U, F, H, NL, rng)
    x = rng.standard_normal((B, L, U)) + offset * rng.choice([-1.0, 1.0], size=(B, L, 1))
    vl = np.array([L, 90])
    xd, vd = to_dev(x, torch.bfloat16), torch.tensor(vl, device='cuda')
    got = m(xd, vd)[0]
    assert _lib.load().gnlp_stack_last_folded() == 1
    plain = _plain(m, xd, vd)
    exp, _ = ref.bert_transformer(prep(x, torch.bfloat16), vl,
                                  [oracle_params(lp, torch.bfloat16) for lp in layers], H)
    e_fold, e_plain = rel_err(to_np(got), exp), rel_err(to_np(plain), exp)
    print('[fold] mean = %g sigma: folded %.3e, plain %.3e against the oracle' % (offset, e_fold, e_plain))
    assert e_fold < e_plain + 5e-3
    if offset <= 3.0:
        assert e_fold < np.sqrt(NL) * 2e-2


def test_folded_roberta_stack_without_final_mask(nlp, tiles256):
    """RobertaEncoder: no final sequence mask, eps 1e-5, padded rows keep their (non-zero) values:
    the folded run against the plain run on every row, and against the oracle."""
    from gluonnlp_amd.models.roberta import RobertaEncoder
    rng = np.random.default_rng(12)
    B, L, U, F, H, NL = 2, 128, 256, 1024, 4, 2
    m = RobertaEncoder(units=U, hidden_size=F, num_layers=NL, num_heads=H, layer_norm_eps=1e-5,
                       dtype='bfloat16').cuda()
    layers = [ref.make_encoder_layer_params(rng, U, F) for _ in range(NL)]
    for i, lp in enumerate(layers):
        load_params(m.all_layers[i], lp, torch.bfloat16)
    x = rng.standard_normal((B, L, U))
    vl = np.array([L, 50])
    xd, vd = to_dev(x, torch.bfloat16), torch.tensor(vl, device='cuda')
    got = m(xd, vd)[0]
    assert _lib.load().gnlp_stack_last_folded() == 1
    plain = _plain(m, xd, vd)
    exp, _ = ref.bert_transformer(prep(x, torch.bfloat16), vl,
                                  [oracle_params(lp, torch.bfloat16) for lp in layers], H,
                                  layer_norm_eps=1e-5, final_sequence_mask=False)
    assert rel_err(to_np(got), exp) < np.sqrt(NL) * 2e-2
    assert rel_err(to_np(got), to_np(plain).astype(np.float64)) < 2e-2


def test_folded_stack_packed_ragged_is_bit_identical(nlp):
    """Folded + packed to the valid tokens == folded + padded, bit for bit (every folded operation
    is local to a row), zeros beyond valid_length."""
    torch.manual_seed(8)
    from gluonnlp_amd.models import BertTransformer
    B, L, U, H, F = 128, 256, 256, 4, 1024   # (128 tiles per GEMM: the dispatcher's own choice)
    m = BertTransformer(units=U, hidden_size=F, num_layers=3, num_heads=H, dtype='bfloat16').cuda()
    for p in m.parameters():
        p.data.normal_(0, 0.05)
    for ly in m.all_layers:
        ly.layer_norm.gamma.data.add_(1.0)
        ly.ffn.layer_norm.gamma.data.add_(1.0)
    gen = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn((B, L, U), device='cuda', generator=gen).to(torch.bfloat16)
    vl = torch.randint(1, L + 1, (B,), device='cuda', generator=gen)
    vl[0], vl[1], vl[2] = L, 1, 0
    try:
        _lib.set_knob('GNLP_PACK_RAGGED', '0')
        want = m(x, vl)[0]
        assert _lib.load().gnlp_stack_last_packed() == 0 and _lib.load().gnlp_stack_last_folded() == 1
    finally:
        _lib.set_knob('GNLP_PACK_RAGGED', None)
    got = m(x, vl)[0]
    assert _lib.load().gnlp_stack_last_packed() == 1 and _lib.load().gnlp_stack_last_folded() == 1
    assert torch.equal(got, want)
    live = torch.arange(L, device='cuda')[None, :] < vl[:, None]
    assert float(got[~live].float().abs().max()) == 0.0


def test_folded_parameters_follow_the_parameters(nlp, tiles256):
    """The folded weights are cached per parameter version: an in-place update through the
    parameter (what an optimizer step or load_state_dict does) and a re-seated `.data` are seen
    at the next forward; an in-place write through `.data` is not (documented) until
    drop_folded_parameters; shapes the folded kernels do not take fall back to the plain stack."""
    from gluonnlp_amd.models.transformer import drop_folded_parameters
    rng = np.random.default_rng(5)
    B, L, U, F, H, NL = 2, 128, 256, 1024, 4, 2
    m, _ = _bert(U, F, H, NL, rng)
    x = to_dev(rng.standard_normal((B, L, U)), torch.bfloat16)
    vl = torch.tensor([L, 64], device='cuda')
    y0 = m(x, vl)[0]
    ly = m.all_layers[1]
    with torch.no_grad():
        ly.layer_norm.gamma.mul_(1.5)                     # bumps the version counter
    y1 = m(x, vl)[0]
    assert _lib.load().gnlp_stack_last_folded() == 1
    assert rel_err(to_np(y1), to_np(_plain(m, x, vl)).astype(np.float64)) < 2e-2
    assert rel_err(to_np(y1), to_np(y0).astype(np.float64)) > 5e-2
    ly.ffn.ffn_1.weight.data = ly.ffn.ffn_1.weight.data * 0.5      # re-seated storage
    y2 = m(x, vl)[0]
    assert rel_err(to_np(y2), to_np(_plain(m, x, vl)).astype(np.float64)) < 2e-2
    ly.attn_qkv.weight.data.mul_(0.25)                    # in place through .data: NOT seen ...
    drop_folded_parameters(m)                             # ... until the cache is dropped
    y3 = m(x, vl)[0]
    assert rel_err(to_np(y3), to_np(_plain(m, x, vl)).astype(np.float64)) < 2e-2
    # rows not a multiple of 256: the plain path runs
    x2 = to_dev(rng.standard_normal((3, 100, U)), torch.bfloat16)
    m(x2, torch.tensor([100, 3, 50], device='cuda'))
    assert _lib.load().gnlp_stack_last_folded() == 0


# ---- T5: RMSNorm folded into the GEMMs that consume it (gnlp_t5_encoder_forward_folded) -----------
def _t5(D, DKV, DFF, H, NL, rng, beta=False):
    from gluonnlp_amd.models import T5Encoder
    lps = [ref.make_t5_block_params(rng, D, DKV, DFF, H, False) for _ in range(NL)]
    if beta:   # (the reference's RMSNorm carries a beta too: b' = beta . W^T of the fold)
        for lp in lps:
            lp['self_attn_layer_norm.beta'] = (0.1 * rng.standard_normal(D)).astype(np.float32)
            lp['ffn.layer_norm.beta'] = (0.1 * rng.standard_normal(D)).astype(np.float32)
    relw = (rng.standard_normal((32, H)) * D ** -0.5).astype(np.float32)
    fin = {'gamma': (1 + 0.1 * rng.standard_normal(D)).astype(np.float32),
           'beta': np.zeros(D, np.float32)}
    enc = T5Encoder(D, DKV, DFF, num_layers=NL, num_heads=H, activation='relu', dtype='bfloat16')
    for i in range(NL):
        load_params(enc.layers[i], lps[i], torch.bfloat16)
    load_params(enc.final_layer_norm, fin, torch.bfloat16)
    load_params(enc.relative_position_encoder, {'_rel_pos_embed.weight': relw}, torch.bfloat16)
    return enc.cuda(), lps, relw, fin


@pytest.mark.parametrize('D,DKV,DFF,H,NL,B,L,beta', [(256, 64, 1024, 4, 2, 2, 128, False),
                                                    (768, 64, 3072, 12, 3, 2, 256, True)])
def test_folded_t5_encoder_against_oracle_and_plain(nlp, tiles256, D, DKV, DFF, H, NL, B, L, beta):
    """T5Encoder (pre-norm RMSNorm blocks, relu FFN, relative-position bias, ragged lengths) with
    the norms folded: against the float64 oracle at the stack tolerance and against the plain run
    (GNLP_LN_FOLD=0); 192-wide GEMM tiles at d_model = 768."""
    rng = np.random.default_rng(D + L)
    enc, lps, relw, fin = _t5(D, DKV, DFF, H, NL, rng, beta)
    x = rng.standard_normal((B, L, D))
    vl = np.array([L, L // 3])
    xd, vd = to_dev(x, torch.bfloat16), torch.tensor(vl, device='cuda')
    got = enc(xd, vd)
    assert _lib.load().gnlp_stack_last_folded() == 1
    try:
        _lib.set_knob('GNLP_LN_FOLD', '0')
        plain = enc(xd, vd)
        assert _lib.load().gnlp_stack_last_folded() == 0
    finally:
        _lib.set_knob('GNLP_LN_FOLD', None)
    o = lambda a: prep(a, torch.bfloat16)
    want = ref.t5_encoder(o(x), vl, o(relw), [{k: o(v) for k, v in p.items()} for p in lps],
                          {k: o(v) for k, v in fin.items()}, H, activation_name='relu')
    worst = [0.0, 0.0]
    for b in range(B):
        n = vl[b]
        worst[0] = max(worst[0], float(rel_err(to_np(got)[b, :n], want[b, :n])))
        worst[1] = max(worst[1], float(rel_err(to_np(plain)[b, :n], want[b, :n])))
    print('[fold] T5 D=%d: folded %.3e, plain %.3e against the oracle' % (D, worst[0], worst[1]))
    assert worst[0] < np.sqrt(NL + 1) * 2e-2
    assert worst[0] < worst[1] + 5e-3


def test_folded_t5_parameters_follow_updates(nlp, tiles256):
    """The folded tensors are rebuilt when a source parameter changes version (copy_), kept
    otherwise; drop_folded_parameters forgets them."""
    from gluonnlp_amd.models.transformer import drop_folded_parameters
    rng = np.random.default_rng(4)
    enc, _, _, _ = _t5(256, 64, 1024, 4, 1, rng)
    xd = to_dev(rng.standard_normal((2, 128, 256)), torch.bfloat16)
    vd = torch.tensor([128, 128], device='cuda')
    a = enc(xd, vd)
    assert torch.equal(a, enc(xd, vd))
    with torch.no_grad():
        enc.layers[0].ffn.layer_norm.gamma.copy_(enc.layers[0].ffn.layer_norm.gamma * 1.5)
    b = enc(xd, vd)
    assert not torch.equal(a, b)
    try:
        _lib.set_knob('GNLP_LN_FOLD', '0')
        plain = enc(xd, vd)
    finally:
        _lib.set_knob('GNLP_LN_FOLD', None)
    assert rel_err(to_np(b), to_np(plain)) < 2e-2
    drop_folded_parameters(enc)
    assert torch.equal(b, enc(xd, vd))


def test_stack_with_tanh_gelu_runs_unfolded(nlp, tiles256):
    """Found by tests/dev_stack_fuzz.py (seed 31, case 89): the persistent kernel has no FOLD 1
    instantiation for the tanh GELU, and the stack driver used to promise the fold anyway (a loud
    "unsupported Dense shape").  Such a stack now runs the plain path."""
    from gluonnlp_amd.models import TransformerEncoder
    rng = np.random.default_rng(89)
    B, L, U, F, H = 1, 256, 256, 512, 8
    lp = [ref.make_encoder_layer_params(rng, U, F)]
    m = TransformerEncoder(num_layers=1, units=U, hidden_size=F, num_heads=H, pre_norm=False,
                           layer_norm_eps=1e-12, activation='gelu(tanh)', dtype='bfloat16')
    load_params(m.layers[0], lp[0], torch.bfloat16)
    m.cuda()
    x = rng.standard_normal((B, L, U))
    vl = np.array([200])
    got = m(to_dev(x, torch.bfloat16), torch.tensor(vl, device='cuda'))
    assert _lib.load().gnlp_stack_last_folded() == 0
    exp = ref.transformer_encoder(prep(x, torch.bfloat16), vl,
                                  [oracle_params(lp[0], torch.bfloat16)], H, 1e-12, False,
                                  'gelu(tanh)', 'NT')
    assert rel_err(to_np(got)[0, :200], exp[0, :200]) < 2e-2
