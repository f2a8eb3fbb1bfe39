"""Parity at BASELINE.json's FULL sizes, through the C ABI on one MI355X.

The oracle cannot run a whole full-size batch in seconds, so each configuration is checked through
size-independent properties plus a direct comparison of a few whole sequences:
  * sequences are independent (the mask is per sequence, LayerNorm per token): the oracle encodes
    2-3 sequences of the batch alone -- the full depth of the stack at the full L -- and the
    device's rows for those sequences must match it within the stated tolerance;
  * permuting the sequences of the batch permutes the output rows BIT FOR BIT (every row is
    computed with the same K order whatever tile / workgroup it lands in);
  * padded positions are exactly zero where the reference applies sequence_mask (BERT / ELECTRA).
Tolerances: bf16 2e-2, fp32 1e-5 (global AND element-wise, tests/util.py rel_err) PER LAYER, as
SURVEY 8d states them ("within 1e-5 rel of the float64 oracle per layer"): every layer of the
full-size stack is checked by feeding the oracle layer the device's own previous output (so
rounding does not accumulate in the comparison).  End to end against the oracle run at full depth:
fp32 at the SAME 1e-5 (no allowance); bf16 within sqrt(layers) x 2e-2 -- 24 bf16 hand-offs between
layers each round to 8 bits, measured 2.3e-2 at 24 layers, over the per-layer figure, and stated
as such in DESIGN.md section 4.  Every measured end-to-end error is printed (pytest -s / -rP)."""
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


def _ragged(rng, B, L):
    vl = rng.integers(1, L + 1, size=B)
    vl[0], vl[-1] = L, 1          # at least one full and one length-1 sequence
    return vl


def _stack_case(kind, dtype, U, F, H, NL, B, L, eps, seed, pick, vl=None):
    from gluonnlp_amd.models import BertTransformer, ElectraEncoder, RobertaEncoder
    rng = np.random.default_rng(seed)
    cls = {'bert': BertTransformer, 'electra': ElectraEncoder, 'roberta': RobertaEncoder}[kind]
    attr = 'all_encoder_layers' if kind == 'electra' else 'all_layers'
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    m = cls(units=U, hidden_size=F, num_layers=NL, num_heads=H, layer_norm_eps=eps, dtype=dn,
            output_all_encodings=True)
    lps = [ref.make_encoder_layer_params(rng, U, F) for _ in range(NL)]
    for layer, lp in zip(getattr(m, attr), lps):
        load_params(layer, lp, dtype)
    m.cuda()
    x = rng.standard_normal((B, L, U)).astype(np.float32)
    vl = _ragged(rng, B, L) if vl is None else np.asarray(vl)
    xd = to_dev(x, dtype)
    vld = torch.from_numpy(vl).cuda()
    outs, _ = m(xd, vld)
    outs = list(outs)
    if kind == 'roberta':
        outs = outs[1:]           # roberta.py:175: the input is the first entry of the list
    assert len(outs) == NL
    out = outs[-1]
    got = to_np(out)
    assert np.isfinite(got).all()
    olps = [{k: prep(v, dtype) for k, v in lp.items()} for lp in lps]
    t = tol(dtype)
    # (1a) every layer at full size: oracle layer on the device's own previous output
    mask = ref.gen_self_attn_mask((len(pick), L), vl[pick])
    prev = prep(x[pick], dtype)
    worst_layer = 0.0
    for i in range(NL):
        want_i, _ = ref.transformer_encoder_layer(prev, mask, olps[i], H, eps, False, 'gelu')
        got_i = to_np(outs[i][torch.tensor(pick, device='cuda')])
        for j, b in enumerate(pick):
            n = vl[b]
            e = rel_err(got_i[j, :n], want_i[j, :n])
            worst_layer = max(worst_layer, float(e))
            assert e < t, (kind, 'layer', i, 'seq', b)
        prev = got_i
    # (1b) whole sequences against the oracle at full depth and full L: fp32 at the per-layer
    # tolerance itself; bf16 with the sqrt(layers) allowance for the NL roundings between layers
    enc = ref.roberta_encoder if kind == 'roberta' else ref.bert_transformer
    want, _ = enc(prep(x[pick], dtype), vl[pick], olps, H, eps, 'gelu')
    t_e2e = t if dtype == torch.float32 else t * NL ** 0.5
    worst_e2e = 0.0
    for j, b in enumerate(pick):
        n = vl[b]
        e = rel_err(got[b, :n], want[j, :n])
        worst_e2e = max(worst_e2e, float(e))
        assert e < t_e2e, (kind, b, n)
    print('[fullsize] {} {} U={} L={} B={} layers={}: worst per-layer err {:.3e} (tol {:.0e}), '
          'end-to-end err {:.3e} (tol {:.2e})'.format(kind, str(dtype).split('.')[-1], U, L, B, NL,
                                                      worst_layer, t, worst_e2e, t_e2e))
    # (2) permutation of the batch = permutation of the rows, bit for bit
    perm = torch.from_numpy(rng.permutation(B)).cuda()
    outs_p, _ = m(xd[perm].contiguous(), vld[perm].contiguous())
    assert torch.equal(list(outs_p)[-1], out[perm])
    # (3) sequence_mask: exact zeros beyond valid_length (bert.py:311-312); RoBERTa has none
    if kind != 'roberta':
        keep = torch.arange(L, device='cuda')[None, :] < vld[:, None]
        assert (out[~keep] == 0).all()
    return got


@pytest.mark.parametrize('lengths', ['full', 'ragged'])
def test_cfg1_bert_base_fp32_full_size(nlp, lengths):
    """Config 1 (BASELINE.json configs[0]): BERT-base fp32, B=8, L=128, the WHOLE
    batch against the float64 oracle, valid_length full (as the reference harness runs it) and
    default_rng(0).integers(1, L+1); 1e-5 per layer and end to end, no allowance."""
    B, L = 8, 128
    vl = np.full(B, L) if lengths == 'full' else np.random.default_rng(0).integers(1, L + 1, B)
    _stack_case('bert', torch.float32, 768, 3072, 12, 12, B, L, 1e-12, 300, list(range(B)), vl=vl)


def test_cfg3_bert_large_full_size(nlp):
    """BERT-large, 24 layers, B=64, L=512, bf16 (the headline workload, ragged lengths)."""
    _stack_case('bert', torch.bfloat16, 1024, 4096, 16, 24, 64, 512, 1e-12, 301, [0, 63, 17])


def test_cfg2_bert_base_ragged_full_size(nlp):
    """BERT-base, 12 layers, B=256, L=128, bf16, ragged valid_length."""
    _stack_case('bert', torch.bfloat16, 768, 3072, 12, 12, 256, 128, 1e-12, 302, [0, 255, 100])


@pytest.mark.parametrize('kind,eps', [('electra', 1e-12), ('roberta', 1e-5)])
def test_cfg5_large_fp32_and_bf16_full_size(nlp, kind, eps):
    """ELECTRA-large / RoBERTa-large, 24 layers, B=32, L=512: the fp32 tolerance path and bf16."""
    _stack_case(kind, torch.float32, 1024, 4096, 16, 24, 32, 512, eps, 303, [0, 31])
    _stack_case(kind, torch.bfloat16, 1024, 4096, 16, 24, 32, 512, eps, 304, [0, 31])


def test_cfg4_t5_base_full_size(nlp):
    """T5-base encoder, 12 blocks, B=16, L=1024, bf16, relative-position bias: buckets bit-exact,
    every block against the oracle block at 2e-2, whole sequences against the oracle at full depth,
    batch permutation bit for bit."""
    from gluonnlp_amd.models import T5Encoder
    ac = nlp.attention_cell
    rng = np.random.default_rng(305)
    D, DKV, DFF, H, NL, B, L = 768, 64, 3072, 12, 12, 16, 1024
    dtype = torch.bfloat16
    lps = [ref.make_t5_block_params(rng, D, DKV, DFF, H, False) for _ in range(NL)]
    relw = (rng.standard_normal((32, H)) * D ** -0.5).astype(np.float32)
    fin = {'gamma': (1 + 0.1 * rng.standard_normal(D)).astype(np.float32),
           'beta': np.zeros(D, np.float32)}
    enc = T5Encoder(D, DKV, DFF, num_layers=NL, num_heads=H, activation='relu', dtype='bfloat16')
    for i in range(NL):
        load_params(enc.layers[i], lps[i], dtype)
    load_params(enc.final_layer_norm, fin, dtype)
    load_params(enc.relative_position_encoder, {'_rel_pos_embed.weight': relw}, dtype)
    enc.cuda()
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    vl = _ragged(rng, B, L)
    xd, vld = to_dev(x, dtype), torch.from_numpy(vl).cuda()
    out = enc(xd, vld)
    got = to_np(out)
    # relative-position buckets at L = 1024: bit-exact against the C restatement of op.py:229-286
    rel = ac.gen_rel_position(xd)
    bk = nlp.op.relative_position_bucket(rel, bidirectional=True, num_buckets=32,
                                         max_distance=128)
    assert np.array_equal(bk.cpu().numpy(),
                          ref.relative_position_bucket(ref.gen_rel_position((B, L)), True, 32, 128))
    pick = [0, B - 1]
    o = lambda a: prep(a, dtype)
    # (1a) every block at full size, as _stack_case does for the BERT-shaped stacks: block i alone
    # through the same C entry (one block, no final norm) on the device's output of block i - 1,
    # the oracle block on that same input -- rounding does not accumulate in the comparison, so
    # each block is held to the plain 2e-2 (the end-to-end check below carries sqrt(layers))
    from gluonnlp_amd.models.t5 import run_t5_blocks
    table = enc.relative_position_encoder.self_attn_bias_table(L)
    pos = ref.rel_attention_score_t5(o(relw), ref.gen_rel_position((len(pick), L)),
                                     bidirectional=True, num_buckets=32, max_distance=128)
    mask = ref.gen_self_attn_mask((len(pick), L), vl[pick])
    prev_d, prev = xd, o(x[pick])
    worst_block, worst_at = 0.0, -1
    idx = torch.tensor(pick, device='cuda')
    for i in range(NL):
        prev_d = run_t5_blocks([enc.layers[i]], prev_d, vld, table, None, 'NT', toeplitz=True)
        want_i = ref.t5_block(prev, mask, pos, {k: o(v) for k, v in lps[i].items()}, H,
                              activation_name='relu')
        got_i = to_np(prev_d[idx])
        for j, b in enumerate(pick):
            n = vl[b]
            e = float(rel_err(got_i[j, :n], want_i[j, :n]))
            if e > worst_block:
                worst_block, worst_at = e, i
            assert e < tol(dtype), ('t5 block', i, 'seq', b)
        prev = got_i
    print('[fullsize] t5_base bf16 L=1024 B=16: worst per-block err {:.3e} at block {} (tol {:.0e})'
          .format(worst_block, worst_at, tol(dtype)))
    # the chain of single-block calls is the stack without its final norm, on the same kernels; not
    # bit for bit since the RMSNorm fold (a block called alone takes its input's statistics from a
    # pass over the tensor, inside the stack they come from the producing GEMM's partial sums)
    fin_d = enc.final_layer_norm(prev_d)
    keep_all = torch.arange(L, device='cuda')[None, :] < vld[:, None]
    assert rel_err(to_np(fin_d[keep_all]), to_np(out[keep_all])) < tol(dtype) * NL ** 0.5   # (12 blocks of roundings apart)
    want = ref.t5_encoder(o(x[pick]), vl[pick], o(relw),
                          [{k: o(v) for k, v in p.items()} for p in lps],
                          {k: o(v) for k, v in fin.items()}, H, activation_name='relu')
    worst = 0.0
    for j, b in enumerate(pick):
        n = vl[b]
        e = rel_err(got[b, :n], want[j, :n])
        worst = max(worst, float(e))
        assert e < tol(dtype) * NL ** 0.5, (b, n)
    print('[fullsize] t5_base bf16 L=1024 B=16 layers=12: end-to-end err {:.3e} (tol {:.2e})'
          .format(worst, tol(dtype) * NL ** 0.5))
    perm = torch.from_numpy(rng.permutation(B)).cuda()
    out_p = enc(xd[perm].contiguous(), vld[perm].contiguous())
    keep = (torch.arange(L, device='cuda')[None, :] < vld[:, None])[perm]
    assert torch.equal(out_p[keep], out[perm][keep])
