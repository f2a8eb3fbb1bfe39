"""RelAttentionScoreCell 'transformer_xl' / 'shaw', Sinusoidal / Learned positional embeddings on
the MI355X (forward; attention_cell.py:659-881, layers.py:370-482) through the
C ABI, against the oracle; the reference test's properties (tests/test_attention_cell.py:272-391)
are checked on the device results themselves."""
import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import prep, rel_err, to_dev, to_np, tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    from gluonnlp_amd import _lib
    _lib.load()
    return gluonnlp_amd


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('units', [16, 33, 64])
def test_sinusoidal_embedding(nlp, dtype, units):
    from gluonnlp_amd.layers import PositionalEmbedding
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    m = PositionalEmbedding(units=units, method='sinusoidal', dtype=dn).cuda()
    pos = np.concatenate([np.arange(-40, 600), [2047, -2047]]).reshape(2, -1)
    got = m(torch.from_numpy(pos).cuda())
    assert tuple(got.shape) == pos.shape + (units,) and got.dtype == dtype
    want = ref.sinusoidal_positional_embedding(pos, units)
    # device sinf / cosf on the same float32 argument: a few ulp of 1.0
    assert np.abs(to_np(got) - want).max() < (2e-6 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize('mode', ['clip', 'wrap', 'raise'])
def test_learned_positional_embedding_modes(nlp, mode):
    from gluonnlp_amd.layers import LearnedPositionalEmbedding
    rng = np.random.default_rng(3)
    m = LearnedPositionalEmbedding(units=24, max_length=11, mode=mode).cuda()
    w = rng.standard_normal((11, 24)).astype(np.float32)
    m.weight.data = to_dev(w)
    pos = np.array([[0, 10, 5], [-1, -11, 3]]) if mode != 'clip' else np.array([[0, 10, 50], [-1, -11, 3]])
    if mode == 'wrap':
        pos = np.array([[0, 10, 50], [-1, -23, 3]])
    got = m(torch.from_numpy(pos).cuda())
    assert torch.equal(got.cpu(), torch.from_numpy(ref.learned_positional_embedding(w, pos, mode)))
    if mode == 'raise':
        with pytest.raises(IndexError):
            m(torch.tensor([11]).cuda())


CASES = [('transformer_xl', False, None), ('transformer_xl', True, 6), ('shaw', False, 20),
         ('shaw', True, 5)]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NKT', 'NTK', 'TNK'])
@pytest.mark.parametrize('method,bidirectional,max_distance', CASES)
def test_rel_attention_score(nlp, method, bidirectional, max_distance, layout, dtype):
    from gluonnlp_amd.attention_cell import RelAttentionScoreCell
    rng = np.random.default_rng(9)
    B, H, Lq, Lm, C = 3, 4, 20, 37, 16          # shapes of tests/test_attention_cell.py:276-281 scaled
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    cell = RelAttentionScoreCell(query_units=H * C, num_heads=H, method=method,
                                 bidirectional=bidirectional, max_distance=max_distance,
                                 layout=layout, dtype=dn).cuda()
    kw = dict(method=method, num_heads=H, max_distance=max_distance, bidirectional=bidirectional)
    if method == 'transformer_xl':
        w = (0.3 * rng.standard_normal((H * C, H * C))).astype(np.float32)
        cell._rel_proj.weight.data = to_dev(w, dtype)
        kw['rel_proj_weight'] = prep(w, dtype)
    else:
        vocab = (2 * max_distance + 1) if bidirectional else (max_distance + 1)
        w = rng.standard_normal((vocab, H * C)).astype(np.float32)
        cell._rel_pos_embed.weight.data = to_dev(w, dtype)
        kw['shaw_weight'] = prep(w, dtype)
    q = rng.standard_normal((B, H, Lq, C))
    rel = np.arange(Lq)[:, None] - np.arange(Lm)[None, :]
    perm = {'NKT': (0, 1, 2, 3), 'NTK': (0, 2, 1, 3), 'TNK': (2, 0, 1, 3)}[layout]
    qd = to_dev(q, dtype).permute(*perm).contiguous()
    reld = torch.from_numpy(rel.astype(np.int32)).cuda()
    got = cell(reld, qd)
    assert tuple(got.shape) == (B, H, Lq, Lm) and got.dtype == dtype
    want = ref.rel_attention_score(prep(q, dtype), rel, layout='NKT', **kw)
    # transformer_xl in bf16: the sinusoid table and its projection are rounded to bf16 on the
    # device (as the reference's dtype=bf16 cell would), not in the float64 oracle
    t = tol(dtype) * (2 if (dtype == torch.bfloat16 and method == 'transformer_xl') else 1)
    assert rel_err(to_np(got), want) < t
    # permutation equivariance on the device (tests/test_attention_cell.py:332-355)
    qp = torch.from_numpy(rng.permutation(Lq)).cuda()
    mp = torch.from_numpy(rng.permutation(Lm)).cuda()
    q_nkt = to_dev(q, dtype)[:, :, qp, :]
    got_p = cell(reld[qp][:, mp].contiguous(), q_nkt.permute(*perm).contiguous())
    assert torch.equal(got_p, got[:, :, qp][:, :, :, mp])
    # strided query (a view of a packed buffer) gives the same bits: layouts are strides
    packed = torch.zeros(qd.shape[:-1] + (3 * C,), dtype=dtype, device='cuda')
    packed[..., C:2 * C] = qd
    assert torch.equal(cell(reld, packed[..., C:2 * C]), got)


def test_rel_attention_score_errors(nlp):
    from gluonnlp_amd.attention_cell import RelAttentionScoreCell
    with pytest.raises(AssertionError):
        RelAttentionScoreCell(query_units=16, num_heads=4, method='shaw')      # needs max_distance
    with pytest.raises(NotImplementedError):
        RelAttentionScoreCell(query_units=16, num_heads=4, method='nope')
    with pytest.raises(ValueError):
        RelAttentionScoreCell(query_units=16, num_heads=4, layout='KNT')
    cell = RelAttentionScoreCell(query_units=16, num_heads=4, method='transformer_xl').cuda()
    with pytest.raises(AssertionError):
        cell(torch.zeros((2, 2), dtype=torch.int32, device='cuda'))           # query is required


# ------------------------------------------------------------------------- backward pass
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NKT', 'NTK', 'TNK'])
@pytest.mark.parametrize('method,bidirectional,max_distance', CASES)
def test_rel_attention_score_backward(nlp, method, bidirectional, max_distance, layout, dtype):
    """tests/test_attention_cell.py:313-391 back-propagates a random cotangent through the cell and
    compares the gradient norms of the query and of the cell's parameters across layouts; here the
    gradients themselves are compared with the oracle (finite-difference pinned, parity otherwise
    unpinned): d query in the cell's layout, d(shaw table) / d(rel_proj.weight)."""
    from gluonnlp_amd.attention_cell import RelAttentionScoreCell
    rng = np.random.default_rng(13)
    B, H, Lq, Lm, C = 2, 4, 20, 37, 16
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    cell = RelAttentionScoreCell(query_units=H * C, num_heads=H, method=method,
                                 bidirectional=bidirectional, max_distance=max_distance,
                                 layout=layout, dtype=dn).cuda()
    kw = dict(method=method, num_heads=H, max_distance=max_distance, bidirectional=bidirectional)
    if method == 'transformer_xl':
        w = (0.3 * rng.standard_normal((H * C, H * C))).astype(np.float32)
        param, pname = cell._rel_proj.weight, 'rel_proj_weight'
    else:
        vocab = (2 * max_distance + 1) if bidirectional else (max_distance + 1)
        w = rng.standard_normal((vocab, H * C)).astype(np.float32)
        param, pname = cell._rel_pos_embed.weight, 'shaw_weight'
    param.data = to_dev(w, dtype)
    param.requires_grad_(True)
    kw[pname] = prep(w, dtype)
    q = rng.standard_normal((B, H, Lq, C))
    g = rng.standard_normal((B, H, Lq, Lm))
    rel = np.arange(Lm)[None, :] - np.arange(Lq)[:, None]
    if method == 'shaw' and not bidirectional:
        rel = np.abs(rel)
    perm = {'NKT': (0, 1, 2, 3), 'NTK': (0, 2, 1, 3), 'TNK': (2, 0, 1, 3)}[layout]
    qd = to_dev(q, dtype).permute(*perm).contiguous().requires_grad_(True)
    out = cell(torch.from_numpy(rel.astype(np.int32)).cuda(), qd)
    out.backward(to_dev(g, dtype))
    dq, dp = ref.rel_attention_score_backward(prep(g, dtype), prep(q, dtype), rel, layout='NKT',
                                              **kw)
    inv = tuple(np.argsort(perm))
    t = tol(dtype) * (2 if (dtype == torch.bfloat16 and method == 'transformer_xl') else 1)
    assert qd.grad.shape == qd.shape and qd.grad.dtype == dtype
    assert rel_err(np.transpose(to_np(qd.grad), inv), dq) < t
    assert param.grad is not None and param.grad.shape == param.shape
    assert rel_err(to_np(param.grad), dp) < t
    # rows of the learned table no position maps to get no gradient
    if method == 'shaw' and not bidirectional:
        assert float(param.grad[:1].abs().sum()) > 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('bidirectional', [True, False])
def test_t5_rel_score_backward(nlp, bidirectional, dtype):
    """method='t5': the bias is a table lookup, its gradient a scatter onto (num_buckets, H)."""
    from gluonnlp_amd.attention_cell import RelAttentionScoreCell
    rng = np.random.default_rng(2)
    H, Lq, Lm, nb, md = 4, 33, 70, 32, 40
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    cell = RelAttentionScoreCell(query_units=H * 8, num_heads=H, method='t5', num_buckets=nb,
                                 max_distance=md, bidirectional=bidirectional, dtype=dn).cuda()
    w = rng.standard_normal((nb, H))
    cell._rel_pos_embed.weight.data = to_dev(w, dtype)
    cell._rel_pos_embed.weight.requires_grad_(True)
    rel = (np.arange(Lm)[None, :] - np.arange(Lq)[:, None]).astype(np.int32)
    g = rng.standard_normal((H, Lq, Lm))
    out = cell(torch.from_numpy(rel).cuda())
    assert tuple(out.shape) == (H, Lq, Lm)
    out.backward(to_dev(g, dtype))
    bucket = ref.relative_position_bucket(rel, bidirectional, nb, md)
    want = np.zeros((nb, H))
    np.add.at(want, bucket, np.transpose(prep(g, dtype), (1, 2, 0)))
    assert rel_err(to_np(cell._rel_pos_embed.weight.grad), want) < tol(dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [(37, 24, 40), (512, 256, 128)])
def test_dense_backward(nlp, dtype, M, N, K):
    """Plain Dense under autograd: dx = g W, dW = g^T x, db = column sums (gnlp_dense on transposed
    operands + gnlp_colsum)."""
    from gluonnlp_amd.layers import Dense
    rng = np.random.default_rng(M + N)
    dn = 'float32' if dtype == torch.float32 else 'bfloat16'
    d = Dense(N, K, dtype=dn).cuda()
    w, b = 0.2 * rng.standard_normal((N, K)), rng.standard_normal(N)
    x, g = rng.standard_normal((3, M, K)), rng.standard_normal((3, M, N))
    d.weight.data, d.bias.data = to_dev(w, dtype), to_dev(b, dtype)
    d.weight.requires_grad_(True)
    d.bias.requires_grad_(True)
    xd = to_dev(x, dtype).requires_grad_(True)
    y = d(xd)
    y.backward(to_dev(g, dtype))
    px, pw, pg = prep(x, dtype), prep(w, dtype), prep(g, dtype)
    t = tol(dtype)
    assert rel_err(to_np(y), px @ pw.T + prep(b, dtype)) < t
    assert rel_err(to_np(xd.grad), pg @ pw) < t
    assert rel_err(to_np(d.weight.grad), pg.reshape(-1, N).T @ px.reshape(-1, K)) < t
    assert rel_err(to_np(d.bias.grad), pg.reshape(-1, N).sum(0)) < t
    # fused epilogues record their backward too since round 3 (tests/test_gpu_backward_layers.py);
    # writing into a caller's buffer cannot be recorded
    assert d(xd, act='gelu').requires_grad
    with pytest.raises(ValueError):
        d(xd, out=torch.empty_like(y))
    with torch.no_grad():
        assert not d(xd, act='gelu').requires_grad
