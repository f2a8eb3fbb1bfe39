"""GPU parity tests (through the C ABI) for the integer / HBM-bound kernels and the dense layer:
masks, sequence_mask, softmax, l2_normalize, activations, norms, buckets, Dense.
Bit-exact for integer / byte work; fp32 within 1e-5, bf16 within 2e-2 (max-abs relative)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import (BF16_TOL, FP32_TOL, prep, rel_err, to_dev, to_np, tol)

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    from gluonnlp_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return gluonnlp_amd


# ----------------------------------------------------------------------------------- masks
@pytest.mark.parametrize('B,L', [(4, 8), (3, 77), (2, 128), (1, 1), (5, 33)])
@pytest.mark.parametrize('vl_dtype', [torch.int32, torch.int64])
def test_self_mask_bit_exact(nlp, B, L, vl_dtype):
    ac = nlp.attention_cell
    rng = np.random.default_rng(B * 100 + L)
    vl = rng.integers(1, L + 1, size=B)
    vl[0] = L
    vl[-1] = 1
    vlt = torch.tensor(vl, dtype=vl_dtype, device='cuda')
    for layout in ['NT', 'TN']:
        data = torch.zeros((B, L, 4) if layout == 'NT' else (L, B, 4), device='cuda')
        for attn_type in ['full', 'causal']:
            m = ac.gen_self_attn_mask(data, vlt, attn_type=attn_type, layout=layout)
            assert m.dtype == torch.bool and tuple(m.shape) == (B, L, L)
            exp = ref.gen_self_attn_mask((B, L), vl, attn_type=attn_type)
            assert np.array_equal(m.cpu().numpy(), exp)
            m = ac.gen_self_attn_mask(data, None, attn_type=attn_type, layout=layout)
            assert np.array_equal(m.cpu().numpy(),
                                  ref.gen_self_attn_mask((B, L), None, attn_type=attn_type))
    with pytest.raises(NotImplementedError):
        ac.gen_self_attn_mask(data, vlt, layout='NTC')
    with pytest.raises(NotImplementedError):
        ac.gen_self_attn_mask(data, vlt, attn_type='sliding')


def test_mask_goldens(nlp):
    g = np.load(os.path.join(G, 'mask_self.npz'))
    ac = nlp.attention_cell
    for tag, B, L in [('t', 4, 8), ('doc', 2, 8)]:
        vl = torch.tensor(g[tag + '_vl'], device='cuda')
        for layout in ['NT', 'TN']:
            data = torch.zeros((B, L, 5) if layout == 'NT' else (L, B, 5), device='cuda')
            for attn_type in ['full', 'causal']:
                m = ac.gen_self_attn_mask(data, vl, attn_type=attn_type, layout=layout)
                assert np.array_equal(m.cpu().numpy(), g['{}_{}_{}'.format(tag, attn_type, layout)])


def test_mem_mask_and_sequence_mask(nlp):
    ac = nlp.attention_cell
    rng = np.random.default_rng(4)
    B, Lq, Lm = 4, 8, 6
    vl = rng.integers(1, Lq + 1, size=B)
    mvl = rng.integers(1, Lm + 1, size=B)
    mem = torch.zeros((B, Lm, 3), device='cuda')
    data = torch.zeros((B, Lq, 3), device='cuda')
    m = ac.gen_mem_attn_mask(mem, torch.tensor(mvl, device='cuda'), data,
                             torch.tensor(vl, device='cuda'))
    assert np.array_equal(m.cpu().numpy(), ref.gen_mem_attn_mask((B, Lm), mvl, (B, Lq), vl))
    m = ac.gen_mem_attn_mask(mem.transpose(0, 1), torch.tensor(mvl, device='cuda'),
                             data.transpose(0, 1), None, layout='TN')
    assert np.array_equal(m.cpu().numpy(), ref.gen_mem_attn_mask((B, Lm), mvl, (B, Lq), None))
    # sequence_mask through the C ABI (both layouts, vector and scalar paths)
    import ctypes
    from gluonnlp_amd import _lib, _tensor as T
    lib = _lib.load()
    for C in [8, 5]:
        for dtype in [torch.float32, torch.bfloat16]:
            x = torch.randn(B, Lq, C, device='cuda').to(dtype)
            vlt = torch.tensor(vl, device='cuda', dtype=torch.int32)
            out = torch.empty_like(x)
            _lib.check(lib.gnlp_sequence_mask(T.ptr(out), T.ptr(x), T.ptr(vlt), _lib.I32,
                                              T.code(x), B, Lq, C, 0, None))
            exp = ref.sequence_mask(to_np(x), vl, 1)
            assert np.array_equal(to_np(out), exp)
            xt = x.transpose(0, 1).contiguous()
            _lib.check(lib.gnlp_sequence_mask(T.ptr(xt), T.ptr(xt), T.ptr(vlt), _lib.I32,
                                              T.code(xt), B, Lq, C, 1, None))
            assert np.array_equal(to_np(xt), np.swapaxes(exp, 0, 1))


# ----------------------------------------------------------------------------------- softmax
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 3, 4, 5), (2, 2, 16, 32), (1, 2, 3, 513), (1, 1, 2, 1500)])
def test_masked_softmax(nlp, dtype, shape):
    ac = nlp.attention_cell
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape) * 3
    mask = rng.integers(0, 2, size=(shape[0], 1, shape[2], shape[3])).astype(bool)
    mask[0, 0, 0, :] = False
    xd = to_dev(x, dtype)
    xo = prep(x, dtype)
    for temperature in [None, 1.7]:
        out = ac.masked_softmax(xd, torch.tensor(mask, device='cuda'), temperature=temperature)
        exp = ref.masked_softmax(xo, mask, temperature=temperature)
        assert rel_err(to_np(out), exp) < tol(dtype)
        assert (to_np(out)[np.broadcast_to(~mask, shape)] == 0).all()
        assert (to_np(out)[0, :, 0, :] == 0).all()
    out = ac.masked_softmax(xd, None, temperature=2.0)
    assert rel_err(to_np(out), ref.masked_softmax(xo, None, temperature=2.0)) < tol(dtype)
    # int mask, axis != -1
    out = ac.masked_softmax(xd, torch.tensor(mask.astype(np.int32), device='cuda'), axis=2)
    exp = ref.masked_softmax(xo, mask, axis=2)
    assert rel_err(to_np(out), exp) < tol(dtype)
    if dtype == torch.float32:
        ls = ac.masked_logsoftmax(xd, torch.tensor(mask, device='cuda'))
        exp = ref.masked_logsoftmax(xo, mask)
        got = to_np(ls)
        bm = np.broadcast_to(mask, shape)
        assert np.isneginf(got[~bm]).all()
        assert np.abs(got[bm] - exp[bm]).max() < 1e-4


# ----------------------------------------------------------------------------------- small ops
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_l2_normalize_and_activations(nlp, dtype):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 5, 7, 8))
    xd, xo = to_dev(x, dtype), prep(x, dtype)
    out = nlp.op.l2_normalize(xd, axis=-1, eps=1e-6)
    assert rel_err(to_np(out), ref.l2_normalize(xo, -1, 1e-6)) < tol(dtype)
    out = nlp.op.l2_normalize(xd, axis=1, eps=1e-3)
    assert rel_err(to_np(out), ref.l2_normalize(xo, 1, 1e-3)) < tol(dtype)
    x = rng.standard_normal((4, 33)) * 3
    xd, xo = to_dev(x, dtype), prep(x, dtype)
    for act in ['relu', 'gelu', 'gelu(tanh)', 'gelu(sigmoid)', 'sigmoid', 'tanh', 'leaky', 'elu',
                'softrelu', 'softsign', 'identity']:
        f = nlp.layers.get_activation(act)
        assert rel_err(to_np(f(xd)), ref.activation(xo, act)) < tol(dtype), act
    with pytest.raises(NotImplementedError):
        nlp.layers.get_activation('swish')
    with pytest.raises(ValueError):
        nlp.layers.GELU('foo')
    with pytest.raises(NotImplementedError):
        nlp.layers.get_norm_layer('group_norm')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C', [48, 50, 768, 1024, 4096])
def test_norms(nlp, dtype, C):
    from gluonnlp_amd.layers import LayerNorm, NoNorm, RMSNorm
    rng = np.random.default_rng(C)
    x = rng.standard_normal((3, 7, C)) * 2 + 0.5
    g = 1 + 0.1 * rng.standard_normal(C)
    b = 0.1 * rng.standard_normal(C)
    xd, xo, go, bo = to_dev(x, dtype), prep(x, dtype), prep(g, dtype), prep(b, dtype)
    for eps in [1e-12, 1e-5]:
        ln = LayerNorm(epsilon=eps, in_channels=C).cuda()
        ln.gamma.data, ln.beta.data = to_dev(g, dtype), to_dev(b, dtype)
        assert rel_err(to_np(ln(xd)), ref.layer_norm(xo, go, bo, eps)) < tol(dtype)
    rn = RMSNorm(in_channels=C, variance_epsilon=1e-6).cuda()
    rn.gamma.data, rn.beta.data = to_dev(g, dtype), to_dev(b, dtype)
    assert rel_err(to_np(rn(xd)), ref.rms_norm(xo, go, bo, 1e-6)) < tol(dtype)
    nn_ = NoNorm(in_channels=C).cuda()
    nn_.gamma.data, nn_.beta.data = to_dev(g, dtype), to_dev(b, dtype)
    assert rel_err(to_np(nn_(xd)), ref.no_norm(xo, go, bo)) < tol(dtype)
    # fused residual
    r = rng.standard_normal((3, 7, C))
    ln = LayerNorm(epsilon=1e-5, in_channels=C).cuda()
    ln.gamma.data, ln.beta.data = to_dev(g, dtype), to_dev(b, dtype)
    got = ln(xd, residual=to_dev(r, dtype))
    assert rel_err(to_np(got), ref.layer_norm(xo + prep(r, dtype), go, bo, 1e-5)) < tol(dtype)


# ----------------------------------------------------------------------------------- buckets
@pytest.mark.parametrize('bidirectional', [True, False])
@pytest.mark.parametrize('num_buckets,max_distance', [(32, 128), (64, 256), (32, 256), (8, 20)])
def test_bucket_bit_exact(nlp, bidirectional, num_buckets, max_distance):
    rel = np.concatenate([np.arange(-10000, 10001), np.arange(-10, 11),
                          np.array([-2 ** 20, 2 ** 20, 2 ** 30, -2 ** 30])]).astype(np.int32)
    got = nlp.op.relative_position_bucket(torch.tensor(rel, device='cuda'), bidirectional,
                                          num_buckets, max_distance)
    assert got.dtype == torch.int32
    exp = ref.relative_position_bucket(rel, bidirectional, num_buckets, max_distance)
    assert np.array_equal(got.cpu().numpy(), exp)


def test_rel_position_and_t5_bias(nlp):
    ac = nlp.attention_cell
    L, H = 1024, 12
    data = torch.zeros((2, L, 4), device='cuda')
    rp = ac.gen_rel_position(data)
    assert rp.dtype == torch.int32
    exp = ref.gen_rel_position((2, L))
    assert np.array_equal(rp.cpu().numpy(), exp)
    assert np.array_equal(ac.gen_rel_position(data.transpose(0, 1), layout='TN').cpu().numpy(), exp)
    # incremental decoding (attention_cell.py:644-652): positions over [past_data, data]
    step = torch.zeros((2, 1, 4), device='cuda')
    past = torch.zeros((2, 6, 3, 8), device='cuda')                       # (B, L_past, H, d_kv)
    rp7 = ac.gen_rel_position(step, past_data=past)
    assert np.array_equal(rp7.cpu().numpy(), ref.gen_rel_position((2, 1, 4), (2, 6, 3, 8)))
    assert tuple(rp7.shape) == (7, 7) and int(rp7[6, 0]) == -6
    rp7t = ac.gen_rel_position(step.transpose(0, 1), past_data=past.transpose(0, 1), layout='TN')
    assert torch.equal(rp7, rp7t)
    # dense (L, L) grid of buckets at the cfg4 size: bit-exact
    b = nlp.op.relative_position_bucket(rp)
    assert np.array_equal(b.cpu().numpy(), ref.relative_position_bucket(exp))
    for dtype in [torch.float32, torch.bfloat16]:
        cell = ac.RelAttentionScoreCell(query_units=H * 64, num_heads=H, method='t5',
                                        bidirectional=True,
                                        dtype='float32' if dtype == torch.float32 else 'bfloat16')
        cell.cuda()
        w = np.random.default_rng(0).standard_normal((32, H)) * 768 ** -0.5
        cell._rel_pos_embed.weight.data = to_dev(w, dtype)
        wo = prep(w, dtype)
        Ls = 96
        rps = ac.gen_rel_position(torch.zeros((1, Ls, 1), device='cuda'))
        got = cell(rps)                                  # reference call path
        fused = cell.self_attn_bias(Ls)                  # fused kernel
        exp_b = ref.rel_attention_score_t5(wo, ref.gen_rel_position((1, Ls)))
        assert np.array_equal(to_np(got), exp_b)         # a gather: exact
        assert np.array_equal(to_np(fused), exp_b)
    # reference test properties (tests/test_layers.py:215-224)
    emb = nlp.layers.BucketPositionalEmbedding(units=4, num_buckets=32, max_distance=128).cuda()
    rel = torch.tensor([[3, 3, 200, 500, -200, -500]], device='cuda', dtype=torch.int32)
    e = to_np(emb(rel))
    assert np.array_equal(e[0, 0], e[0, 1]) and np.array_equal(e[0, 2], e[0, 3])
    assert np.array_equal(e[0, 4], e[0, 5])
    with pytest.raises(NotImplementedError):
        ac.RelAttentionScoreCell(query_units=8, num_heads=2, method='rotary')


# ----------------------------------------------------------------------------------- Dense
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [
    (8, 64, 48),          # generic VALU path (K % 32 != 0)
    (37, 72, 64),         # MFMA, ragged M and N
    (300, 128, 128),      # MFMA 128x128 tiles
    (512, 768, 256),      # bf16: 256x256-tile configuration needs >= 128 tiles -> stays 128
    (4096, 4096, 128),    # bf16: 256x256 tiles (256 of them)
    (2304, 4096, 128),    # bf16: 144 tiles of 256x256 on 256 CUs (persistent, one tile each)
    (4100, 4104, 192),    # 256x256 tiles with ragged edges
])
def test_dense(nlp, dtype, M, N, K):
    from gluonnlp_amd.layers import Dense
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K))
    w = rng.standard_normal((N, K)) * 0.1
    b = rng.standard_normal(N)
    o = rng.standard_normal((M, N))
    name = 'float32' if dtype == torch.float32 else 'bfloat16'
    d = Dense(N, K, dtype=name).cuda()
    d.weight.data, d.bias.data = to_dev(w, dtype), to_dev(b, dtype)
    xd, od = to_dev(x, dtype), to_dev(o, dtype)
    xo, wo, bo, oo = prep(x, dtype), prep(w, dtype), prep(b, dtype), prep(o, dtype)
    base = ref.dense(xo, wo, bo)
    t = tol(dtype)
    assert rel_err(to_np(d(xd)), base) < t
    assert rel_err(to_np(d(xd, act='gelu')), ref.activation(base, 'gelu')) < t
    assert rel_err(to_np(d(xd, act='relu')), ref.activation(base, 'relu')) < t
    assert rel_err(to_np(d(xd, act='gelu(tanh)')), ref.activation(base, 'gelu(tanh)')) < t
    assert rel_err(to_np(d(xd, act='tanh')), ref.activation(base, 'tanh')) < t   # unfused act
    assert rel_err(to_np(d(xd, other=od)), base + oo) < t
    assert rel_err(to_np(d(xd, other=od, combine='mul', act='gelu(tanh)')),
                   ref.activation(base, 'gelu(tanh)') * oo) < t
    nb = Dense(N, K, use_bias=False, dtype=name).cuda()
    nb.weight.data = to_dev(w, dtype)
    assert rel_err(to_np(nb(xd)), ref.dense(xo, wo)) < t


@pytest.mark.parametrize('M,N,K', [
    (8192, 4096, 192),    # 512 tiles: two per workgroup, tile switch with nk = 3
    (4608, 4096, 128),    # 288 tiles: workgroups with one and with two tiles, nk = 2 (minimum)
    (2048, 8192, 320),    # widest N the LDS bias image holds
    (2048, 8448, 128),    # N > 8192: one-tile-per-workgroup kernel
    (16384, 768, 192),    # 256x192 tiles (N = 768): 256 tiles, one per workgroup
    (32768, 768, 128),    # 256x192 tiles: 512 tiles, two per workgroup, nk = 2
    (8448, 2304, 320),    # 256x192 tiles: 396 tiles, workgroups with one and with two
    (8192, 960, 256),     # N a multiple of 192 only: 160 tiles of 256x192
])
def test_dense_persistent(nlp, M, N, K, monkeypatch):
    """The persistent big-GEMM kernels (csrc/gemm_bf16_w4.hip, csrc/gemm_bf16_ps.hip) against the
    oracle, and bit for bit against each other and the one-tile-per-workgroup kernel (same
    accumulation order, same fp32 epilogue)."""
    from gluonnlp_amd.layers import Dense
    dtype = torch.bfloat16
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K))
    w = rng.standard_normal((N, K)) * 0.1
    b = rng.standard_normal(N)
    o = rng.standard_normal((M, N))
    d = Dense(N, K, dtype='bfloat16').cuda()
    d.weight.data, d.bias.data = to_dev(w, dtype), to_dev(b, dtype)
    nb = Dense(N, K, use_bias=False, dtype='bfloat16').cuda()
    nb.weight.data = to_dev(w, dtype)
    xd, od = to_dev(x, dtype), to_dev(o, dtype)
    xo, wo, bo, oo = prep(x, dtype), prep(w, dtype), prep(b, dtype), prep(o, dtype)
    base = ref.dense(xo, wo, bo)
    t = tol(dtype)

    def variants():
        return [d(xd), d(xd, act='gelu'), d(xd, act='relu'), d(xd, other=od),
                d(xd, other=od, combine='mul', act='gelu(tanh)'), nb(xd), nb(xd, other=od)]
    got = variants()
    want = [base, ref.activation(base, 'gelu'), ref.activation(base, 'relu'), base + oo,
            ref.activation(base, 'gelu(tanh)') * oo, ref.dense(xo, wo), ref.dense(xo, wo) + oo]
    for g, e in zip(got, want):
        assert rel_err(to_np(g), e) < t
    from gluonnlp_amd import _lib
    reload_knobs = _lib.load().gnlp_reload_knobs   # the knobs are read once per process
    # a second call on the same buffers gives the same bits (no state left in LDS / scratch)
    for g, h in zip(got, variants()):
        assert torch.equal(g, h)
    try:
        # the two 256x256x64 kernels -- persistent with one wave per SIMD (gemm_bf16_w4.hip, the
        # default) and one tile per workgroup (gemm_bf16_pp.hip, what ragged shapes take) --
        # accumulate in the same order with the same fp32 epilogue: bit for bit equal
        monkeypatch.setenv('GNLP_GEMM_W4', '0')
        reload_knobs()
        for g, h in zip(got, variants()):
            assert torch.equal(g, h)
    finally:
        monkeypatch.delenv('GNLP_GEMM_W4', raising=False)
        reload_knobs()


@pytest.mark.parametrize('M,N,K', [
    (1, 768, 768),        # a single row (one sequence decoding): FM = 1
    (16, 3072, 768),      # one full fragment of rows
    (17, 772, 3072),      # FM = 2, ragged rows, N not a multiple of 16, long K (12 k-steps a wave)
    (33, 64, 64),         # FM = 4 in one group, K = 64: six of the eight waves get no k-step
    (64, 2304, 768),      # 64-row group
    (65, 768, 1024),      # two row groups, the second with one live row
    (128, 768, 768),      # 32-row groups (fewer than 128 workgroups of 64 rows)
    (384, 3072, 768),     # the last row count the kernel takes by default
])
def test_dense_few_rows(nlp, M, N, K, monkeypatch):
    """The K-split bf16 kernel for few rows (decoding steps, small-batch inference;
    csrc/gemm_bf16.hip dense_bf16_skinny_kernel) against the oracle on every epilogue, and against
    the tiled kernel on the same buffers (same products, another summation order: fp32 rounding
    of the accumulator apart); a strided input (the pooler's CLS rows) through the C ABI."""
    import ctypes
    from gluonnlp_amd import _lib
    from gluonnlp_amd.layers import Dense
    dtype = torch.bfloat16
    rng = np.random.default_rng(M * 7 + N + K)
    x = rng.standard_normal((M, K))
    w = rng.standard_normal((N, K)) * 0.1
    b = rng.standard_normal(N)
    o = rng.standard_normal((M, N))
    d = Dense(N, K, dtype='bfloat16').cuda()
    d.weight.data, d.bias.data = to_dev(w, dtype), to_dev(b, dtype)
    nb = Dense(N, K, use_bias=False, dtype='bfloat16').cuda()
    nb.weight.data = to_dev(w, dtype)
    xd, od = to_dev(x, dtype), to_dev(o, dtype)
    xo, wo, bo, oo = prep(x, dtype), prep(w, dtype), prep(b, dtype), prep(o, dtype)
    base = ref.dense(xo, wo, bo)
    t = tol(dtype)

    def variants():
        return [d(xd), d(xd, act='gelu'), d(xd, act='relu'), d(xd, act='gelu(tanh)'),
                d(xd, other=od), d(xd, other=od, combine='mul', act='gelu(tanh)'), nb(xd),
                nb(xd, other=od)]
    want = [base, ref.activation(base, 'gelu'), ref.activation(base, 'relu'),
            ref.activation(base, 'gelu(tanh)'), base + oo, ref.activation(base, 'gelu(tanh)') * oo,
            ref.dense(xo, wo), ref.dense(xo, wo) + oo]
    got = variants()
    for g, e in zip(got, want):
        assert rel_err(to_np(g), e) < t
    for g, h in zip(got, variants()):
        assert torch.equal(g, h)
    reload_knobs = _lib.load().gnlp_reload_knobs
    try:
        monkeypatch.setenv('GNLP_SKINNY_MAX_M', '0')          # the tiled kernel
        reload_knobs()
        for g, h in zip(got, variants()):
            assert rel_err(to_np(g), to_np(h)) < 1e-2
    finally:
        monkeypatch.delenv('GNLP_SKINNY_MAX_M', raising=False)
        reload_knobs()
    # rows 0, 3, 6, ... of a wider buffer (ldx = 3 K), output rows 2 N apart
    xs = torch.zeros((M, 3 * K), dtype=dtype, device='cuda')
    xs[:, :K] = xd
    out = torch.full((M, 2 * N), 7.0, dtype=dtype, device='cuda')
    rc = _lib.load().gnlp_dense(out.data_ptr(), xs.data_ptr(), d.weight.data_ptr(),
                                d.bias.data_ptr(), None, _lib.COMBINE_ADD, _lib.ACT['identity'],
                                _lib.BF16, M, N, K, 3 * K, 2 * N, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out[:, :N], got[0]) and (out[:, N:] == 7.0).all()


def test_dense_mfma_layout_asymmetric(nlp):
    """A = I against an asymmetric B catches transposed fragment maps (exact integer data)."""
    from gluonnlp_amd.layers import Dense
    for dtype, name, n in [(torch.bfloat16, 'bfloat16', 256), (torch.float32, 'float32', 128)]:
        K = n
        x = np.eye(n, K)
        w = (np.arange(n)[:, None] % 7 - 3) + (np.arange(K)[None, :] % 5) * 0.5   # asymmetric
        d = Dense(n, K, use_bias=False, dtype=name).cuda()
        d.weight.data = to_dev(w, dtype)
        got = to_np(d(to_dev(x, dtype)))
        assert np.array_equal(got, x @ w.T)


@pytest.mark.parametrize('M', [256, 32768])
def test_dense_gelu_epilogue_wide_range(nlp, M):
    """The bf16 GELU epilogue (x * sat(0.5 + u Q(u^2)), u = x / 4, u^2 and Phi saturated through the
    clamp modifier; round 2: a sigmoid of a polynomial) over pre-activations in [-30, 30], through the 128x128 kernel (M = 256) and the
    persistent one (M = 32768): the erf GELU within bf16 resolution everywhere, exactly x for large
    x and (-)0 for very negative x -- nothing turns over where the polynomial would."""
    from gluonnlp_amd.layers import Dense
    N = K = 256
    d = Dense(N, K, dtype='bfloat16').cuda()
    d.weight.data = torch.eye(N, device='cuda').to(torch.bfloat16)
    d.bias.data.zero_()
    vals = torch.linspace(-30.0, 30.0, N * 8, device='cuda').to(torch.bfloat16)
    x = vals.reshape(8, N).repeat(M // 8, 1).contiguous()
    y = d(x, act='gelu')
    xo = to_np(x[:8])
    exp = ref.activation(xo, 'gelu')
    got = to_np(y[:8])
    assert np.all(np.abs(got - exp) <= 4e-3 * np.abs(exp) + 3e-4)
    assert np.all(got[xo > 9] == xo[xo > 9])
    assert np.all(np.abs(got[xo < -9]) < 1e-10)
    assert torch.equal(y[:8], y[-8:])
