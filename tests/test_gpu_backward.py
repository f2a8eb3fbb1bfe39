"""GPU parity tests of the backward pass, through torch autograd -> the C ABI
(gnlp_multi_head_dot_attn_backward / gnlp_norm_backward / gnlp_l2_normalize_backward).

What the reference checks (tests/test_attention_cell.py:82-151): the gradients of query / key /
value / rel_scores after ``out.backward()`` agree across the NKT / NTK / TNK layouts.  Here they
are additionally compared with known answers: float64 autograd through the reference's PyTorch
sibling (tests/golden/grad.npz, oracle/gen_golden_grad.py) and the float64 oracle restatement
(oracle/ref.py: multi_head_dot_attn_backward, layer_norm_backward ...).
Tolerances: fp32 1e-5, bf16 2e-2 (tests/util.py rel_err: global and element-wise).  Where a case
cannot be held to them by ANY evaluation at the working precision, the bar is MEASURED, not argued:
tests/util.py attn_backward_floor evaluates the same case plainly at that precision (float32 numpy;
float64 with the roundings of a bf16-MFMA backward) and the HIP result must stay within
max(tolerance, 2 x that floor) -- ``bar()`` below.  tests/dev_bwd_floor.py prints both columns."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ref
from tests.util import attn_backward_floor, prep, rel_err, to_dev, to_np, tol

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), 'golden')

LAYOUTS = [('NKT', None), ('NTK', (0, 2, 1, 3)), ('TNK', (2, 0, 1, 3))]


@pytest.fixture(scope='module')
def nlp():
    import gluonnlp_amd
    from gluonnlp_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return gluonnlp_amd


def bar(t, floor, exp):
    """max(t, 2 x the worst error of the plain evaluation ``floor`` against ``exp``)."""
    worst = max(float(rel_err(f, b)) for f, b in zip(floor, exp) if b is not None)
    return max(t, 2.0 * worst)


def lay(x, perm):
    return x if perm is None else np.transpose(x, perm)


def run_attn(ac, layout, perm, q, k, v, mask, e, g_ctx, scaled, dtype, units, normalized=False,
             **kw):
    """-> ctx, dq, dk, dv, de as float64 numpy, gradients brought back to NKT."""
    qq, kk, vv = [to_dev(lay(x, perm), dtype).requires_grad_(True) for x in (q, k, v)]
    ee = None if e is None else to_dev(e, dtype).requires_grad_(True)
    out, (scores, w) = ac.multi_head_dot_attn(qq, kk, vv, mask, ee, scaled=scaled,
                                              normalized=normalized, query_head_units=units,
                                              layout=layout, **kw)
    assert out.requires_grad
    assert scores is None or not scores.requires_grad
    g = g_ctx if layout != 'TNK' else np.transpose(g_ctx, (1, 0, 2))
    out.backward(to_dev(g, dtype))
    inv = None if perm is None else tuple(np.argsort(perm))
    res = [to_np(out) if layout != 'TNK' else np.transpose(to_np(out), (1, 0, 2))]
    for t in (qq, kk, vv):
        assert t.grad is not None and t.grad.shape == t.shape and t.grad.dtype == dtype
        res.append(lay(to_np(t.grad), inv))
    res.append(None if ee is None else to_np(ee.grad))
    return res


# ------------------------------------------------------------------------------- attention
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('tag,biases', [('small', ('none', 'share', 'head')),
                                        ('d64', ('none', 'head'))])
def test_attention_backward_goldens(nlp, tag, biases, dtype):
    """float64 autograd through the reference's torch sibling; all three layouts."""
    ac = nlp.attention_cell
    g = np.load(os.path.join(G, 'grad.npz'))
    q, k, v, gc, mask = [g['{}_{}'.format(tag, n)] for n in ('q', 'k', 'v', 'g_ctx', 'mask')]
    t = tol(dtype)
    md = torch.tensor(mask, device='cuda')
    for scaled in (1, 0):
        for ename in biases:
            e = None if ename == 'none' else g['{}_e_{}'.format(tag, ename)]
            key = '{}_s{}_{}'.format(tag, scaled, ename)
            if dtype == torch.float32:
                exp = [g[key + '_' + n] for n in ('ctx', 'dq', 'dk', 'dv')]
                exp.append(g[key + '_de'] if e is not None else None)
                # d = 64 un-scaled: logits ~ N(0, 64) reach |s| ~ 30 where one fp32 ulp is 1.9e-6:
                # merely HOLDING a logit in fp32 moves its weight by ~1e-6 relative.  The float32
                # numpy evaluation of the case measures 0.5-1.0e-5 there (HIP: 0.5-1.7e-5), so
                # the bar is max(1e-5, 2 x floor) ~ 2e-5 (it was an argued 1e-4)
                te = t
                if tag == 'd64' and not scaled:
                    te = bar(t, attn_backward_floor(gc, q, k, v, mask, e, False, 'f32'), exp)
                    assert te < 3e-5
            else:   # what the device holds is the bf16 rounding of the inputs
                te = t
                pq, pk, pv, pg = [prep(x, dtype) for x in (q, k, v, gc)]
                pe = None if e is None else prep(e, dtype)
                ectx, _ = ref.multi_head_dot_attn(pq, pk, pv, mask, pe, scaled=bool(scaled),
                                                  query_head_units=q.shape[-1])
                exp = [ectx] + list(ref.multi_head_dot_attn_backward(
                    pg, pq, pk, pv, mask, pe, scaled=bool(scaled), query_head_units=q.shape[-1]))
            for layout, perm in LAYOUTS:
                got = run_attn(ac, layout, perm, q, k, v, md, e, gc, bool(scaled), dtype,
                               q.shape[-1])
                for name, a, b in zip(('ctx', 'dq', 'dk', 'dv', 'de'), got, exp):
                    if b is None:
                        assert a is None
                        continue
                    assert rel_err(a, b) < te, (key, layout, name)
    if tag == 'small':   # the fully masked query row takes and gives no gradient
        assert (got[1][0, :, 3] == 0).all()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_backward_valid_length_predicate(nlp, dtype):
    """A mask straight from gen_self_attn_mask is evaluated as the (i < vl) & (j < vl) predicate in
    the backward kernels too: same answers as the dense bytes (goldens of the d = 64 case)."""
    ac = nlp.attention_cell
    g = np.load(os.path.join(G, 'grad.npz'))
    q, k, v, gc = [g['d64_' + n] for n in ('q', 'k', 'v', 'g_ctx')]
    vl = torch.tensor(g['d64_vl'], device='cuda')
    e = g['d64_e_head']
    B, H, L, d = q.shape
    mask = ac.gen_self_attn_mask(torch.zeros((B, L, 1), device='cuda'), vl)
    assert ac.mask_predicate(mask) is not None
    dense = mask.clone()                                   # untagged copy -> byte path
    a = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, mask, e, gc, True, dtype, d)
    b = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, dense, e, gc, True, dtype, d)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    if dtype == torch.float32:
        for name, x in zip(('ctx', 'dq', 'dk', 'dv', 'de'), a):
            assert rel_err(x, g['d64_s1_head_' + name]) < 1e-5, name
    # the fused bf16 forward (nothing but ctx wanted) hands its row statistics to the backward,
    # which then takes D = <dO, O> instead of recomputing it: both against the oracle on the
    # bf16-rounded inputs, bar = max(2e-2, 2 x measured floor) (measured 0.7-1.1e-2 = the floor of
    # rounding the outputs to bf16 alone; it was an argued 3e-2 against the UNROUNDED inputs' golden)
    if dtype == torch.bfloat16:
        c = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, mask, e, gc, True, dtype, d,
                     _need_scores=False, _need_weights=False)
        from gluonnlp_amd import _lib
        assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_FUSED_BF16
        pq, pk, pv, pg, pe = [prep(x, dtype) for x in (q, k, v, gc, e)]   # what the device holds
        mk = mask.cpu().numpy()
        exp = ref.multi_head_dot_attn_backward(pg, pq, pk, pv, mk, pe, scaled=True,
                                               query_head_units=d)
        tb = bar(tol(dtype), attn_backward_floor(pg, pq, pk, pv, mk, pe, True, 'bf16')[1:], exp)
        assert tb < 2.5e-2
        for name, x, y, w in zip(('dq', 'dk', 'dv', 'de'), a[1:], c[1:], exp):
            assert rel_err(y, w) < tb, name
            assert rel_err(x, w) < tb, name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('Lq,Lm,d,fused', [(70, 130, 64, True), (70, 130, 64, False), (33, 200, 24, False),
                                           (130, 40, 64, True)])
def test_attention_backward_memory_mask_predicate(nlp, dtype, Lq, Lm, d, fused):
    """A mask straight from gen_mem_attn_mask is evaluated as (j < mem_vl[b]) & (i < data_vl[b])
    in the backward kernels too (general fp32-MFMA pair, bf16-MFMA pair, with and without the
    fused forward's saved statistics): the gradients of the dense bytes and of the float64 oracle;
    switched-off keys and query rows take no gradient."""
    ac = nlp.attention_cell
    B, H = 3, 2
    rng = np.random.default_rng(Lq * 7 + Lm + d)
    q, k, v = (rng.standard_normal((B, H, Lq, d)), rng.standard_normal((B, H, Lm, d)),
               rng.standard_normal((B, H, Lm, d)))
    gc = rng.standard_normal((B, Lq, H * d))
    e = rng.standard_normal((H, Lq, Lm))
    mem_vl, data_vl = np.array([Lm, 1, Lm * 2 // 3]), np.array([Lq // 2, Lq, 1])
    tagged = ac.gen_mem_attn_mask(torch.zeros((B, Lm, 1), device='cuda'), torch.tensor(mem_vl, device='cuda'),
                                  torch.zeros((B, Lq, 1), device='cuda'), torch.tensor(data_vl, device='cuda'))
    assert ac.mask_predicate(tagged, mem=True)[0] == 'mem'
    kw = dict(_need_scores=False, _need_weights=False) if fused else {}
    a = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, tagged, e, gc, True, dtype, d, **kw)
    b = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, tagged.clone(), e, gc, True, dtype, d, **kw)
    mask_np = ref.gen_mem_attn_mask((B, Lm), mem_vl, (B, Lq), data_vl).astype(bool)
    pq, pk, pv, pg, pe = [prep(x, dtype) for x in (q, k, v, gc, e)]
    want = [ref.multi_head_dot_attn(pq, pk, pv, mask_np, pe, scaled=True, query_head_units=d)[0]] + \
        list(ref.multi_head_dot_attn_backward(pg, pq, pk, pv, mask_np, pe, scaled=True,
                                              query_head_units=d))
    t = tol(dtype)
    if dtype == torch.bfloat16:   # the bf16-MFMA pair: held to max(2e-2, 2 x the measured floor)
        t = min(bar(t, attn_backward_floor(pg, pq, pk, pv, mask_np, pe, True, 'bf16'), want), 3e-2)
    for name, x, y, w in zip(('ctx', 'dq', 'dk', 'dv', 'de'), a, b, want):
        assert np.isfinite(x).all(), name
        assert rel_err(x, y) < t, name
        assert rel_err(x, w) < t, name
    for bb in range(B):
        assert (a[1][bb, :, data_vl[bb]:] == 0).all()            # dq of padded queries
        assert (a[2][bb, :, mem_vl[bb]:] == 0).all() and (a[3][bb, :, mem_vl[bb]:] == 0).all()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,H,Lq,Lm,dk,dv,eshape', [
    (2, 3, 70, 130, 96, 128, 'bh'),      # 128-wide template, cross attention, ragged tiles
    (1, 2, 33, 31, 20, 32, '1h'),        # 32-wide template, bias shared over the batch (T5)
    (2, 1, 257, 64, 64, 40, 'ij'),       # > 2 workgroups of queries, bias shared over b and h
    (3, 2, 1, 5, 8, 8, None),            # a single query
    (2, 2, 130, 300, 64, 64, 'b1'),      # bias shared over the heads only
])
def test_attention_backward_shapes(nlp, dtype, B, H, Lq, Lm, dk, dv, eshape):
    ac = nlp.attention_cell
    rng = np.random.default_rng(B * 1000 + Lq + Lm + dk)
    q = rng.standard_normal((B, H, Lq, dk))
    k = rng.standard_normal((B, H, Lm, dk))
    v = rng.standard_normal((B, H, Lm, dv))
    gc = rng.standard_normal((B, Lq, H * dv))
    k[0, 0, Lm - 1] = q[0, 0, 0] * 2.0      # a late spike: the running max moves in the last tile
    mask = rng.integers(0, 4, size=(B, Lq, Lm)) > 0
    mask[0, Lq // 2] = False                 # a fully masked row
    e = {None: None, 'bh': (B, H, Lq, Lm), '1h': (1, H, Lq, Lm), 'ij': (Lq, Lm),
         'b1': (B, 1, Lq, Lm)}[eshape]
    if e is not None:
        e = rng.standard_normal(e)
    pq, pk, pv, pg = [prep(x, dtype) for x in (q, k, v, gc)]
    pe = None if e is None else prep(e, dtype)
    ectx, _ = ref.multi_head_dot_attn(pq, pk, pv, mask, pe, scaled=True, query_head_units=dk)
    exp = [ectx] + list(ref.multi_head_dot_attn_backward(pg, pq, pk, pv, mask, pe, scaled=True,
                                                         query_head_units=dk))
    md = torch.tensor(mask, device='cuda')
    t = tol(dtype)
    for layout, perm in LAYOUTS:
        got = run_attn(ac, layout, perm, q, k, v, md, e, gc, True, dtype, dk)
        for name, a, b in zip(('ctx', 'dq', 'dk', 'dv', 'de'), got, exp):
            if b is None:
                assert a is None
            else:
                assert a.shape == b.shape
                assert rel_err(a, b) < t, (layout, name)


def test_attention_backward_normalized_and_partial_grads(nlp):
    """normalized=True chains l2_normalize's backward (reference test parameter `normalized`);
    only the tensors that ask get a gradient."""
    ac = nlp.attention_cell
    rng = np.random.default_rng(11)
    B, H, Lq, Lm, d = 2, 2, 16, 32, 8
    q, k = rng.standard_normal((B, H, Lq, d)), rng.standard_normal((B, H, Lm, d))
    v = rng.standard_normal((B, H, Lm, 6))
    gc = rng.standard_normal((B, Lq, H * 6))
    mask = rng.integers(0, 2, size=(B, Lq, Lm))
    qn, kn = ref.l2_normalize(q, -1, 1e-6), ref.l2_normalize(k, -1, 1e-6)
    dqn, dkn, dv_, _ = ref.multi_head_dot_attn_backward(gc, qn, kn, v, mask, None, scaled=True,
                                                        query_head_units=d)
    exp = [ref.l2_normalize_backward(dqn, q, -1, 1e-6), ref.l2_normalize_backward(dkn, k, -1, 1e-6),
           dv_]
    got = run_attn(ac, 'NKT', None, q, k, v, torch.tensor(mask, device='cuda'), None, gc, True,
                   torch.float32, d, normalized=True)
    for name, a, b in zip(('dq', 'dk', 'dv'), got[1:4], exp):
        assert rel_err(a, b) < 1e-5, name
    # partial: only key asks
    qq, kk, vv = [to_dev(x) for x in (q, k, v)]
    kk.requires_grad_(True)
    out, _ = ac.multi_head_dot_attn(qq, kk, vv, torch.tensor(mask, device='cuda'), scaled=True,
                                    query_head_units=d)
    out.backward(to_dev(gc))
    _, edk, _, _ = ref.multi_head_dot_attn_backward(gc, q, k, v, mask, None, scaled=True,
                                                    query_head_units=d)
    assert qq.grad is None and vv.grad is None
    assert rel_err(to_np(kk.grad), edk) < 1e-5
    # no grad mode / nothing requires grad: plain forward, no graph
    out, _ = ac.multi_head_dot_attn(qq.detach(), kk.detach(), vv, scaled=True, query_head_units=d)
    assert not out.requires_grad
    with torch.no_grad():
        out, _ = ac.multi_head_dot_attn(qq, kk, vv, scaled=True, query_head_units=d)
    assert not out.requires_grad


def test_attention_backward_c_abi_errors(nlp):
    from gluonnlp_amd import _lib
    lib = _lib.load()
    d = _lib.AttnDesc()
    d.B, d.H, d.Lq, d.Lm, d.dk, d.dv = 1, 1, 4, 4, 256, 256
    gd = _lib.AttnGradDesc()
    x = torch.zeros(4 * 256, device='cuda')
    ws = torch.zeros(1024, dtype=torch.uint8, device='cuda')
    args = lambda dd, wsn: (x.data_ptr(), x.data_ptr(), x.data_ptr(), None, 0, None, x.data_ptr(),
                            x.data_ptr(), x.data_ptr(), x.data_ptr(), ctypes.byref(dd),
                            ctypes.byref(gd), _lib.F32, None, None, None, 0, None, None, 1.0,
                            ws.data_ptr(), wsn, None)
    assert lib.gnlp_multi_head_dot_attn_backward(*args(d, 1024)) == -2      # head dims > 128
    assert b'128' in lib.gnlp_last_error()
    d.dk = d.dv = 64
    assert lib.gnlp_attn_backward_workspace_bytes(ctypes.byref(d)) == 256
    assert lib.gnlp_multi_head_dot_attn_backward(*args(d, 8)) == -1         # workspace too small
    assert b'workspace' in lib.gnlp_last_error()


def test_attention_backward_full_size_properties(nlp):
    """BERT-large attention shape (cfg2: H = 16, L = 512, d = 64), bf16, valid_length mask: the
    size-independent identities of the softmax Jacobian --
      rows of d(edge) sum to 0;  sum_j dV_j = sum_i dO_i over the live rows;
      <dQ, Q> = <dK, K>  (both are <dS, Q K^T>)."""
    ac = nlp.attention_cell
    B, H, L, d = 4, 16, 512, 64
    gen = torch.Generator(device='cuda').manual_seed(3)
    q, k, v = [torch.randn((B, L, H, d), device='cuda', generator=gen).to(torch.bfloat16)
               .requires_grad_(True) for _ in range(3)]
    e = (0.5 * torch.randn((B, H, L, L), device='cuda', generator=gen)).to(torch.bfloat16) \
        .requires_grad_(True)
    vl = torch.tensor([512, 100, 333, 1], device='cuda')
    mask = ac.gen_self_attn_mask(torch.zeros((B, L, 1), device='cuda'), vl)
    out, _ = ac.multi_head_dot_attn(q, k, v, mask, e, scaled=True, query_head_units=d,
                                    layout='NTK', _need_scores=False, _need_weights=False)
    go = torch.randn(out.shape, device='cuda', generator=gen).to(torch.bfloat16)
    out.backward(go)
    de = e.grad.double()
    assert float(de.sum(-1).abs().max()) < 2e-2 * float(de.abs().max()) * 8
    live = (torch.arange(L, device='cuda')[None, :] < vl[:, None])          # (B, L)
    gsum = (go.double().reshape(B, L, H, d) * live[:, :, None, None]).sum(1)
    dvsum = v.grad.double().sum(1)
    assert float((gsum - dvsum).abs().max()) < 2e-2 * float(gsum.abs().max())
    lhs = (q.grad.double() * q.detach().double()).sum((1, 3))
    rhs = (k.grad.double() * k.detach().double()).sum((1, 3))
    assert float((lhs - rhs).abs().max()) < 2e-2 * float(lhs.abs().max())
    # padded rows / keys: no gradient at all
    dead = ~live
    assert float(q.grad.double()[dead].abs().max()) == 0.0
    assert float(k.grad.double()[dead].abs().max()) == 0.0
    assert float(v.grad.double()[dead].abs().max()) == 0.0


# ----------------------------------------------------------------------------------- norms
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('tag', ['ln48', 'ln1024'])
def test_layer_norm_backward_goldens(nlp, tag, dtype):
    g = np.load(os.path.join(G, 'grad.npz'))
    x, gam, bet, go = [g['{}_{}'.format(tag, n)] for n in ('x', 'gamma', 'beta', 'g')]
    eps = float(g[tag + '_eps'])
    ln = nlp.layers.get_norm_layer('layer_norm', epsilon=eps, in_channels=x.shape[-1]).cuda()
    ln.gamma.data, ln.beta.data = to_dev(gam, dtype), to_dev(bet, dtype)
    ln.gamma.requires_grad_(True)
    ln.beta.requires_grad_(True)
    xx = to_dev(x, dtype).requires_grad_(True)
    y = ln(xx)
    y.backward(to_dev(go, dtype))
    if dtype == torch.float32:
        exp = [g[tag + '_' + n] for n in ('y', 'dx', 'dgamma', 'dbeta')]
    else:
        px, pg, pb, pgo = [prep(a, dtype) for a in (x, gam, bet, go)]
        exp = [ref.layer_norm(px, pg, pb, eps)] + list(ref.layer_norm_backward(pgo, px, pg, eps))
    t = tol(dtype)
    for name, a, b in zip(('y', 'dx', 'dgamma', 'dbeta'),
                          (y, xx.grad, ln.gamma.grad, ln.beta.grad), exp):
        assert a.dtype == dtype
        assert rel_err(to_np(a), b) < t, name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('kind', ['layer_norm', 'rms_norm', 'no_norm'])
@pytest.mark.parametrize('rows,C', [(7, 768), (5000, 96), (33, 3000), (4100, 1024)])
def test_norm_backward_kinds_and_sizes(nlp, kind, rows, C, dtype):
    """register path (C <= 2048), the three-pass path (C = 3000) and the row-strided grid
    (rows > 4 x 1024); with a residual for layer_norm."""
    rng = np.random.default_rng(rows + C)
    x = rng.standard_normal((rows, C)) * 1.5 + 0.2
    res = rng.standard_normal((rows, C)) if kind == 'layer_norm' else None
    gam = 1.0 + 0.2 * rng.standard_normal(C)
    bet = 0.1 * rng.standard_normal(C)
    go = rng.standard_normal((rows, C))
    layer = nlp.layers.get_norm_layer(kind, epsilon=1e-5, in_channels=C).cuda()
    layer.gamma.data, layer.beta.data = to_dev(gam, dtype), to_dev(bet, dtype)
    layer.gamma.requires_grad_(True)
    layer.beta.requires_grad_(True)
    xx = to_dev(x, dtype).requires_grad_(True)
    rr = None if res is None else to_dev(res, dtype).requires_grad_(True)
    y = layer(xx) if rr is None else layer(xx, rr)
    y.backward(to_dev(go, dtype))
    px, pg, pgo = prep(x, dtype), prep(gam, dtype), prep(go, dtype)
    if res is not None:
        px = px + prep(res, dtype)
    if kind == 'layer_norm':
        edx, edg, edb = ref.layer_norm_backward(pgo, px, pg, 1e-5)
    elif kind == 'rms_norm':
        edx, edg, edb = ref.rms_norm_backward(pgo, px, pg, 1e-6)   # get_norm_layer drops epsilon
    else:
        edx, edg, edb = pgo * pg, (pgo * px).sum(0), pgo.sum(0)
    # dgamma / dbeta sum `rows` products: bf16 outputs, fp32 atomics -> the bf16 / fp32 bars hold
    t = tol(dtype)
    assert rel_err(to_np(xx.grad), edx) < t
    if rr is not None:
        assert torch.equal(rr.grad, xx.grad)
    assert rel_err(to_np(layer.gamma.grad), edg) < t
    assert rel_err(to_np(layer.beta.grad), edb) < t


def test_norm_backward_frozen_parameters_and_errors(nlp):
    """Parameters are frozen by default: only data gets a gradient; C > 8192 is refused."""
    from gluonnlp_amd import _lib
    ln = nlp.layers.LayerNorm(epsilon=1e-5, in_channels=64).cuda()
    x = torch.randn(5, 64, device='cuda', requires_grad=True)
    ln(x).sum().backward()
    assert x.grad is not None and ln.gamma.grad is None and ln.beta.grad is None
    assert float(x.grad.abs().max()) < 1e-5          # d/dx of sum(LN(x)) with gamma = 1 is 0
    big = nlp.layers.LayerNorm(epsilon=1e-5, in_channels=8200).cuda()
    xb = torch.randn(2, 8200, device='cuda', requires_grad=True)
    with pytest.raises(_lib.GnlpError):
        big(xb).sum().backward()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_l2_normalize_backward(nlp, dtype):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, 5, 40))
    x[0, 0] = 0.0                                     # a zero row: the norm's subgradient
    go = rng.standard_normal((3, 5, 40))
    xx = to_dev(x, dtype).requires_grad_(True)
    y = nlp.op.l2_normalize(xx, axis=-1, eps=1e-6)
    y.backward(to_dev(go, dtype))
    exp = ref.l2_normalize_backward(prep(go, dtype), prep(x, dtype), -1, 1e-6)
    assert rel_err(to_np(xx.grad)[1:], exp[1:]) < tol(dtype)
    assert rel_err(to_np(xx.grad)[0, 0], exp[0, 0]) < tol(dtype)
    # another axis goes through movedim (torch plumbing) and back
    x2 = to_dev(x, dtype).requires_grad_(True)
    nlp.op.l2_normalize(x2, axis=1, eps=1e-6).backward(to_dev(go, dtype))
    exp2 = ref.l2_normalize_backward(prep(go, dtype), prep(x, dtype), 1, 1e-6)
    assert rel_err(to_np(x2.grad), exp2) < tol(dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('L,d', [(150, 64), (40, 8), (70, 128)])
def test_attention_backward_causal_predicate(nlp, dtype, L, d):
    """gen_self_attn_mask(attn_type='causal') as the in-kernel predicate in both backward kernels
    (query tiles before a key block / key tiles after a query block are skipped): bit for bit the
    gradients of the same bytes handed over untagged, and the oracle's."""
    ac = nlp.attention_cell
    rng = np.random.default_rng(L * d)
    B, H = 2, 2
    q, k, v = [rng.standard_normal((B, H, L, d)) for _ in range(3)]
    gc = rng.standard_normal((B, L, H * d))
    e = rng.standard_normal((H, L, L))
    vl = np.array([L, L // 2])
    mask = ac.gen_self_attn_mask(torch.zeros((B, L, 1), device='cuda'),
                                 torch.tensor(vl, device='cuda'), attn_type='causal')
    dense = mask.clone()
    a = run_attn(ac, 'NKT', None, q, k, v, mask, e, gc, True, dtype, d)
    b = run_attn(ac, 'NKT', None, q, k, v, dense, e, gc, True, dtype, d)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    pq, pk, pv, pg, pe = [prep(x, dtype) for x in (q, k, v, gc, e)]
    exp = ref.multi_head_dot_attn_backward(pg, pq, pk, pv, dense.cpu().numpy(), pe, scaled=True,
                                           query_head_units=d)
    for name, x, y in zip(('dq', 'dk', 'dv', 'de'), a[1:], exp):
        assert rel_err(x, y) < tol(dtype), name


@pytest.mark.parametrize('attn_type', ['full', 'causal'])
@pytest.mark.parametrize('B,H,L,vl', [
    (2, 2, 200, (200, 77)),      # predicate-free blocks, then the ragged last ones
    (3, 1, 19, (19, 1, 7)),      # less than one 32-block
    (2, 3, 129, (128, 129)),     # one key past a workgroup of 128
    (1, 2, 320, (0,)),           # an empty sequence: every gradient is zero
])
def test_attention_backward_bf16_kernels_without_bias(nlp, attn_type, B, H, L, vl):
    """bf16, d = 64, no bias, mask from gen_self_attn_mask: the bf16-MFMA backward kernels
    (attn_bwd_bf16.hip) on their in-kernel predicate, including the path that drops the
    per-element predicates for blocks that are switched on entirely.  Against the float64 oracle,
    against the same bytes handed over untagged (dense-mask path, bit for bit) and against the
    general fp32-MFMA kernels (GNLP_ATTN_BWD_BF16=0).

    Tolerance: these kernels round P and dS to bf16 for the second matrix products (as the forward
    kernel rounds P).  Measured with tests/dev_bwd_err.py on the first case: rms error 0.0023 of
    rms(ref) against 0.0017 from rounding the exact gradient to bf16 alone (what the fp32-MFMA
    path shows), but heavy-tailed: early keys under the causal mask carry a few large dS terms
    that cancel in the sum, and one element of dk reaches 2.02e-2 on the element-wise metric.
    The bar of the bf16-MFMA path is therefore max(2e-2, 2 x the floor MEASURED on the case by
    tests/util.py attn_backward_floor with the same roundings -- 2.02e-2 on the causal case: the
    plain evaluation shows the same heavy tail), and never more than 3e-2; the fp32-MFMA path is held to the 2e-2 of every other bf16 test."""
    from gluonnlp_amd import _lib
    ac = nlp.attention_cell
    d = 64
    rng = np.random.default_rng(B * 100 + L)
    q, k, v = [rng.standard_normal((B, H, L, d)) for _ in range(3)]
    gc = rng.standard_normal((B, L, H * d))
    mask = ac.gen_self_attn_mask(torch.zeros((B, L, 1), device='cuda'),
                                 torch.tensor(vl, device='cuda'), attn_type=attn_type)
    dense = mask.clone()
    dtype = torch.bfloat16
    a = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, mask, None, gc, True, dtype, d)
    b = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, dense, None, gc, True, dtype, d)
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    # nothing but ctx wanted: the fused forward kernel leaves its row statistics and the backward
    # skips its statistics pass (D = <dO, O> from the bf16 output)
    f = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, mask, None, gc, True, dtype, d,
                 _need_scores=False, _need_weights=False)
    try:
        _lib.set_knob('GNLP_ATTN_BWD_BF16', '0')
        c = run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, mask, None, gc, True, dtype, d)
    finally:
        _lib.set_knob('GNLP_ATTN_BWD_BF16', None)
    pq, pk, pv, pg = [prep(x, dtype) for x in (q, k, v, gc)]
    exp = ref.multi_head_dot_attn_backward(pg, pq, pk, pv, dense.cpu().numpy(), None, scaled=True,
                                           query_head_units=d)
    live = float(max(np.abs(y).max() for y in exp[:3])) > 0.0
    tb = tol(dtype)
    if live:
        fl = attn_backward_floor(pg, pq, pk, pv, dense.cpu().numpy(), None, True, 'bf16')
        tb = min(bar(tb, fl[1:4], exp[:3]), 3e-2)   # the causal case's floor itself is 2.02e-2
    for name, x, y, z in zip(('dq', 'dk', 'dv'), a[1:4], exp[:3], c[1:4]):
        if float(np.abs(y).max()) == 0.0:
            assert float(np.abs(x).max()) == 0.0 and float(np.abs(z).max()) == 0.0, name
            continue
        assert rel_err(x, y) < tb, name
        assert rel_err(z, y) < tol(dtype), name
        assert rel_err(x, z) < tb, name
    for name, x, y in zip(('dq', 'dk', 'dv'), f[1:4], exp[:3]):
        if float(np.abs(y).max()) == 0.0:
            assert float(np.abs(x).max()) == 0.0, name
        else:
            assert rel_err(x, y) < tb, name


@pytest.mark.parametrize('layout,perm', LAYOUTS)
def test_attention_backward_with_forward_statistics(nlp, layout, perm):
    """gnlp_multi_head_dot_attn_lse + gnlp_attn_grad_desc.fwd_ctx / fwd_lse (bf16, d = 64, nothing
    but ctx wanted): dense mask with a fully masked row, bias shared over the batch with its
    gradient, cross attention with ragged tiles -- against the float64 oracle; and the C ABI
    refuses one saved tensor without the other."""
    from gluonnlp_amd import _lib
    ac = nlp.attention_cell
    B, H, Lq, Lm, d = 2, 3, 70, 161, 64
    rng = np.random.default_rng(5)
    q = rng.standard_normal((B, H, Lq, d))
    k, v = rng.standard_normal((B, H, Lm, d)), rng.standard_normal((B, H, Lm, d))
    gc = rng.standard_normal((B, Lq, H * d))
    e = rng.standard_normal((1, H, Lq, Lm))
    mask = rng.integers(0, 4, size=(B, Lq, Lm)) > 0
    mask[1, 17] = False
    dtype = torch.bfloat16
    pq, pk, pv, pg, pe = [prep(x, dtype) for x in (q, k, v, gc, e)]
    exp = ref.multi_head_dot_attn_backward(pg, pq, pk, pv, mask, pe, scaled=True,
                                           query_head_units=d)
    got = run_attn(ac, layout, perm, q, k, v, torch.tensor(mask, device='cuda'), e, gc, True,
                   dtype, d, _need_scores=False, _need_weights=False)
    assert _lib.load().gnlp_attn_last_kernel() == _lib.ATTN_KERNEL_FUSED_BF16
    tb = min(bar(tol(dtype), attn_backward_floor(pg, pq, pk, pv, mask, pe, True, 'bf16')[1:], exp),
             3e-2)
    for name, x, y in zip(('dq', 'dk', 'dv', 'de'), got[1:], exp):
        assert x.shape == y.shape
        assert rel_err(x, y) < tb, (layout, name)
    # the saved tensors come together
    lib = _lib.load()
    dd = _lib.AttnDesc()
    dd.B, dd.H, dd.Lq, dd.Lm, dd.dk, dd.dv = 1, 1, 4, 4, 64, 64
    gd = _lib.AttnGradDesc()
    x = torch.zeros(4 * 64, device='cuda', dtype=torch.bfloat16)
    gd.fwd_ctx = x.data_ptr()
    ws = torch.zeros(1024, dtype=torch.uint8, device='cuda')
    rc = lib.gnlp_multi_head_dot_attn_backward(
        x.data_ptr(), x.data_ptr(), x.data_ptr(), None, 0, None, x.data_ptr(), x.data_ptr(),
        x.data_ptr(), x.data_ptr(), ctypes.byref(dd), ctypes.byref(gd), _lib.BF16, None, None, None,
        0, None, None, 8.0, ws.data_ptr(), 1024, None)
    assert rc == -1 and b'fwd_lse' in lib.gnlp_last_error()
    # no other kernel writes the statistics: float32 is refused, nothing launched
    xf = torch.zeros(4 * 64, device='cuda')
    rc = lib.gnlp_multi_head_dot_attn_lse(xf.data_ptr(), xf.data_ptr(), xf.data_ptr(), xf.data_ptr(),
                                          xf.data_ptr(), xf.data_ptr(), ctypes.byref(dd), _lib.F32,
                                          None, None, None, 0, None, None, 8.0, None)
    assert rc == -2
