"""Shared helpers for the parity tests (GPU side compares the HIP path with oracle/ref.py)."""
import numpy as np
import torch


def bf16_round(a):
    """Round a float array to bf16 precision (returned as float64) -- what the device sees."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16)
    return t.to(torch.float64).numpy()


def to_dev(a, dtype=torch.float32, dev='cuda'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dtype)


def to_np(t):
    return t.detach().to(torch.float64).cpu().numpy()


class Err(float):
    """A float (so ``rel_err(a, b) < tol`` reads as before) whose repr -- what pytest prints when
    the assertion fails -- names both metrics and the worst element."""

    def __new__(cls, value, detail):
        o = super().__new__(cls, value)
        o.detail = detail
        return o

    def __repr__(self):
        return '{:.3e} [{}]'.format(float(self), self.detail)


def rel_err(a, b):
    """The error that ``< tol`` is asserted on: the LARGER of
      * the global figure   max |a - b| / max |b|, and
      * the element-wise    max_i |a_i - b_i| / (|b_i| + rms(b)),
    so ``rel_err(a, b) < tol`` means |delta| <= tol * max|ref| everywhere AND, element by element,
    |delta_i| <= tol * |ref_i| + tol * rms(ref): a small element beside a large one may not be
    wrong by more than ``tol`` of its own size plus ``tol`` of the tensor's typical size
    (north star: "within 1e-5 relative (fp32) or 2e-2 (bf16)")."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        raise AssertionError('shape mismatch {} vs {}'.format(a.shape, b.shape))
    if a.size == 0:
        return Err(0.0, 'empty')
    d = np.abs(a - b)
    s = np.abs(b).max()
    glob = d.max() / (s if s > 0 else 1.0)
    rms = float(np.sqrt(np.mean(b * b)))
    den = np.abs(b) + (rms if rms > 0 else 1.0)
    ratio = d / den
    i = int(np.argmax(ratio))
    elem = float(ratio.flat[i])
    idx = np.unravel_index(i, a.shape)
    detail = ('global max|d|/max|ref| = {:.3e}; element-wise max |d|/(|ref|+rms) = {:.3e} at {} '
              '(got {:.6g}, ref {:.6g}, rms(ref) {:.3g})').format(
                  glob, elem, tuple(int(x) for x in idx), a.flat[i], b.flat[i], rms)
    return Err(max(glob, elem), detail)


FP32_TOL = 1e-5     # north star: within 1e-5 relative (fp32)
BF16_TOL = 2e-2     # north star: within 2e-2 (bf16)


def tol(dtype):
    return FP32_TOL if dtype == torch.float32 else BF16_TOL


def prep(a, dtype):
    """Oracle-side view of an input that the device holds in ``dtype``."""
    a = np.asarray(a)
    if dtype == torch.bfloat16:
        return bf16_round(a)
    return a.astype(np.float32).astype(np.float64)


def load_params(module, params, dtype, dev='cuda'):
    """Copy a {reference-name: ndarray} dict into a gluonnlp_amd module (names are identical)."""
    module.to(dev)
    own = dict(module.named_parameters())
    for k, v in params.items():
        if k not in own:
            raise KeyError('module has no parameter {} (has {})'.format(k, sorted(own)[:5]))
        own[k].data = to_dev(v, dtype, dev)
    for k, p in own.items():
        if k not in params:
            p.data = p.data.to(dev).to(dtype)
    return module


def oracle_params(params, dtype):
    return {k: prep(v, dtype) for k, v in params.items()}


def attn_backward_floor(g_ctx, q, k, v, mask, e, scaled, mode):
    """A plain evaluation of multi_head_dot_attn and its gradients (NKT layout, oracle/ref.py's
    formulas) at a stated working precision -- NOT the oracle, but the measured FLOOR a widened
    tolerance is held against (the error any evaluation at that precision shows on the case):
      mode 'f32'  : every array and every product in float32 (numpy float32 matmul);
      mode 'bf16' : float64 arithmetic on bf16 inputs with the roundings of a bf16-MFMA backward:
                    P and dS rounded to bf16 before the second matrix products, outputs to bf16.
    -> ctx (B, Lq, H*dv), dq, dk, dv, de (shape of e, or None)."""
    ft = np.float32 if mode == 'f32' else np.float64
    rnd = (lambda a: a) if mode == 'f32' else bf16_round
    q, k, v = [np.asarray(a).astype(ft) for a in (q, k, v)]
    B, H, Lq, d = q.shape
    dvu = v.shape[-1]
    g = np.asarray(g_ctx).astype(ft).reshape(B, Lq, H, dvu).transpose(0, 2, 1, 3)
    scale = ft(np.sqrt(ft(d))) if scaled else ft(1)
    s = np.matmul(q, np.swapaxes(k, -1, -2))
    if e is not None:
        s = s + np.asarray(e).astype(ft)
    m = np.broadcast_to(np.asarray(mask).astype(bool)[:, None], s.shape)
    s = np.where(m, s / scale, ft(-np.inf))
    mx = s.max(-1, keepdims=True)
    mx = np.where(np.isfinite(mx), mx, ft(0))
    p = np.where(m, np.exp(s - mx), ft(0))
    den = p.sum(-1, keepdims=True)
    p = (p / np.where(den > 0, den, ft(1))).astype(ft)
    ctx = np.matmul(rnd(p), v).transpose(0, 2, 1, 3).reshape(B, Lq, H * dvu)
    d_v = np.matmul(np.swapaxes(rnd(p), -1, -2), g)
    d_p = np.matmul(g, np.swapaxes(v, -1, -2))
    d_s = (p * (d_p - (d_p * p).sum(-1, keepdims=True)) / scale).astype(ft)
    d_q = np.matmul(rnd(d_s), k)
    d_k = np.matmul(np.swapaxes(rnd(d_s), -1, -2), q)
    d_e = None
    if e is not None:
        es = np.asarray(e).shape
        full = (1,) * (4 - len(es)) + tuple(es)
        d_e = d_s
        for ax, n in enumerate(full):
            if n == 1 and d_e.shape[ax] != 1:
                d_e = d_e.sum(axis=ax, keepdims=True)
        d_e = d_e.reshape(es)
    return [rnd(np.asarray(a, dtype=np.float64)) if a is not None else None
            for a in (ctx, d_q, d_k, d_v, d_e)]
