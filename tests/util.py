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
