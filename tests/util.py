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


def rel_err(a, b):
    """max |a - b| / max |b|  (the 'relative' of the 1e-5 / 2e-2 tolerances)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b).max() if a.size else 0.0
    s = np.abs(b).max() if b.size else 1.0
    return d / (s if s > 0 else 1.0)


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
