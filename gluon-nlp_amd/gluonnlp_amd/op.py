"""Functional ops on the encoder path -- mirrors ``gluonnlp/op.py`` (reference
src/gluonnlp/op.py:229-307)."""
import torch

from . import _lib
from . import _tensor as T
from .autograd import L2NormalizeFn, wants_grad

__all__ = ['l2_normalize', 'relative_position_bucket']


def l2_normalize(data, axis=-1, eps=1e-6):
    """x / (||x||_2 + eps) along ``axis`` (op.py:289-307; eps is added to the norm).
    Differentiable under torch autograd (gnlp_l2_normalize_backward)."""
    x = T.as_tensor(data)
    dev = T.require_device(x)
    nd = x.dim()
    axis = axis % nd
    xt = x.movedim(axis, -1).contiguous() if axis != nd - 1 else x.contiguous()

    def call(xt):
        out = torch.empty_like(xt)
        d = xt.shape[-1] if nd > 0 else 1
        rows = xt.numel() // max(d, 1)
        _lib.check(_lib.load().gnlp_l2_normalize(T.ptr(out), T.ptr(xt), T.code(xt), rows, d,
                                                 float(eps), T.stream(dev)))
        return out

    out = L2NormalizeFn.apply(xt, eps, call) if wants_grad(xt) else call(xt)
    if axis != nd - 1:
        out = out.movedim(-1, axis)
    return T.like_caller(out, data)


def relative_position_bucket(relative_position, bidirectional: bool = True,
                             num_buckets: int = 32, max_distance: int = 128):
    """Map relative positions (mem_i - query_j, int32) to T5 buckets, int32 (op.py:229-286).
    Bit-exact with the float32-log formulation of the reference."""
    if bidirectional:
        assert num_buckets % 2 == 0, 'When bidirectional is True, the number of buckets must be ' \
                                     'divisible by 2.'
    rp = T.as_tensor(relative_position)
    dev = T.require_device(rp)
    rp32 = rp.to(torch.int32).contiguous()
    out = torch.empty_like(rp32)
    _lib.check(_lib.load().gnlp_relative_position_bucket(
        T.ptr(out), T.ptr(rp32), rp32.numel(), int(bool(bidirectional)), int(num_buckets),
        int(max_distance), T.stream(dev)))
    return T.like_caller(out, relative_position)
