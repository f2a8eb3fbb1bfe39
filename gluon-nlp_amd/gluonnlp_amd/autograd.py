"""Backward pass of the attention / norm kernels, hooked into ``torch.autograd``.

The reference differentiates its ops with ``mx.autograd`` (``with mx.autograd.record(): ...;
out.backward()``, tests/test_attention_cell.py:82-89); callers of this package hold torch tensors,
so the same contract is ``requires_grad_()`` + ``out.backward()``.  torch is plumbing here: every
gradient is computed by a HIP kernel behind the C ABI (``gnlp_multi_head_dot_attn_backward``,
``gnlp_norm_backward``, ``gnlp_l2_normalize_backward``); the Functions below only save the
forward's inputs and hand device pointers over.  Nothing of size (B, H, Lq, Lm) is kept for the
backward: the softmax is recomputed from q, k, edge_scores and the mask.
"""
import ctypes

import torch

from . import _lib
from . import _tensor as T


def wants_grad(*tensors):
    """True when autograd is recording and one of the (torch) tensors asks for a gradient."""
    if not torch.is_grad_enabled():
        return False
    return any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _cast_from_f32(acc, dtype, dev):
    if dtype == torch.float32:
        return acc
    out = torch.empty(acc.shape, dtype=dtype, device=dev)
    _lib.check(_lib.load().gnlp_cast_from_f32(T.ptr(out), _lib.BF16, T.ptr(acc), acc.numel(),
                                              T.stream(dev)))
    return out


class AttnFn(torch.autograd.Function):
    """``multi_head_dot_attn`` with gradients for query / key / value / edge_scores.

    ``call`` is the forward closure of ``attention_cell.multi_head_dot_attn`` (it owns the layout,
    mask and valid_length handling); ``state`` carries what the backward needs to repeat the
    descriptor: the layout helpers, the prepared mask, the valid-length tensor and the scale."""

    @staticmethod
    def forward(ctx, q, k, v, e, call, state):
        out, scores, weights = call(q, k, v, e)
        ctx.save_for_backward(q, k, v, e if e is not None else torch.empty(0, device=q.device))
        ctx.has_e = e is not None
        ctx.state = state
        marks = [t for t in (scores, weights) if t is not None]
        if marks:
            ctx.mark_non_differentiable(*marks)
        return out, scores, weights

    @staticmethod
    def backward(ctx, g_out, _g_scores, _g_weights):
        q, k, v, e = ctx.saved_tensors
        e = e if ctx.has_e else None
        st = ctx.state
        dev = q.device
        layout = st['layout']
        d = st['desc'](q, k, v)                      # gnlp_attn_desc of the forward call
        B, H, Lq, Lm = d.B, d.H, d.Lq, d.Lm
        g = g_out.contiguous()
        gd = _lib.AttnGradDesc()
        hdv = H * d.dv
        gd.g_stride[:] = [hdv, d.dv, B * hdv] if layout == 'TNK' else [Lq * hdv, d.dv, hdv]
        dq, dk, dv = torch.empty_like(q, memory_format=torch.contiguous_format), \
            torch.empty_like(k, memory_format=torch.contiguous_format), \
            torch.empty_like(v, memory_format=torch.contiguous_format)
        gd.dq_stride[:] = st['bht'](dq, layout)
        gd.dk_stride[:] = st['bht'](dk, layout)
        gd.dv_stride[:] = st['bht'](dv, layout)
        if B * H * Lq * Lm * d.dk * d.dv == 0:       # an empty axis: every gradient is zero
            return (dq.zero_(), dk.zero_(), dv.zero_(),
                    torch.zeros_like(e) if e is not None and ctx.needs_input_grad[3] else None,
                    None, None)
        eptr = es = de = deptr = des = None
        de_n = 0
        if e is not None:
            e4 = e
            while e4.dim() < 4:
                e4 = e4.unsqueeze(0)
            eb = torch.broadcast_to(e4, (B, H, Lq, Lm))
            eptr, es = T.ptr(eb), (ctypes.c_int64 * 4)(*eb.stride())
            if ctx.needs_input_grad[3]:
                de = torch.empty(e4.shape, dtype=torch.float32, device=dev)   # zero-filled by the call
                des = (ctypes.c_int64 * 4)(*[0 if e4.shape[i] == 1 else de.stride(i)
                                             for i in range(4)])
                deptr, de_n = T.ptr(de), de.numel()
        m = st['mask']
        mptr, ms = (T.ptr(m), (ctypes.c_int64 * 3)(*m.stride())) if m is not None else (None, None)
        vlt = st['vl']
        vlp, vlc = (None, 0) if vlt is None or m is not None else \
            (vlt.data_ptr(), _lib.I64 if vlt.dtype == torch.int64 else _lib.I32)
        lib = _lib.load()
        ws = T.workspace(dev, lib.gnlp_attn_backward_workspace_bytes(ctypes.byref(d)))
        _lib.check(lib.gnlp_multi_head_dot_attn_backward(
            T.ptr(dq), T.ptr(dk), T.ptr(dv), deptr, de_n, des, T.ptr(g), T.ptr(q), T.ptr(k),
            T.ptr(v), ctypes.byref(d), ctypes.byref(gd), T.code(q), mptr, ms, vlp, vlc, eptr, es,
            st['scale'], T.ptr(ws), ws.numel(), T.stream(dev)))
        if de is not None:
            de = _cast_from_f32(de, e.dtype, dev).reshape(e.shape)
        return dq, dk, dv, de, None, None


class NormFn(torch.autograd.Function):
    """(residual +) LayerNorm / RMSNorm / NoNorm with gradients for data, residual, gamma, beta."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, kind, eps, call):
        out = call(x, residual, gamma, beta)
        ctx.save_for_backward(x, residual if residual is not None else torch.empty(0, device=x.device),
                              gamma)
        ctx.has_res = residual is not None
        ctx.kind, ctx.eps = kind, eps
        ctx.beta_dtype = beta.dtype if beta is not None else None
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, residual, gamma = ctx.saved_tensors
        residual = residual if ctx.has_res else None
        dev = x.device
        xc = x.contiguous()
        rc = residual.contiguous() if residual is not None else None
        g = g_out.contiguous()
        C = xc.shape[-1]
        rows = xc.numel() // max(C, 1)
        dx = torch.empty_like(xc)
        acc = torch.empty((2, C), dtype=torch.float32, device=dev)      # zero-filled by the call
        _lib.check(_lib.load().gnlp_norm_backward(
            T.ptr(dx), T.ptr(acc[0]), T.ptr(acc[1]), T.ptr(g), T.ptr(xc), T.ptr(rc), T.ptr(gamma),
            T.code(xc), ctx.kind, float(ctx.eps), rows, C, T.stream(dev)))
        dgamma = _cast_from_f32(acc[0].contiguous(), gamma.dtype, dev) \
            if ctx.needs_input_grad[2] else None
        dbeta = _cast_from_f32(acc[1].contiguous(), ctx.beta_dtype, dev) \
            if ctx.beta_dtype is not None and ctx.needs_input_grad[3] else None
        return (dx if ctx.needs_input_grad[0] else None,
                dx if residual is not None and ctx.needs_input_grad[1] else None,
                dgamma, dbeta, None, None, None)


class L2NormalizeFn(torch.autograd.Function):
    """``l2_normalize`` over the last axis of a contiguous tensor."""

    @staticmethod
    def forward(ctx, x, eps, call):
        ctx.save_for_backward(x)
        ctx.eps = eps
        return call(x)

    @staticmethod
    def backward(ctx, g_out):
        (x,) = ctx.saved_tensors
        dev = x.device
        g = g_out.contiguous()
        d = x.shape[-1] if x.dim() > 0 else 1
        dx = torch.empty_like(x)
        _lib.check(_lib.load().gnlp_l2_normalize_backward(
            T.ptr(dx), T.ptr(g), T.ptr(x), T.code(x), x.numel() // max(d, 1), d, float(ctx.eps),
            T.stream(dev)))
        return dx, None, None
