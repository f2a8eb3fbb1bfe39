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
        out, scores, weights, saved = call(q, k, v, e, want_lse=True)
        none = torch.empty(0, device=q.device)
        # with the fused kernel's row statistics the backward skips its own statistics pass
        ctx.save_for_backward(q, k, v, e if e is not None else none,
                              *((out, saved[1], saved[0]) if saved is not None else (none,) * 3))
        ctx.has_e = e is not None
        ctx.has_lse = saved is not None
        ctx.state = state
        marks = [t for t in (scores, weights) if t is not None]
        if marks:
            ctx.mark_non_differentiable(*marks)
        return out, scores, weights

    @staticmethod
    def backward(ctx, g_out, _g_scores, _g_weights):
        q, k, v, e, fwd_out, fwd_out_lo, fwd_lse = ctx.saved_tensors
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
        if ctx.has_lse:
            gd.fwd_ctx, gd.fwd_ctx_lo, gd.fwd_lse = \
                fwd_out.data_ptr(), fwd_out_lo.data_ptr(), fwd_lse.data_ptr()
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


class DenseFn(torch.autograd.Function):
    """Plain ``Dense`` (identity activation, no fused operand): y = x W^T + b.
    dx = g W and dW = g^T x are gnlp_dense calls on transposed operands (the transposes are torch
    copies: plumbing), db = column sums of g (gnlp_colsum)."""

    @staticmethod
    def forward(ctx, x, weight, bias, call):
        ctx.save_for_backward(x, weight)
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return call(x, weight, bias)

    @staticmethod
    def backward(ctx, g_out):
        x, w = ctx.saved_tensors
        N, K = w.shape
        dx, dw, db = _dense_grads(ctx.needs_input_grad, g_out.contiguous().reshape(-1, N),
                                  x.contiguous().reshape(-1, K), w, ctx.bias_dtype, x.device)
        if dx is not None:
            dx = dx.reshape(x.shape)
        return dx, dw, db, None


def _dense_grads(ctx_needs, g2, x2, w, bias_dtype, dev):
    """dx = g W, dW = g^T x (gnlp_dense on transposed operands; the transposes are torch copies:
    plumbing), db = column sums of g (gnlp_colsum) for y = x W^T + b with g = dL/dy."""
    lib = _lib.load()
    N, K = w.shape
    M = x2.shape[0]

    def mm(a, bt):     # a (m, k) @ bt (n, k)^T -> (m, n) through gnlp_dense
        m, k = a.shape
        n = bt.shape[0]
        out = torch.empty((m, n), dtype=a.dtype, device=dev)
        _lib.check(lib.gnlp_dense(T.ptr(out), T.ptr(a), T.ptr(bt), None, None,
                                  _lib.COMBINE_ADD, _lib.ACT['identity'], T.code(a), m, n, k,
                                  k, n, n, T.stream(dev)))
        return out
    dx = dw = db = None
    if ctx_needs[0]:
        dx = mm(g2, w.detach().t().contiguous())
    if ctx_needs[1]:
        dw = mm(g2.t().contiguous(), x2.t().contiguous())
    if bias_dtype is not None and ctx_needs[2]:
        acc = torch.empty((N,), dtype=torch.float32, device=dev)
        _lib.check(lib.gnlp_colsum(T.ptr(acc), T.ptr(g2), T.code(g2), M, N, T.stream(dev)))
        db = _cast_from_f32(acc, bias_dtype, dev)
    return dx, dw, db


class DenseEpilogueFn(torch.autograd.Function):
    """``Dense`` with its fused epilogue under autograd: y = combine(act(x W^T + b), other)
    (layers.py:640-642, models/transformer.py:266-268 differentiated op by op by mx.autograd).
    The fused GEMM epilogues keep no pre-activation, so the recording forward runs the plain GEMM,
    saves z = x W^T + b and applies the epilogue as an elementwise kernel (gnlp_dense_epilogue);
    the backward is gnlp_dense_epilogue_backward (dz, and d_other of a gate) followed by the plain
    Dense gradients.  For combine 'add' d_other is the cotangent itself."""

    @staticmethod
    def forward(ctx, x, weight, bias, other, act, combine, call_plain):
        z = call_plain(x, weight, bias)
        dev = z.device
        y = torch.empty_like(z)
        _lib.check(_lib.load().gnlp_dense_epilogue(
            T.ptr(y), T.ptr(z), T.ptr(other), T.code(z), z.numel(), act, combine, T.stream(dev)))
        ctx.save_for_backward(x, weight, z, other)
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.act, ctx.combine = act, combine
        return y

    @staticmethod
    def backward(ctx, g_out):
        x, w, z, other = ctx.saved_tensors
        dev = z.device
        lib = _lib.load()
        N, K = w.shape
        g = g_out.contiguous()
        mul = ctx.combine == _lib.COMBINE_MUL
        dz = torch.empty_like(z)
        d_other = None
        if other is not None and ctx.needs_input_grad[3]:
            d_other = torch.empty_like(z) if mul else g
        _lib.check(lib.gnlp_dense_epilogue_backward(
            T.ptr(dz), T.ptr(d_other) if mul else None, T.ptr(g), T.ptr(z),
            T.ptr(other) if mul else None, T.code(z), z.numel(), ctx.act, ctx.combine,
            T.stream(dev)))
        dx, dw, db = _dense_grads(ctx.needs_input_grad, dz.reshape(-1, N),
                                  x.contiguous().reshape(-1, K), w, ctx.bias_dtype, dev)
        if dx is not None:
            dx = dx.reshape(x.shape)
        return dx, dw, db, d_other, None, None, None


class ActFn(torch.autograd.Function):
    """get_activation(...) as a module of its own (layers.py:236-338) under autograd."""

    @staticmethod
    def forward(ctx, x, act, call):
        ctx.save_for_backward(x)
        ctx.act = act
        return call(x)

    @staticmethod
    def backward(ctx, g_out):
        x, = ctx.saved_tensors
        dx = torch.empty_like(x)
        _lib.check(_lib.load().gnlp_dense_epilogue_backward(
            T.ptr(dx), None, T.ptr(g_out.contiguous()), T.ptr(x), None, T.code(x), x.numel(),
            ctx.act, _lib.COMBINE_ADD, T.stream(x.device)))
        return dx, None, None


class SeqMaskFn(torch.autograd.Function):
    """npx.sequence_mask (models/bert.py:311-312) under autograd: the cotangent is masked alike."""

    @staticmethod
    def forward(ctx, x, call):
        ctx.call = call
        return call(x)

    @staticmethod
    def backward(ctx, g_out):
        return ctx.call(g_out.contiguous()), None


class BucketEmbedFn(torch.autograd.Function):
    """BucketPositionalEmbedding: weight[bucket(rel)]; the gradient scatters onto the table."""

    @staticmethod
    def forward(ctx, weight, rel32, cfg, call):
        ctx.rel32, ctx.cfg = rel32, cfg
        ctx.wshape, ctx.wdtype = weight.shape, weight.dtype
        return call(weight)

    @staticmethod
    def backward(ctx, g_out):
        dev = g_out.device
        g = g_out.contiguous()
        bidirectional, num_buckets, max_distance = ctx.cfg
        units = ctx.wshape[1]
        acc = torch.empty(ctx.wshape, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().gnlp_bucket_embedding_backward(
            T.ptr(acc), T.ptr(g), T.ptr(ctx.rel32), T.code(g), ctx.rel32.numel(), units,
            int(bidirectional), num_buckets, max_distance, T.stream(dev)))
        return _cast_from_f32(acc, ctx.wdtype, dev), None, None, None


class RelScoreFn(torch.autograd.Function):
    """RelAttentionScoreCell 'transformer_xl' / 'shaw': <q[b, h, i], table[idx(rel[i, j]), h]>
    with gradients for the query and the (R, H * C) table."""

    @staticmethod
    def forward(ctx, q, table, rel, args, call):
        ctx.save_for_backward(q, table)
        ctx.rel, ctx.args, ctx.args_layout = rel, args[:-1], args[-1]
        return call(q, table)

    @staticmethod
    def backward(ctx, g_out):
        q, table = ctx.saved_tensors
        dev = q.device
        qs, B, H, Lq, Lm, C, lo, hi, base, wrap = ctx.args
        g = g_out.contiguous()
        dq = torch.empty_like(q, memory_format=torch.contiguous_format) \
            if ctx.needs_input_grad[0] else None
        dqs = None
        if dq is not None:
            from .attention_cell import _bht_strides
            dqs = (ctypes.c_int64 * 3)(*_bht_strides(dq, ctx.args_layout))
        dt = torch.empty(table.shape, dtype=torch.float32, device=dev) \
            if ctx.needs_input_grad[1] else None
        _lib.check(_lib.load().gnlp_rel_attn_score_backward(
            T.ptr(dq), dqs, T.ptr(dt), T.ptr(g), T.ptr(q), (ctypes.c_int64 * 3)(*qs),
            T.ptr(table), T.ptr(ctx.rel), T.code(q), B, H, Lq, Lm, C, table.shape[0], lo, hi,
            base, wrap, T.stream(dev)))
        if dt is not None:
            dt = _cast_from_f32(dt, table.dtype, dev)
        return dq, dt, None, None, None
