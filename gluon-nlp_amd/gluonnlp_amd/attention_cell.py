"""Attention primitives -- same names, argument order, defaults and error behaviour as
``gluonnlp/attention_cell.py`` (reference src/gluonnlp/attention_cell.py:19-22), computed by the
gfx950 HIP kernels behind the C ABI (include/gluonnlp_amd.h)."""
import ctypes
import math
from typing import Optional

import torch

from . import _lib
from . import _tensor as T
from .autograd import AttnFn, RelScoreFn, wants_grad
from .op import l2_normalize

__all__ = ['gen_self_attn_mask', 'gen_mem_attn_mask', 'masked_softmax', 'masked_logsoftmax',
           'multi_head_dot_attn', 'MultiHeadAttentionCell', 'gen_rel_position',
           'RelAttentionScoreCell']


def _axes(layout):
    if layout == 'NT':
        return 0, 1
    if layout == 'TN':
        return 1, 0
    raise NotImplementedError('Unsupported layout={}'.format(layout))


def _tag(mask, vl, kind):
    """Remember how a mask was generated so the layers can evaluate the predicate in-kernel
    instead of re-reading (B, L, L) bytes per head.  The tag holds a PRIVATE copy of
    ``valid_length`` (B integers): the mask bytes alone define attention in the reference
    (models/transformer.py:234-265), so an in-place edit of the caller's ``valid_length`` after the
    mask was built must not change what the layers compute.  An in-place edit of the mask itself
    invalidates the tag (version counter)."""
    mask._gnlp_vl = vl.clone() if vl is not None else None
    mask._gnlp_kind = kind
    mask._gnlp_version = mask._version
    return mask


def mask_predicate(mask, causal=False, mem=False):
    """-> (kind, valid_length) if ``mask`` is an unmodified product of gen_self_attn_mask (or,
    for callers that pass ``mem=True``, of gen_mem_attn_mask), else None.  kind: 'full'
    ((i < vl) & (j < vl)), 'ones' (no valid_length), -- only for callers that pass ``causal=True``
    because they can evaluate it -- 'causal' ((j <= i) & (i < vl), valid_length possibly None), and
    'mem' ((j < vl[b]) & (i < vl[B + b]): 2 B lengths, keys then queries); every other caller sees
    an untagged mask and reads its bytes."""
    kind = getattr(mask, '_gnlp_kind', None)
    if kind is None or getattr(mask, '_gnlp_version', -1) != mask._version:
        return None
    if (kind == 'causal' and not causal) or (kind == 'mem' and not mem):
        return None
    return kind, getattr(mask, '_gnlp_vl', None)


def gen_self_attn_mask(data, valid_length=None, dtype=None, attn_type: str = 'full',
                       layout: str = 'NT'):
    """Self-attention mask, bool (batch_size, seq_length, seq_length); 1 = attend.
    attention_cell.py:40-183.  Only ``data.shape`` is read; ``dtype`` is accepted for signature
    parity (the compare is done on integers, which is what the float compare computes exactly)."""
    batch_axis, time_axis = _axes(layout)
    if attn_type not in ('full', 'causal'):
        raise NotImplementedError
    data = T.as_tensor(data)
    B, L = data.shape[batch_axis], data.shape[time_axis]
    vlt, vlp, vlc = T.valid_length_arg(valid_length)
    dev = T.require_device(data, vlt)
    mask = torch.empty((B, L, L), dtype=torch.bool, device=dev)
    _lib.check(_lib.load().gnlp_gen_self_attn_mask(
        T.ptr(mask), vlp, vlc, B, L, _lib.ATTN_FULL if attn_type == 'full' else _lib.ATTN_CAUSAL,
        T.stream(dev)))
    if attn_type == 'full':
        _tag(mask, vlt, 'full' if vlt is not None else 'ones')
    else:
        _tag(mask, vlt, 'causal')
    return mask


def gen_mem_attn_mask(mem, mem_valid_length, data, data_valid_length=None, dtype=None,
                      layout: str = 'NT'):
    """Decoder memory mask, bool (batch_size, query_length, mem_length).
    attention_cell.py:186-274."""
    batch_axis, time_axis = _axes(layout)
    mem, data = T.as_tensor(mem), T.as_tensor(data)
    B, Lm, Lq = mem.shape[batch_axis], mem.shape[time_axis], data.shape[time_axis]
    mvl, mp, mc = T.valid_length_arg(mem_valid_length)
    dvl, dp, dc = T.valid_length_arg(data_valid_length)
    if dvl is not None and dvl.dtype != mvl.dtype:
        dvl = dvl.to(mvl.dtype)
        dp = dvl.data_ptr()
    dev = T.require_device(mem, data, mvl, dvl)
    mask = torch.empty((B, Lq, Lm), dtype=torch.bool, device=dev)
    _lib.check(_lib.load().gnlp_gen_mem_attn_mask(T.ptr(mask), mp, dp, mc, B, Lq, Lm,
                                                  T.stream(dev)))
    # the fused attention kernel can evaluate this mask from the two length vectors
    if mvl is not None:
        lengths = torch.cat([mvl.reshape(-1), dvl.reshape(-1) if dvl is not None
                             else torch.full_like(mvl.reshape(-1), Lq)])
        _tag(mask, None, 'mem')
        mask._gnlp_vl = lengths      # (already a private tensor)
    return mask


def _softmax_call(att_score, mask, axis, temperature, log_mode):
    x = T.as_tensor(att_score)
    m = T.as_tensor(mask)
    dev = T.require_device(x, m)
    nd = x.dim()
    axis = axis % nd
    xt = x.movedim(axis, -1) if axis != nd - 1 else x
    xt = xt.contiguous()
    if m is not None:
        if m.dtype != torch.bool:
            m = m != 0                      # mask.astype(np.bool)  attention_cell.py:301
        if axis != nd - 1:
            while m.dim() < nd:
                m = m.unsqueeze(0)
            m = m.movedim(axis, -1)
        m = torch.broadcast_to(m, xt.shape)  # a view: broadcast dims get stride 0
    shape4 = list(xt.shape)
    if len(shape4) > 4:                      # fold leading dims (needs a materialised mask)
        lead = 1
        for s in shape4[:-3]:
            lead *= s
        if m is not None:
            m = m.contiguous().reshape([lead] + shape4[-3:])
        shape4 = [lead] + shape4[-3:]
    ms = list(m.stride()) if m is not None else [0, 0, 0, 0]
    while len(shape4) < 4:
        shape4.insert(0, 1)
        ms.insert(0, 0)
    out = torch.empty_like(xt)
    t = 0.0 if temperature is None else float(temperature)
    _lib.check(_lib.load().gnlp_masked_softmax(
        T.ptr(out), T.ptr(xt), T.code(xt), T.ptr(m), (ctypes.c_int64 * 4)(*ms),
        (ctypes.c_int64 * 4)(*shape4), t, int(log_mode), T.stream(dev)))
    if axis != nd - 1:
        out = out.movedim(-1, axis)
    return T.like_caller(out, att_score)


def masked_softmax(att_score, mask, axis: int = -1, temperature=None):
    """softmax(where(mask, x, -inf) / temperature) * mask; the mask may be broadcastable; a fully
    masked row gives zeros.  attention_cell.py:277-302."""
    return _softmax_call(att_score, mask, axis, temperature, 0)


def masked_logsoftmax(att_score, mask, axis: int = -1):
    """Masked log-softmax; masked positions are -inf.  attention_cell.py:305-329."""
    return _softmax_call(att_score, mask, axis, None, 1)


_LAYOUT_PERM = {'NKT': (0, 1, 2), 'NTK': (0, 2, 1), 'TNK': (1, 2, 0)}


def _bht_strides(x, layout):
    """Element strides of a (..., d) tensor for the logical axes (batch, head, time)."""
    st = x.stride()
    p = _LAYOUT_PERM[layout]
    return [st[p[0]], st[p[1]], st[p[2]]]


def _bht_shape(x, layout):
    p = _LAYOUT_PERM[layout]
    return x.shape[p[0]], x.shape[p[1]], x.shape[p[2]]


def _inner_contig(x):
    return x if (x.dim() == 4 and (x.stride(-1) == 1 or x.shape[-1] == 1)) else x.contiguous()


def multi_head_dot_attn(query, key, value, mask=None, edge_scores=None, dropout: float = 0.0,
                        scaled: bool = True, normalized: bool = False, eps: float = 1E-6,
                        query_head_units: Optional[int] = None, layout: str = 'NKT',
                        use_einsum: bool = False, *, _need_scores: bool = True,
                        _need_weights: bool = True, _valid_length=None):
    """Multi-head dot-product attention.  attention_cell.py:332-524.

    Returns ``(context_vec, [scores, attn_weights])`` like the reference.  The layouts are passed
    to the kernels as strides (no transposes, ``use_einsum`` is irrelevant).  Callers that discard
    ``scores`` / ``attn_weights`` (the encoder layers) pass ``_need_scores=False`` /
    ``_need_weights=False``: the list then holds ``None`` and, for bf16 with 64-wide heads, the
    fused kernel runs without ever forming the (B, H, Lq, Lm) tensor.  Dropout is inference-only
    (identity); a non-zero rate is accepted and ignored like ``npx.dropout`` outside autograd.

    Gradients: when torch autograd is recording and query / key / value / edge_scores require a
    gradient, ``context_vec.backward()`` fills them through the HIP backward kernels
    (gluonnlp_amd/autograd.py); ``scores`` / ``attn_weights`` are returned detached."""
    if normalized:
        query = l2_normalize(query, axis=-1, eps=eps)      # :435-437
        key = l2_normalize(key, axis=-1, eps=eps)
    if scaled:
        if query_head_units is None:
            raise NotImplementedError('You will need to specify query_head_units!')
        scale = math.sqrt(query_head_units)
    else:
        scale = None
    if layout not in _LAYOUT_PERM:
        raise NotImplementedError('layout="{}" is not supported! '
                                  'We only support layout = "NKT", "NTK", and "TNK".'
                                  .format(layout))
    q, k, v = [_inner_contig(T.as_tensor(t)) for t in (query, key, value)]
    m = T.as_tensor(mask)
    e = T.as_tensor(edge_scores)
    vlt, vlp, vlc = T.valid_length_arg(_valid_length)
    dev = T.require_device(q, k, v, m, e, vlt)
    if not (q.dtype == k.dtype == v.dtype):
        raise TypeError('query / key / value must share a dtype')
    B, H, Lq = _bht_shape(q, layout)
    _, _, Lm = _bht_shape(k, layout)
    dk, dv = q.shape[-1], v.shape[-1]

    def desc(q, k, v):
        d = _lib.AttnDesc()
        d.B, d.H, d.Lq, d.Lm, d.dk, d.dv = B, H, Lq, Lm, dk, dv
        d.causal = int(causal)
        d.mem_lengths = int(mem)
        d.q_stride[:] = _bht_strides(q, layout)
        d.k_stride[:] = _bht_strides(k, layout)
        d.v_stride[:] = _bht_strides(v, layout)
        if layout == 'TNK':
            d.o_stride[:] = [H * dv, dv, B * H * dv]
        else:
            d.o_stride[:] = [Lq * H * dv, dv, H * dv]
        return d

    mptr, ms = None, None
    causal = False
    mem = False            # gen_mem_attn_mask's predicate: valid_length = B key + B query lengths
    if m is not None:
        # (the causal and memory predicates live in the kernels for head dims <= 128)
        small = dk <= 128 and dv <= 128
        pred = mask_predicate(m, causal=small, mem=small) if _valid_length is None else None
        if pred is not None and m.shape == (B, Lq, Lm) and (Lq == Lm or pred[0] == 'mem'):
            # the mask came straight from gen_self_attn_mask / gen_mem_attn_mask: evaluate the
            # predicate in-kernel
            kind, pvl = pred
            if kind in ('full', 'causal', 'mem') and pvl is not None:
                vlt, vlp, vlc = T.valid_length_arg(pvl)
            causal = kind == 'causal'
            mem = kind == 'mem'
            m = None
        else:
            if m.dtype != torch.bool:
                m = m != 0                                  # :449
            m = torch.broadcast_to(m, (B, Lq, Lm))
            mptr, ms = T.ptr(m), (ctypes.c_int64 * 3)(*m.stride())
            vlt, vlp = None, None
    if e is not None and e.dtype != q.dtype:
        e = e.to(q.dtype)
    temperature = 0.0 if scale is None else float(scale)

    def call(q, k, v, e, want_lse=False):
        """-> ctx, scores, weights[, saved]: with want_lse (a backward pass will follow) the fused
        bf16 kernel also leaves (lse, ctx_lo) = its row statistics and the residual of ctx's
        rounding (None when another kernel serves the call)."""
        d = desc(q, k, v)
        ctx = torch.empty((Lq, B, H * dv) if layout == 'TNK' else (B, Lq, H * dv), dtype=q.dtype,
                          device=dev)
        eptr, es = None, None
        if e is not None:
            while e.dim() < 4:
                e = e.unsqueeze(0)
            e = torch.broadcast_to(e, (B, H, Lq, Lm))
            eptr, es = T.ptr(e), (ctypes.c_int64 * 4)(*e.stride())
        scores = torch.empty((B, H, Lq, Lm), dtype=q.dtype, device=dev) if _need_scores else None
        weights = torch.empty((B, H, Lq, Lm), dtype=q.dtype, device=dev) if _need_weights else None
        lib = _lib.load()
        if want_lse:
            # (scaled attention only: with raw scores -- T5 -- the softmax is peaked, dP - D cancels
            # hard, and D = <dO, ctx> inherits the bf16 rounding of P inside ctx: measured tails of
            # 5-10 % on d(edge) / dq against 3 % when D is recomputed; tools/attn_bwd_fuzz.py)
            if (not _need_scores and not _need_weights and q.dtype == torch.bfloat16
                    and dk == 64 and dv == 64 and B * H * Lq * Lm > 0 and temperature > 1.0):
                lse = torch.empty((B, H, Lq), dtype=torch.float32, device=dev)
                ctx_lo = torch.empty_like(ctx)
                rc = lib.gnlp_multi_head_dot_attn_lse(
                    T.ptr(ctx), T.ptr(ctx_lo), T.ptr(lse), T.ptr(q), T.ptr(k), T.ptr(v),
                    ctypes.byref(d), T.code(q), mptr, ms, vlp, vlc, eptr, es, temperature,
                    T.stream(dev))
                if rc == 0:
                    return ctx, None, None, (lse, ctx_lo)
            return call(q, k, v, e) + (None,)
        nws = lib.gnlp_attn_workspace_bytes(ctypes.byref(d), T.code(q), int(_need_scores),
                                            int(_need_weights))
        fused_ok = (not _need_scores and not _need_weights and q.dtype == torch.bfloat16
                    and dk == 64 and dv == 64)
        ws = None if fused_ok else T.workspace(dev, nws)
        rc = lib.gnlp_multi_head_dot_attn(
            T.ptr(ctx), T.ptr(scores), T.ptr(weights), T.ptr(q), T.ptr(k), T.ptr(v),
            ctypes.byref(d), T.code(q), mptr, ms, vlp, vlc, eptr, es, temperature, T.ptr(ws),
            0 if ws is None else ws.numel(), T.stream(dev))
        if rc != 0 and fused_ok:        # fused kernel declined (alignment): retry with scratch
            ws = T.workspace(dev, nws)
            rc = lib.gnlp_multi_head_dot_attn(
                T.ptr(ctx), None, None, T.ptr(q), T.ptr(k), T.ptr(v), ctypes.byref(d), T.code(q),
                mptr, ms, vlp, vlc, eptr, es, temperature, T.ptr(ws), ws.numel(), T.stream(dev))
        _lib.check(rc)
        return ctx, scores, weights

    if wants_grad(q, k, v, e):
        # gradients for query / key / value / edge_scores (tests/test_attention_cell.py:82-151)
        ctx, scores, weights = AttnFn.apply(
            q, k, v, e, call, {'layout': layout, 'desc': desc, 'bht': _bht_strides, 'mask': m,
                               'vl': vlt, 'scale': temperature})
    else:
        ctx, scores, weights = call(q, k, v, e)
    return (T.like_caller(ctx, query),
            [T.like_caller(scores, query) if scores is not None else None,
             T.like_caller(weights, query) if weights is not None else None])


class MultiHeadAttentionCell(torch.nn.Module):
    """Stateless multi-head attention cell.  attention_cell.py:527-608."""

    def __init__(self, query_units=None, num_heads=None, attention_dropout=0.0,
                 scaled: bool = True, normalized: bool = False, eps: float = 1E-6,
                 dtype='float32', layout='NTK', use_einsum=False):
        super().__init__()
        self._query_units = query_units
        self._num_heads = num_heads
        self._attention_dropout = attention_dropout
        self._scaled = scaled
        self._normalized = normalized
        self._eps = eps
        self._dtype = dtype
        assert layout in ['NTK', 'NKT', 'TNK']
        self._layout = layout
        self._use_einsum = use_einsum
        if self._query_units is not None:
            assert self._num_heads is not None
            assert self._query_units % self._num_heads == 0, \
                'The units must be divisible by the number of heads.'
            self._query_head_units = self._query_units // self._num_heads
        else:
            self._query_head_units = None

    @property
    def layout(self):
        return self._layout

    def forward(self, query, key, value, mask=None, edge_scores=None, **kw):
        return multi_head_dot_attn(query=query, key=key, value=value, mask=mask,
                                   edge_scores=edge_scores, dropout=self._attention_dropout,
                                   scaled=self._scaled, normalized=self._normalized,
                                   eps=self._eps, query_head_units=self._query_head_units,
                                   layout=self._layout, use_einsum=self._use_einsum, **kw)

    def __repr__(self):
        return ('{}(\n   query_units={},\n   num_heads={},\n   attention_dropout={},\n'
                '   scaled={},\n   normalized={},\n   layout="{}",\n   use_einsum={},\n'
                '   dtype={}\n)').format(self.__class__.__name__, self._query_units,
                                         self._num_heads, self._attention_dropout, self._scaled,
                                         self._normalized, self._layout, self._use_einsum,
                                         self._dtype)


def gen_rel_position(data, past_data=None, dtype=None, layout='NT'):
    """(L, L) int32 matrix of ``mem_j - query_i``.  attention_cell.py:611-656.  Under incremental
    decoding (``past_data`` = the stacked previous steps, (B, L_past, H, d) / (L_past, B, H, d) or
    already merged to 3-D) the positions run over the concatenated sequence, L = L_past + L_cur
    (:644-652); only the time extents are read."""
    time_axis = 1 if layout == 'NT' else 0
    data = T.as_tensor(data)
    past = T.as_tensor(past_data)
    dev = T.require_device(data, past)
    L = data.shape[time_axis] + (past.shape[time_axis] if past is not None else 0)
    out = torch.empty((L, L), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().gnlp_gen_rel_position(T.ptr(out), L, T.stream(dev)))
    return out


class RelAttentionScoreCell(torch.nn.Module):
    """Relative-position score cell.  attention_cell.py:659-881: 't5' (bucketed bias, the encoder
    path of BASELINE config 4), 'transformer_xl' (<Q, W S_{i-j}>, sinusoidal S) and 'shaw'
    (<Q, E_{i-j}>, learned E)."""

    def __init__(self, query_units, num_heads, pos_embed_units: Optional[int] = None,
                 max_distance=None, bidirectional=False, num_buckets=None, method='transformer_xl',
                 dropout: float = 0.0, dtype='float32', layout='NTK', use_einsum=False,
                 embed_initializer=None):
        super().__init__()
        from .layers import (BucketPositionalEmbedding, Dense, LearnedPositionalEmbedding,
                             SinusoidalPositionalEmbedding)
        self._dropout = dropout
        self._method = method
        self._query_units = query_units
        self._num_heads = num_heads
        self._bidirectional = bidirectional
        self._num_buckets = num_buckets
        assert query_units % num_heads == 0, 'The units must be divisible by the number of heads.'
        self._head_query_units = query_units // num_heads
        self._max_distance = max_distance
        self._pos_embed_units = pos_embed_units
        self._dtype = dtype
        self._use_einsum = use_einsum
        self._layout = layout
        if self._layout not in ['NKT', 'NTK', 'TNK']:
            raise ValueError('layout="{}" is not supported'.format(self._layout))
        if method == 'transformer_xl':
            if pos_embed_units is None:
                pos_embed_units = self._num_heads * self._head_query_units
            self._rel_pos_embed = SinusoidalPositionalEmbedding(units=pos_embed_units,
                                                                dtype=self._dtype)
            self._rel_proj = Dense(units=query_units, in_units=pos_embed_units, flatten=False,
                                   use_bias=False, dtype=self._dtype)
        elif method == 'shaw':
            assert self._max_distance is not None, 'Must set max_distance when method="shaw".'
            if self._bidirectional:
                vocab_size = self._max_distance * 2 + 1
            else:
                vocab_size = self._max_distance + 1

            def xavier_in(t):        # Xavier(gaussian, factor_type='in', magnitude=1)
                t.normal_(0.0, (1.0 / t.shape[1]) ** 0.5)
            self._rel_pos_embed = LearnedPositionalEmbedding(
                units=self._num_heads * self._head_query_units, max_length=vocab_size,
                weight_initializer=xavier_in, mode='wrap' if self._bidirectional else 'raise',
                dtype=self._dtype)
        elif method == 't5':
            if self._num_buckets is None:
                self._num_buckets = 32
            if self._max_distance is None:
                self._max_distance = 128
            self._rel_pos_embed = BucketPositionalEmbedding(
                units=num_heads, num_buckets=self._num_buckets, max_distance=self._max_distance,
                bidirectional=self._bidirectional, embed_initializer=embed_initializer,
                dtype=self._dtype)
        else:
            raise NotImplementedError('method="{}" is currently not supported!'.format(method))

    @property
    def layout(self) -> str:
        return self._layout

    def self_attn_bias(self, Lq, Lm=None):
        """Fused gen_rel_position + bucket + gather + transpose: (H, Lq, Lm) directly."""
        Lm = Lq if Lm is None else Lm
        w = self._rel_pos_embed.weight.data
        dev = T.require_device(w)
        out = torch.empty((self._num_heads, Lq, Lm), dtype=w.dtype, device=dev)
        _lib.check(_lib.load().gnlp_t5_rel_bias(
            T.ptr(out), T.ptr(w), T.code(w), self._num_heads, Lq, Lm, int(self._bidirectional),
            self._num_buckets, self._max_distance, T.stream(dev)))
        return out

    def self_attn_bias_table(self, Lq, Lm=None):
        """The same bias as one row per head, (H, Lq + Lm - 1): bias[h, i, j] =
        table[h, (Lq - 1) + j - i] (the relative position of gen_rel_position is j - i only)."""
        Lm = Lq if Lm is None else Lm
        w = self._rel_pos_embed.weight.data
        dev = T.require_device(w)
        out = torch.empty((self._num_heads, Lq + Lm - 1), dtype=w.dtype, device=dev)
        _lib.check(_lib.load().gnlp_t5_rel_bias_table(
            T.ptr(out), T.ptr(w), T.code(w), self._num_heads, Lq, Lm, int(self._bidirectional),
            self._num_buckets, self._max_distance, T.stream(dev)))
        return out

    def forward(self, rel_positions, query=None):
        """rel_positions (Lq, Lm) -> 't5': (num_heads, Lq, Lm); 'transformer_xl' / 'shaw' (need
        ``query`` in the cell's layout): (batch, num_heads, Lq, Lm).  attention_cell.py:796-881."""
        if self._method == 't5':
            emb = self._rel_pos_embed(rel_positions)          # (Lq, Lm, H)   :876-878
            return emb.permute(2, 0, 1).contiguous()
        assert query is not None, 'Must specify query if method={}'.format(self._method)
        rel = T.as_tensor(rel_positions)
        q = _inner_contig(T.as_tensor(query))
        dev = T.require_device(rel, q)
        rel = rel.to(torch.int32).contiguous()
        Lq, Lm = rel.shape
        B, H, Lq2 = _bht_shape(q, self._layout)
        if Lq2 != Lq or H != self._num_heads or q.shape[-1] != self._head_query_units:
            raise ValueError('query {} does not match rel_positions {} / the cell ({} heads of {})'
                             .format(tuple(q.shape), tuple(rel.shape), self._num_heads,
                                     self._head_query_units))
        lo, hi = -2 ** 31, 2 ** 31 - 1
        if self._max_distance is not None:                    # :818-825
            lo, hi = (-self._max_distance if self._bidirectional else 0), self._max_distance
        if self._method == 'shaw':
            # the learned table itself: row = clipped position, negatives wrap (np.take 'wrap')
            table, base, wrap = self._rel_pos_embed.weight, 0, int(self._bidirectional)
        else:
            # 'transformer_xl': project the sinusoids of every position in the (clipped) range
            if self._max_distance is not None:
                p_lo, p_hi = lo, hi
            elif rel.numel():
                mn, mx = torch.aminmax(rel)
                p_lo, p_hi = int(mn), int(mx)
            else:
                p_lo = p_hi = 0
            positions = torch.arange(p_lo, p_hi + 1, dtype=torch.float32, device=dev)
            table = T.as_tensor(self._rel_proj(self._rel_pos_embed(positions)))   # :829-831
            base, wrap = p_lo, 0
        if table.dtype != q.dtype:
            raise TypeError('query dtype {} != position embedding dtype {}'.format(q.dtype,
                                                                                  table.dtype))
        qs = _bht_strides(q, self._layout)
        C, R = self._head_query_units, table.shape[0]

        def call(q, table):
            table = table.contiguous()
            out = torch.empty((B, H, Lq, Lm), dtype=q.dtype, device=dev)
            _lib.check(_lib.load().gnlp_rel_attn_score(
                T.ptr(out), T.ptr(q), (ctypes.c_int64 * 3)(*qs), T.ptr(table), T.ptr(rel),
                T.code(q), B, H, Lq, Lm, C, R, lo, hi, base, wrap, T.stream(dev)))
            return out

        if wants_grad(q, table):
            # gradients for the query and the cell's parameters (through the table: the learned
            # 'shaw' embedding itself, or rel_proj's weight behind the sinusoids)
            out = RelScoreFn.apply(q, table.contiguous(), rel,
                                   (qs, B, H, Lq, Lm, C, lo, hi, base, wrap, self._layout), call)
        else:
            out = call(q, table)
        return T.like_caller(out, query)
