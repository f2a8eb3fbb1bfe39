"""Reusable blocks of the encoder path -- mirrors ``gluonnlp/layers.py`` (reference
src/gluonnlp/layers.py): get_norm_layer / NoNorm / RMSNorm, get_activation / GELU,
BucketPositionalEmbedding, PositionwiseFFN; plus the two MXNet layers the path uses
(``nn.Dense(flatten=False)`` and ``nn.LayerNorm``) with MXNet's parameter names
(weight / bias, gamma / beta) so reference state dicts load unchanged."""
import ctypes
import math
from collections import OrderedDict

import torch
from torch import nn

from . import _lib
from . import _tensor as T
from .autograd import ActFn, BucketEmbedFn, DenseEpilogueFn, DenseFn, NormFn, wants_grad

__all__ = ['PositionwiseFFN', 'GELU', 'get_activation', 'get_norm_layer', 'RMSNorm', 'NoNorm',
           'LayerNorm', 'Dense', 'BucketPositionalEmbedding', 'IdentityActivation', 'Activation',
           'Embedding', 'PositionalEmbedding', 'LearnedPositionalEmbedding', 'embedding_forward']


def _empty_param(shape, dtype):
    # inference is the path: parameters are created frozen; every layer records its backward once
    # the caller flips ``requires_grad_(True)`` (autograd.py).
    return nn.Parameter(torch.zeros(shape, dtype=T.torch_dtype(dtype)), requires_grad=False)


def trunc_norm_(tensor, stdev=0.02, scale=2.0, generator=None):
    """initializer.TruncNorm (reference src/gluonnlp/initializer.py:52-81): uniform -> erfinv ->
    scale -> clip to +-scale*stdev."""
    lo = (1.0 + math.erf(-scale / math.sqrt(2.0))) / 2.0
    hi = (1.0 + math.erf(scale / math.sqrt(2.0))) / 2.0
    u = torch.empty(tensor.shape, dtype=torch.float32, device=tensor.device)
    u.uniform_(2 * lo - 1, 2 * hi - 1, generator=generator)
    w = torch.erfinv(u) * (stdev * math.sqrt(2.0))
    tensor.copy_(w.clamp_(-scale * stdev, scale * stdev).to(tensor.dtype))
    return tensor


# ------------------------------------------------------------------------------------------
# activations  (layers.py:236-338)
# ------------------------------------------------------------------------------------------
class Activation(nn.Module):
    """Elementwise activation evaluated by the HIP activation kernel."""

    def __init__(self, act):
        super().__init__()
        if act not in _lib.ACT:
            raise NotImplementedError('act={} is not supported'.format(act))
        self._act = act

    def forward(self, x):
        xt = T.as_tensor(x)
        dev = T.require_device(xt)
        xt = xt.contiguous()

        def call(xt):
            out = torch.empty_like(xt)
            _lib.check(_lib.load().gnlp_activation(T.ptr(out), T.ptr(xt), T.code(xt), xt.numel(),
                                                   _lib.ACT[self._act], T.stream(dev)))
            return out
        out = ActFn.apply(xt, _lib.ACT[self._act], call) if wants_grad(xt) else call(xt)
        return T.like_caller(out, x)

    def __repr__(self):
        return '{}({})'.format(self.__class__.__name__, self._act)


class IdentityActivation(nn.Module):
    def forward(self, x):
        return x


class GELU(Activation):
    """Gaussian Error Linear Unit, mode in {'erf', 'tanh', 'sigmoid'}.  layers.py:281-338."""

    def __init__(self, mode='erf'):
        if mode not in ['erf', 'tanh', 'sigmoid']:
            raise ValueError('Unsupported mode, only support "erf", "tanh", or "sigmoid". '
                             'Received mode={}'.format(mode))
        super().__init__({'erf': 'gelu', 'tanh': 'gelu(tanh)', 'sigmoid': 'gelu(sigmoid)'}[mode])
        self._mode = mode

    def __repr__(self):
        return '{}(mode={})'.format(self.__class__.__name__, self._mode)


def get_activation(act):
    """layers.py:236-271."""
    if act is None:
        return lambda x: x
    if isinstance(act, str):
        if act == 'identity':
            return IdentityActivation()
        if act == 'gelu':
            return GELU(mode='erf')
        if act == 'gelu(tanh)':
            return GELU(mode='tanh')
        if act == 'gelu(sigmoid)':
            return GELU(mode='sigmoid')
        if act in ['leaky', 'elu', 'relu', 'sigmoid', 'tanh', 'softrelu', 'softsign']:
            return Activation(act)
        raise NotImplementedError('act={} is not supported'.format(act))
    return act


def _act_name(act):
    """Name understood by the fused epilogues, or None for a user-supplied callable."""
    if act is None:
        return 'identity'
    if isinstance(act, str):
        if act not in _lib.ACT:
            raise NotImplementedError('act={} is not supported'.format(act))
        return act
    return getattr(act, '_act', None)


# ------------------------------------------------------------------------------------------
# norms  (layers.py:38-181; mxnet nn.LayerNorm)
# ------------------------------------------------------------------------------------------
def _norm_call(x, gamma, beta, kind, eps, residual=None):
    xt = T.as_tensor(x)
    rt = T.as_tensor(residual)
    dev = T.require_device(xt, rt, gamma, beta)

    def call(xt, rt, gamma, beta):
        xt = xt.contiguous()
        if rt is not None:
            rt = rt.contiguous()
        C = xt.shape[-1]
        out = torch.empty_like(xt)
        _lib.check(_lib.load().gnlp_norm(
            T.ptr(out), T.ptr(xt), T.ptr(rt), T.ptr(gamma), T.ptr(beta), T.code(xt), kind,
            float(eps), xt.numel() // max(C, 1), C, None, 0, 0, 0, 0, T.stream(dev)))
        return out

    if wants_grad(xt, rt, gamma, beta):
        out = NormFn.apply(xt, rt, gamma, beta, kind, eps, call)
    else:
        out = call(xt, rt, gamma, beta)
    return T.like_caller(out, x)


class LayerNorm(nn.Module):
    """mxnet.gluon.nn.LayerNorm(axis=-1, epsilon, in_channels): biased variance,
    gamma * (x - mean) / sqrt(var + eps) + beta  (used models/transformer.py:217-218)."""

    def __init__(self, axis=-1, epsilon=1e-5, center=True, scale=True, in_channels=0,
                 dtype='float32', **kwargs):
        super().__init__()
        assert axis == -1, 'only the last axis is supported'
        self._epsilon = epsilon
        self.gamma = _empty_param((in_channels,), dtype)
        self.beta = _empty_param((in_channels,), dtype)
        self.gamma.data.fill_(1.0)

    def forward(self, data, residual=None):
        return _norm_call(data, self.gamma, self.beta, _lib.NORM['layer_norm'],
                          self._epsilon, residual)


class NoNorm(nn.Module):
    """out = gamma * data + beta.  layers.py:81-147."""

    def __init__(self, in_channels, center=True, scale=True, beta_initializer='zeros',
                 gamma_initializer='ones', dtype='float32', **kwargs):
        super().__init__()
        self._kwargs = {'center': center, 'scale': scale}
        self._in_channels = in_channels
        self.gamma = _empty_param((in_channels,), dtype)
        self.beta = _empty_param((in_channels,), dtype)
        self.gamma.data.fill_(1.0)

    def forward(self, data):
        return _norm_call(data, self.gamma, self.beta, _lib.NORM['no_norm'], 0.0)


class RMSNorm(nn.Module):
    """x * rsqrt(mean(x^2) + eps) * gamma + beta.  layers.py:151-181."""

    def __init__(self, in_channels, center=True, scale=True, beta_initializer='zeros',
                 gamma_initializer='ones', variance_epsilon=1E-6, dtype='float32', **kwargs):
        super().__init__()
        self._kwargs = {'center': center, 'scale': scale}
        self._in_channels = in_channels
        self._epsilon = variance_epsilon
        self.gamma = _empty_param((in_channels,), dtype)
        self.beta = _empty_param((in_channels,), dtype)
        self.gamma.data.fill_(1.0)

    def forward(self, data):
        return _norm_call(data, self.gamma, self.beta, _lib.NORM['rms_norm'],
                          self._epsilon)


def get_norm_layer(normalization: str = 'layer_norm', axis: int = -1, epsilon: float = 1e-5,
                   in_channels: int = 0, **kwargs):
    """layers.py:38-77.  NOTE (kept from the reference): 'rms_norm' does not receive ``epsilon``."""
    if isinstance(normalization, str):
        if normalization == 'layer_norm':
            return LayerNorm(axis=axis, epsilon=epsilon, in_channels=in_channels, **kwargs)
        if normalization == 'no_norm':
            return NoNorm(in_channels=in_channels, **kwargs)
        if normalization == 'rms_norm':
            return RMSNorm(in_channels=in_channels, **kwargs)
        if normalization == 'identity':
            return IdentityActivation()
        raise NotImplementedError('normalization={} is not supported'.format(normalization))
    raise NotImplementedError('The type of normalization must be str')


def _norm_kind(layer):
    if isinstance(layer, LayerNorm):
        return _lib.NORM['layer_norm'], layer._epsilon
    if isinstance(layer, RMSNorm):
        return _lib.NORM['rms_norm'], layer._epsilon
    if isinstance(layer, NoNorm):
        return _lib.NORM['no_norm'], 0.0
    return None, None


# ------------------------------------------------------------------------------------------
# Dense  (mxnet nn.Dense(flatten=False))
# ------------------------------------------------------------------------------------------
def memory_kv(prev_state, mem, k_dense, v_dense):
    """-> (k_dense(mem), v_dense(mem), kept) for a decoder layer's memory attention during
    incremental decoding.  The reference projects the whole memory again at every target position
    (models/transformer.py:716-724, models/t5.py:462-470): B * L_mem rows through two GEMMs per
    layer and step for a result that does not change.  Here the pair travels with the decoding
    state: the caller attaches ``kept`` to the key cache it returns (attribute
    ``_gnlp_memory_kv``), and the next step takes the pair from the cache it is handed
    (``prev_state``) when the very same memory tensor object comes in again with its version
    counter, and the four parameters with their storage and version counters, unchanged.  The
    first step, states that went through a reordering (beam search), autograd, non-torch callers
    (a fresh DLPack view per call) and every mismatch recompute.

    Limit: version counters do not see in-place writes through ``.data`` (``p.data.copy_(...)``,
    ``mem.data.mul_(...)``); after such an edit of the memory or of the key / value projections,
    call ``drop_memory_kv(states)`` on the decoding states (or start from fresh states) -- a state
    that still carries the pair would serve the projections of the old values."""
    m = T.as_tensor(mem)
    params = [p for p in (k_dense.weight, k_dense.bias, v_dense.weight, v_dense.bias)
              if p is not None]
    key = None
    if m is mem and not wants_grad(m, *params):
        key = (m._version, m.data_ptr(), tuple(m.shape), tuple(m.stride()), m.dtype) + \
            tuple((p.data_ptr(), p._version) for p in params)
        hit = getattr(prev_state, '_gnlp_memory_kv', None)
        if hit is not None and hit[0] is m and hit[1] == key:
            return hit[2], hit[3], hit
    k, v = T.as_tensor(k_dense(m)), T.as_tensor(v_dense(m))
    return k, v, ((m, key, k, v) if key is not None else None)


def drop_memory_kv(states):
    """Forget the memory key / value projections that travel with incremental-decoding states
    (any nesting of lists / tuples of the tensors ``incremental_decode`` returned): the next step
    projects the memory again.  For callers that edited the memory or the projection weights in
    place through ``.data`` (see ``memory_kv``)."""
    if isinstance(states, (list, tuple)):
        for s_ in states:
            drop_memory_kv(s_)
    elif hasattr(states, '_gnlp_memory_kv'):
        try:
            del states._gnlp_memory_kv
        except AttributeError:
            states._gnlp_memory_kv = None


class Dense(nn.Module):
    """y = act(x W^T + b) with W of shape (units, in_units); optional fused ``other`` operand
    (residual add or gate multiply)."""

    def __init__(self, units, in_units, use_bias=True, flatten=False, activation=None,
                 weight_initializer=None, bias_initializer='zeros', dtype='float32'):
        super().__init__()
        assert not flatten, 'only flatten=False is used on the encoder path'
        self._units, self._in_units = units, in_units
        self._act = _act_name(activation)
        self.weight = _empty_param((units, in_units), dtype)
        self.bias = _empty_param((units,), dtype) if use_bias else None
        with torch.no_grad():
            if callable(weight_initializer):
                weight_initializer(self.weight.data)
            else:
                trunc_norm_(self.weight.data, 0.02)

    def forward(self, x, other=None, combine='add', act=None, out=None):
        xt = T.as_tensor(x)
        ot = T.as_tensor(other)
        w = self.weight
        b = self.bias
        dev = T.require_device(xt, ot, w, b)
        if xt.dtype != w.dtype:
            raise TypeError('input dtype {} != weight dtype {}'.format(xt.dtype, w.dtype))
        K = xt.shape[-1]
        assert K == self._in_units, 'last axis {} != in_units {}'.format(K, self._in_units)
        N = self._units
        act = self._act if act is None else act
        if ot is not None:
            ot = ot.contiguous()

        def call(xt, w, b, out=out):
            xt = xt.contiguous()
            M = xt.numel() // max(K, 1)
            if out is None:
                out = torch.empty(xt.shape[:-1] + (N,), dtype=xt.dtype, device=dev)
            _lib.check(_lib.load().gnlp_dense(
                T.ptr(out), T.ptr(xt), T.ptr(w), T.ptr(b), T.ptr(ot),
                _lib.COMBINE_MUL if combine == 'mul' else _lib.COMBINE_ADD, _lib.ACT[act],
                T.code(xt), M, N, K, K, N, N, T.stream(dev)))
            return out

        if wants_grad(xt, ot, w, b):
            if out is not None:
                raise ValueError('Dense(out=...) writes into a caller buffer and cannot be '
                                 'recorded by autograd; drop out= or run under torch.no_grad()')
            if ot is None and act == 'identity':   # y = x W^T + b
                res = DenseFn.apply(xt, w, b, call)
            else:   # the fused epilogue, differentiated through the saved pre-activation
                def call_plain(xt, w, b):
                    xt = xt.contiguous()
                    M = xt.numel() // max(K, 1)
                    z = torch.empty(xt.shape[:-1] + (N,), dtype=xt.dtype, device=dev)
                    _lib.check(_lib.load().gnlp_dense(
                        T.ptr(z), T.ptr(xt), T.ptr(w), T.ptr(b), None, _lib.COMBINE_ADD,
                        _lib.ACT['identity'], T.code(xt), M, N, K, K, N, N, T.stream(dev)))
                    return z
                if ot is not None and tuple(ot.shape) != tuple(xt.shape[:-1]) + (N,):
                    raise ValueError('other {} does not match the output {}'.format(
                        tuple(ot.shape), tuple(xt.shape[:-1]) + (N,)))
                res = DenseEpilogueFn.apply(
                    xt, w, b, ot, _lib.ACT[act],
                    _lib.COMBINE_MUL if combine == 'mul' else _lib.COMBINE_ADD, call_plain)
        else:
            res = call(xt, w, b)
        return T.like_caller(res, x)


# ------------------------------------------------------------------------------------------
# Embedding tables of the front-end  (mxnet nn.Embedding; layers.py:370-482)
# ------------------------------------------------------------------------------------------
class Embedding(nn.Module):
    """mxnet.gluon.nn.Embedding(input_dim, output_dim): the table only -- the gather is fused
    with the other tables and the LayerNorm in ``embedding_forward``."""

    def __init__(self, input_dim, output_dim, weight_initializer=None, dtype='float32'):
        super().__init__()
        self._input_dim, self._output_dim = input_dim, output_dim
        self.weight = _empty_param((input_dim, output_dim), dtype)
        with torch.no_grad():
            if callable(weight_initializer):
                weight_initializer(self.weight.data)
            else:
                trunc_norm_(self.weight.data, 0.02)

    def forward(self, ids):
        return embedding_forward(ids, None, self.weight.data, None, None, None, None, 0.0, 'NT')


class SinusoidalPositionalEmbedding(nn.Module):
    """layers.py:405-456: a geometric sequence of timescales; ``base_mult`` is the reference's
    Constant of the same name (float32, computed the way ``_init_sinusoidal_base`` does)."""

    def __init__(self, units: int, dtype='float32'):
        super().__init__()
        import numpy as np
        self._units = units
        self._dtype = dtype
        half_units = units // 2
        with np.errstate(divide='ignore', invalid='ignore'):
            val = np.log(10000) / (half_units - 1)
            base = np.exp(np.arange(half_units, dtype=np.float32) * -val)
        self.register_buffer('base_mult', torch.from_numpy(base.astype(np.float32)))

    def forward(self, positions):
        """positions (...,) -> (..., units)."""
        pos = T.as_tensor(positions)
        base = self.base_mult.to(torch.float32)     # a module-wide .to(dtype) must not narrow it
        dev = T.require_device(pos, base)
        p32 = pos.to(torch.float32).contiguous()
        dt = T.torch_dtype(self._dtype)
        out = torch.empty(tuple(pos.shape) + (self._units,), dtype=dt, device=dev)
        _lib.check(_lib.load().gnlp_sinusoidal_embedding(
            T.ptr(out), T.ptr(p32), T.ptr(base), T.code(out), p32.numel(), self._units,
            T.stream(dev)))
        return T.like_caller(out, positions)

    def __repr__(self):
        return '{}(units={}, dtype={})'.format(self.__class__.__name__, self._units, self._dtype)


class LearnedPositionalEmbedding(nn.Module):
    """layers.py:460-482: a (max_length, units) table read with np.take(mode=...)."""

    def __init__(self, units, max_length, mode='clip', dtype='float32', weight_initializer=None):
        super().__init__()
        if mode not in ('clip', 'wrap', 'raise'):
            raise NotImplementedError('mode={} is not supported'.format(mode))
        self._units, self._max_length, self._mode = units, max_length, mode
        self._dtype = dtype
        self.weight = _empty_param((max_length, units), dtype)
        with torch.no_grad():
            if callable(weight_initializer):
                weight_initializer(self.weight.data)
            else:
                trunc_norm_(self.weight.data, 0.02)

    def forward(self, positions):
        """positions (...,) integer -> (..., units): np.take(weight, positions, axis=0, mode)."""
        pos = T.as_tensor(positions)
        if pos.dtype not in (torch.int32, torch.int64):
            pos = pos.to(torch.int64)
        if self._mode == 'wrap':
            pos = torch.remainder(pos, self._max_length)
        elif self._mode == 'raise':
            if pos.numel() and bool(((pos < -self._max_length) | (pos >= self._max_length)).any()):
                raise IndexError('position index out of range for a table of {} rows'.format(
                    self._max_length))
            pos = torch.where(pos < 0, pos + self._max_length, pos)   # python-style negatives
        out = embedding_forward(pos.reshape(1, -1), None, self.weight.data, None, None, None, None,
                                0.0, 'NT')
        return T.like_caller(T.as_tensor(out).reshape(tuple(pos.shape) + (self._units,)), positions)

    def __repr__(self):
        return '{}(units={}, max_length={}, mode={}, dtype={})'.format(
            self.__class__.__name__, self._units, self._max_length, self._mode, self._dtype)


class PositionalEmbedding(nn.Module):
    """layers.py:370-401; the table / constants live under ``_embed`` as in the reference."""

    def __init__(self, units, max_length=None, method='sinusoidal', dtype='float32'):
        super().__init__()
        self._units, self._max_length, self._method = units, max_length, method
        self._dtype = dtype
        if method == 'sinusoidal':
            self._embed = SinusoidalPositionalEmbedding(units=units, dtype=dtype)
        elif method == 'learned':
            self._embed = LearnedPositionalEmbedding(units=units, max_length=max_length,
                                                     dtype=dtype)
        else:
            raise NotImplementedError

    def forward(self, positions):
        return self._embed(positions)

    def table(self, inputs, layout):
        """Rows 0 .. L-1 of the position embedding for the fused front-end kernel: the learned
        table itself, or the sinusoids of arange(L) (what ``self(arange_like(inputs))`` is)."""
        if self._method == 'learned':
            return self._embed.weight.data
        ids = T.as_tensor(inputs)
        L = ids.shape[1 if layout == 'NT' else 0]
        return T.as_tensor(self._embed(torch.arange(L, device=ids.device)))


def embedding_forward(inputs, token_types, word_w, type_w, pos_w, ln_gamma, ln_beta, eps, layout):
    """word + token-type + position gathers, add, LayerNorm in one kernel
    (gnlp_embedding_forward; models/bert.py:458-504)."""
    ids = T.as_tensor(inputs)
    tt = T.as_tensor(token_types)
    dev = T.require_device(ids, tt, word_w, type_w, pos_w, ln_gamma, ln_beta)
    if ids.dtype not in (torch.int32, torch.int64):
        raise TypeError('token ids must be int32 or int64, got {}'.format(ids.dtype))
    if ids.dim() != 2:
        raise ValueError('inputs must be (batch, time) or (time, batch), got {}'.format(
            tuple(ids.shape)))
    ids = ids.contiguous()
    if tt is not None:
        if tuple(tt.shape) != tuple(ids.shape):
            raise ValueError('token_types shape {} != inputs shape {}'.format(tuple(tt.shape),
                                                                              tuple(ids.shape)))
        tt = tt.to(ids.dtype).contiguous()
    if layout == 'NT':
        B, L = ids.shape
    elif layout == 'TN':
        L, B = ids.shape
    else:
        raise NotImplementedError('layout={} is not supported'.format(layout))
    U = word_w.shape[1]
    out = torch.empty(tuple(ids.shape) + (U,), dtype=word_w.dtype, device=dev)
    _lib.check(_lib.load().gnlp_embedding_forward(
        T.ptr(out), T.ptr(ids), T.ptr(tt), _lib.I64 if ids.dtype == torch.int64 else _lib.I32,
        T.ptr(word_w), T.ptr(type_w), T.ptr(pos_w), T.ptr(ln_gamma), T.ptr(ln_beta),
        T.code(word_w), float(eps), B, L, U, word_w.shape[0],
        0 if type_w is None else type_w.shape[0], 0 if pos_w is None else pos_w.shape[0],
        int(layout == 'TN'), T.stream(dev)))
    return T.like_caller(out, inputs)


# ------------------------------------------------------------------------------------------
# BucketPositionalEmbedding  (layers.py:486-524)
# ------------------------------------------------------------------------------------------
class BucketPositionalEmbedding(nn.Module):
    """T5 bucketed relative-position embedding: weight[bucket(rel)]  -> (..., units)."""

    def __init__(self, units, bidirectional=True, num_buckets=32, max_distance=128,
                 dtype='float32', embed_initializer=None):
        super().__init__()
        self._units = units
        self._bidirectional = bidirectional
        self._num_buckets = num_buckets
        self._max_distance = max_distance
        self._dtype = dtype
        self.weight = _empty_param((num_buckets, units), dtype)
        with torch.no_grad():
            if callable(embed_initializer):
                embed_initializer(self.weight.data)
            else:
                self.weight.data.normal_(0.0, 0.02)

    def forward(self, relative_positions):
        rp = T.as_tensor(relative_positions)
        dev = T.require_device(rp, self.weight)
        rp32 = rp.to(torch.int32).contiguous()

        def call(w):
            out = torch.empty(tuple(rp32.shape) + (self._units,), dtype=w.dtype, device=dev)
            _lib.check(_lib.load().gnlp_bucket_embedding(
                T.ptr(out), T.ptr(w), T.ptr(rp32), T.code(w), rp32.numel(), self._units,
                int(self._bidirectional), self._num_buckets, self._max_distance, T.stream(dev)))
            return out

        if wants_grad(self.weight):
            out = BucketEmbedFn.apply(self.weight, rp32, (self._bidirectional, self._num_buckets,
                                                          self._max_distance), call)
        else:
            out = call(self.weight)
        return T.like_caller(out, relative_positions)

    def __repr__(self):
        return ('{}(units={}, bidirectional={}, num_buckets={}, max_distance={}, dtype={})'
                .format(self.__class__.__name__, self._units, self._bidirectional,
                        self._num_buckets, self._max_distance, self._dtype))


# ------------------------------------------------------------------------------------------
# PositionwiseFFN  (layers.py:528-650)
# ------------------------------------------------------------------------------------------
def ffn_params_struct(ffn):
    p = _lib.FfnParams()
    p.ffn1_w = T.ptr(ffn.ffn_1.weight.data)
    p.ffn1_b = T.ptr(ffn.ffn_1.bias.data) if ffn.ffn_1.bias is not None else None
    if ffn._use_gated_activation:
        p.gate_w = T.ptr(ffn.gated_ffn_1.weight.data)
        p.gate_b = T.ptr(ffn.gated_ffn_1.bias.data) if ffn.gated_ffn_1.bias is not None else None
    p.ffn2_w = T.ptr(ffn.ffn_2.weight.data)
    p.ffn2_b = T.ptr(ffn.ffn_2.bias.data) if ffn.ffn_2.bias is not None else None
    p.ln_g = T.ptr(ffn.layer_norm.gamma.data)
    p.ln_b = T.ptr(ffn.layer_norm.beta.data)
    return p


class PositionwiseFFN(nn.Module):
    """Position-wise FFN.  layers.py:528-650.

    pre_norm:  norm(data) -> fc1 -> act -> fc2 -> res(+data)
    else:      data -> fc1 -> act -> fc2 -> norm(res(+data))
    gated:     act(gated_ffn_1(x)) * ffn_1(x)  (layers.py:636-638)."""

    def __init__(self, units: int = 512, hidden_size: int = 2048, use_bias=True,
                 activation_dropout: float = 0.0, dropout: float = 0.1, weight_initializer=None,
                 bias_initializer='zeros', activation='relu', use_gated_activation=False,
                 normalization: str = 'layer_norm', layer_norm_eps: float = 1E-5,
                 pre_norm: bool = False, dtype='float32', **kwargs):
        super().__init__()
        self._dtype = dtype
        self._pre_norm = pre_norm
        self._use_gated_activation = use_gated_activation
        self._kwargs = OrderedDict([
            ('units', units), ('hidden_size', hidden_size),
            ('activation_dropout', activation_dropout), ('activation', activation),
            ('dropout', dropout), ('normalization', normalization),
            ('layer_norm_eps', layer_norm_eps), ('pre_norm', pre_norm), ('dtype', self._dtype)])
        self.ffn_1 = Dense(units=hidden_size, in_units=units, use_bias=use_bias,
                           weight_initializer=weight_initializer, dtype=dtype)
        if use_gated_activation:
            self.gated_ffn_1 = Dense(units=hidden_size, in_units=units, use_bias=use_bias,
                                     weight_initializer=weight_initializer, dtype=dtype)
        self.activation = get_activation(activation)
        self._act_name = _act_name(activation)
        self.ffn_2 = Dense(units=units, in_units=hidden_size, use_bias=use_bias,
                           weight_initializer=weight_initializer, dtype=dtype)
        kwargs.pop('gamma_initializer', None)
        self.layer_norm = get_norm_layer(in_channels=units, normalization=normalization,
                                         epsilon=layer_norm_eps, dtype=dtype, **kwargs)

    def cfg_struct(self, dtype_code):
        kind, eps = _norm_kind(self.layer_norm)
        c = _lib.FfnCfg()
        c.units, c.hidden = self._kwargs['units'], self._kwargs['hidden_size']
        c.act = _lib.ACT[self._act_name]
        c.gated = int(self._use_gated_activation)
        c.norm_kind, c.pre_norm, c.dtype, c.ln_eps = kind, int(self._pre_norm), dtype_code, eps
        return c

    def forward(self, data):
        """data (..., units) -> (..., units).  One C call launches the whole block."""
        kind, _ = _norm_kind(self.layer_norm)
        if self._act_name is None or kind is None:
            return self._forward_unfused(data)
        if wants_grad(T.as_tensor(data), *self.parameters()):
            return self._forward_unfused(data)   # op by op: every op records its backward
        x = T.as_tensor(data)
        dev = T.require_device(x, self.ffn_1.weight.data)
        x = x.contiguous()
        U = x.shape[-1]
        rows = x.numel() // max(U, 1)
        out = torch.empty_like(x)
        cfg = self.cfg_struct(T.code(x))
        lib = _lib.load()
        ws = T.workspace(dev, lib.gnlp_ffn_workspace_bytes(ctypes.byref(cfg), rows))
        p = ffn_params_struct(self)
        _lib.check(lib.gnlp_positionwise_ffn(
            T.ptr(out), T.ptr(x), ctypes.byref(p), ctypes.byref(cfg), rows, None, 0, 0, 0, 0,
            T.ptr(ws), 0 if ws is None else ws.numel(), T.stream(dev)))
        return T.like_caller(out, data)

    def _forward_unfused(self, data):
        """Op-by-op path for user-supplied activation / identity normalisation (layers.py:633-647)."""
        residual = data
        if self._pre_norm:
            data = self.layer_norm(data)
        if self._use_gated_activation:
            out = self.ffn_1(data, other=self.activation(self.gated_ffn_1(data)), combine='mul')
        else:
            out = self.activation(self.ffn_1(data))
        out = self.ffn_2(out, other=residual, combine='add')
        if not self._pre_norm:
            out = self.layer_norm(out)
        return out

    def __repr__(self):
        s = '{}(\n'.format(self.__class__.__name__)
        s += ',\n'.join('\t{}={}'.format(k, v) for k, v in self._kwargs.items())
        return s + '\n)'
