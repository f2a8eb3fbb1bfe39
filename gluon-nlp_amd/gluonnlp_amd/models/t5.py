"""T5Block (encoder form) / T5Encoder -- mirrors the encoder half of ``gluonnlp/models/t5.py``
(reference src/gluonnlp/models/t5.py:173-350, 492-572, 576-711).  The decoder / incremental
decoding is outside the encoder hot path."""
import ctypes

import torch
from torch import nn

from .. import _lib
from .. import _tensor as T
from ..attention_cell import (MultiHeadAttentionCell, RelAttentionScoreCell, gen_rel_position,
                              gen_self_attn_mask, mask_predicate)
from ..layers import Dense, PositionwiseFFN, RMSNorm, ffn_params_struct

__all__ = ['T5Block', 'T5Encoder']


def _normal(std):
    def init(t):
        t.normal_(0.0, std)
    return init


class T5Block(nn.Module):
    """Pre-RMSNorm block with bias-free q/k/v/proj, ``scaled=False`` attention and the shared
    relative-position bias as ``edge_scores``.  models/t5.py:173-350, forward :492-572."""

    def __init__(self, d_model, d_kv, d_ff, is_decoder, num_heads=12, dropout_prob=0.1,
                 layer_norm_eps=1E-6, activation='relu', init_factor=1.0, layout='NT',
                 dtype='float32'):
        super().__init__()
        if is_decoder:
            raise NotImplementedError('T5 decoder blocks are outside the encoder hot path '
                                      '(SURVEY.md §8f)')
        self._d_model, self._d_kv, self._d_ff = d_model, d_kv, d_ff
        self._is_decoder = is_decoder
        self._num_heads = num_heads
        self._inner_dim = self._num_heads * self._d_kv
        self._dtype = dtype
        self._layer_norm_eps = layer_norm_eps
        assert layout in ['TN', 'NT'], \
            'Invalid layout: {}. Only "TN" and "NT" are supported.'.format(layout)
        self._layout = layout
        self._time_axis = 1 if self.layout == 'NT' else 0
        self.self_attn_layer_norm = RMSNorm(in_channels=d_model, center=False, scale=True,
                                            variance_epsilon=layer_norm_eps, dtype=dtype)
        self.self_attn_q = Dense(self._inner_dim, d_model, use_bias=False, dtype=dtype,
                                 weight_initializer=_normal((d_model * d_kv) ** -0.5 * init_factor))
        self.self_attn_k = Dense(self._inner_dim, d_model, use_bias=False, dtype=dtype,
                                 weight_initializer=_normal(d_model ** -0.5 * init_factor))
        self.self_attn_v = Dense(self._inner_dim, d_model, use_bias=False, dtype=dtype,
                                 weight_initializer=_normal(d_model ** -0.5 * init_factor))
        self.self_attn = MultiHeadAttentionCell(
            query_units=self._inner_dim, num_heads=num_heads, attention_dropout=dropout_prob,
            scaled=False, normalized=False, dtype=dtype,
            layout='NTK' if layout == 'NT' else 'TNK', use_einsum=False)
        self.self_attn_proj = Dense(d_model, self._inner_dim, use_bias=False, dtype=dtype,
                                    weight_initializer=_normal(self._inner_dim ** -0.5 * init_factor))
        assert activation in ['relu', 'gated-gelu'], \
            '{} is not supported. Please choose from "relu" and "gated-gelu"'.format(activation)
        self._activation = activation
        self.ffn = PositionwiseFFN(
            units=d_model, hidden_size=d_ff, use_bias=False, activation_dropout=dropout_prob,
            dropout=dropout_prob, weight_initializer=_normal(d_model ** -0.5 * init_factor),
            activation='relu' if activation == 'relu' else 'gelu(tanh)',
            use_gated_activation=False if activation == 'relu' else True,
            normalization='rms_norm', layer_norm_eps=layer_norm_eps, pre_norm=True, dtype=dtype,
            center=False, scale=True)

    @property
    def layout(self):
        return self._layout

    def params_struct(self):
        p = _lib.T5BlockParams()
        p.attn_ln_g = T.ptr(self.self_attn_layer_norm.gamma.data)
        p.attn_ln_b = T.ptr(self.self_attn_layer_norm.beta.data)
        p.q_w = T.ptr(self.self_attn_q.weight.data)
        p.k_w = T.ptr(self.self_attn_k.weight.data)
        p.v_w = T.ptr(self.self_attn_v.weight.data)
        p.proj_w = T.ptr(self.self_attn_proj.weight.data)
        p.ffn = ffn_params_struct(self.ffn)
        return p

    def cfg_struct(self, dtype_code):
        c = _lib.T5Cfg()
        c.d_model, c.d_kv, c.d_ff, c.heads = self._d_model, self._d_kv, self._d_ff, self._num_heads
        c.gated = int(self._activation != 'relu')
        c.dtype = dtype_code
        c.time_major = int(self._layout == 'TN')
        c.ln_eps = self._layer_norm_eps
        return c

    def forward(self, hidden_states, self_attn_mask, position_embeddings, mem_states=None,
                mem_attn_mask=None):
        """models/t5.py:492-572 (encoder form)."""
        m = T.as_tensor(self_attn_mask)
        pred = mask_predicate(m) if m is not None else ('ones', None)
        if pred is not None:
            vl = pred[1] if pred[0] == 'full' else None
            return run_t5_blocks([self], hidden_states, vl, position_embeddings, None,
                                 self._layout)
        # arbitrary dense mask: op-by-op path through the same kernels (t5.py:537-571)
        out = self.self_attn_layer_norm(hidden_states)
        shp = tuple(out.shape[:2]) + (self._num_heads, self._d_kv)
        q = self.self_attn_q(out).reshape(shp)
        k = self.self_attn_k(out).reshape(shp)
        v = self.self_attn_v(out).reshape(shp)
        out, _ = self.self_attn(q, k, v, m, position_embeddings, _need_scores=False,
                                _need_weights=False)
        hidden_states = self.self_attn_proj(out, other=hidden_states, combine='add')
        return self.ffn(hidden_states)


def run_t5_blocks(blocks, hidden_states, valid_length, position_bias, final_ln, layout):
    """All blocks (+ the final RMSNorm) in ONE C call (gnlp_t5_encoder_forward)."""
    x = T.as_tensor(hidden_states)
    pb = T.as_tensor(position_bias)
    vlt, vlp, vlc = T.valid_length_arg(valid_length)
    b0 = blocks[0]
    dev = T.require_device(x, pb, vlt, b0.self_attn_q.weight.data)
    x = x.contiguous()
    if x.dtype != b0.self_attn_q.weight.dtype:
        raise TypeError('hidden_states dtype {} != parameter dtype {}'.format(
            x.dtype, b0.self_attn_q.weight.dtype))
    if layout == 'NT':
        B, L = x.shape[0], x.shape[1]
    else:
        L, B = x.shape[0], x.shape[1]
    if pb is not None:
        if pb.dtype != x.dtype:
            pb = pb.to(x.dtype)
        pb = torch.broadcast_to(pb, (b0._num_heads, L, L)).contiguous()
    n = len(blocks)
    cfg = b0.cfg_struct(T.code(x))
    arr = (_lib.T5BlockParams * n)(*[b.params_struct() for b in blocks])
    lib = _lib.load()
    ws = T.workspace(dev, lib.gnlp_t5_workspace_bytes(ctypes.byref(cfg), B, L))
    out = torch.empty_like(x)
    _lib.check(lib.gnlp_t5_encoder_forward(
        T.ptr(out), T.ptr(x), arr, n,
        T.ptr(final_ln.gamma.data) if final_ln is not None else None,
        T.ptr(final_ln.beta.data) if final_ln is not None else None,
        ctypes.byref(cfg), B, L, vlp, vlc, T.ptr(pb), T.ptr(ws),
        0 if ws is None else ws.numel(), T.stream(dev)))
    return T.like_caller(out, hidden_states)


class T5Encoder(nn.Module):
    """models/t5.py:576-711."""

    def __init__(self, d_model, d_kv, d_ff, num_layers=12, num_heads=12, dropout_prob=0.1,
                 layer_norm_eps=1E-6, activation='relu', init_factor=1.0, layout='NT',
                 dtype='float32'):
        super().__init__()
        self._d_model, self._d_kv, self._d_ff = d_model, d_kv, d_ff
        self._num_layers = num_layers
        self._num_heads = num_heads
        self._inner_dim = num_heads * d_kv
        self._dtype = dtype
        assert layout in ['TN', 'NT'], \
            'Invalid layout: {}. Only "TN" and "NT" are supported.'.format(layout)
        self._layout = layout
        self._time_axis = 1 if self.layout == 'NT' else 0
        self.relative_position_encoder = RelAttentionScoreCell(
            query_units=self._inner_dim, num_heads=num_heads, method='t5', bidirectional=True,
            embed_initializer=_normal(d_model ** -0.5 * init_factor),
            layout='NTK' if layout == 'NT' else 'TNK', dtype=dtype)
        self.layers = nn.ModuleList([
            T5Block(d_model=d_model, d_kv=d_kv, d_ff=d_ff, is_decoder=False, num_heads=num_heads,
                    dropout_prob=dropout_prob, layer_norm_eps=layer_norm_eps,
                    activation=activation, init_factor=init_factor, layout=layout, dtype=dtype)
            for _ in range(num_layers)])
        self.final_layer_norm = RMSNorm(in_channels=d_model, center=False, scale=True,
                                        variance_epsilon=layer_norm_eps, dtype=dtype)

    @property
    def layout(self):
        return self._layout

    def forward(self, hidden_states, valid_length):
        """models/t5.py:666-711: the (H, L, L) bias is built once (fused rel-position + bucket +
        gather kernel) and shared by every layer and batch element."""
        x = T.as_tensor(hidden_states)
        L = x.shape[self._time_axis]
        position_embeddings = self.relative_position_encoder.self_attn_bias(L)
        return run_t5_blocks(list(self.layers), hidden_states, valid_length, position_embeddings,
                             self.final_layer_norm, self._layout)
