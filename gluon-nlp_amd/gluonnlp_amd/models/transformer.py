"""TransformerEncoderLayer / TransformerEncoder -- mirrors the encoder half of
``gluonnlp/models/transformer.py`` (reference src/gluonnlp/models/transformer.py:136-385) --
and, composed from the same kernels, the forward of TransformerDecoderLayer /
TransformerDecoder (:389-598, :738-840), including incremental decoding with key / value caches
(TransformerDecoderLayer.incremental_decode, :599-737)."""
import ctypes

import torch
from torch import nn

from .. import _lib
from .. import _tensor as T
from ..attention_cell import (MultiHeadAttentionCell, gen_mem_attn_mask, gen_self_attn_mask,
                              mask_predicate)
from ..autograd import SeqMaskFn, wants_grad
from ..layers import Dense, LayerNorm, PositionwiseFFN, ffn_params_struct, memory_kv

__all__ = ['TransformerEncoderLayer', 'TransformerEncoder', 'run_encoder_stack',
           'drop_folded_parameters',
           'TransformerDecoderLayer', 'TransformerDecoder']


class TransformerEncoderLayer(nn.Module):
    """Transformer Encoder Layer.  models/transformer.py:136-272."""

    def __init__(self, units: int = 512, hidden_size: int = 2048, num_heads: int = 8,
                 attention_dropout_prob: float = 0.1, hidden_dropout_prob: float = 0.1,
                 activation_dropout_prob: float = 0.0, layer_norm_eps: float = 1e-12,
                 pre_norm: bool = False, use_qkv_bias: bool = True, weight_initializer=None,
                 bias_initializer='zeros', activation: str = 'relu', dtype='float32',
                 layout='NT'):
        super().__init__()
        self._units = units
        self._hidden_size = hidden_size
        self._num_heads = num_heads
        self._attention_dropout_prob = attention_dropout_prob
        self._hidden_dropout_prob = hidden_dropout_prob
        self._activation_dropout_prob = activation_dropout_prob
        self._layer_norm_eps = layer_norm_eps
        self._pre_norm = pre_norm
        self._dtype = dtype
        self._layout = layout
        self._activation = activation
        assert layout in ['TN', 'NT'], 'Invalid layout received = {}. ' \
                                       'Only "TN" and "NT" are accepted!'.format(layout)
        assert self._units % self._num_heads == 0, 'units must be divisive by the number of heads'
        self.attn_qkv = Dense(3 * units, in_units=units, use_bias=use_qkv_bias,
                              weight_initializer=weight_initializer, dtype=dtype)
        self.attention_proj = Dense(units=units, in_units=units, use_bias=True,
                                    weight_initializer=weight_initializer, dtype=dtype)
        attention_layout = 'NTK' if self._layout == 'NT' else 'TNK'
        self.attention_cell = MultiHeadAttentionCell(
            query_units=self._units, num_heads=self._num_heads,
            attention_dropout=self._attention_dropout_prob, scaled=True, dtype=self._dtype,
            layout=attention_layout)
        self.layer_norm = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        self.ffn = PositionwiseFFN(units=units, hidden_size=hidden_size,
                                   dropout=hidden_dropout_prob,
                                   activation_dropout=activation_dropout_prob,
                                   weight_initializer=weight_initializer,
                                   layer_norm_eps=layer_norm_eps, activation=activation,
                                   pre_norm=pre_norm, dtype=self._dtype)

    @property
    def layout(self) -> str:
        return self._layout

    # ---- C structs -------------------------------------------------------------------------
    def params_struct(self):
        p = _lib.EncoderLayerParams()
        p.qkv_w = T.ptr(self.attn_qkv.weight.data)
        p.qkv_b = T.ptr(self.attn_qkv.bias.data) if self.attn_qkv.bias is not None else None
        p.proj_w = T.ptr(self.attention_proj.weight.data)
        p.proj_b = T.ptr(self.attention_proj.bias.data)
        p.ln_g = T.ptr(self.layer_norm.gamma.data)
        p.ln_b = T.ptr(self.layer_norm.beta.data)
        p.ffn = ffn_params_struct(self.ffn)
        return p

    def cfg_struct(self, dtype_code):
        c = _lib.EncoderCfg()
        c.units, c.hidden, c.heads = self._units, self._hidden_size, self._num_heads
        c.act = _lib.ACT[self.ffn._act_name]
        c.pre_norm, c.dtype = int(self._pre_norm), dtype_code
        c.time_major = int(self._layout == 'TN')
        c.ln_eps = self._layer_norm_eps
        return c

    def forward(self, data, attn_mask, *, _need_weights=True, _valid_length=None,
                _final_seq_mask=False):
        """data (B, L, C) ['NT'] or (L, B, C) ['TN'], attn_mask (B, L, L)
        -> (out, attn_weight (B, H, L, L)).  models/transformer.py:234-272.

        One C call launches the whole layer.  ``_need_weights=False`` (used by the stacks when
        ``output_attention`` is off) returns ``None`` for attn_weight and lets the fused
        attention kernel run."""
        x = T.as_tensor(data)
        if wants_grad(x, *self.parameters()):
            return self._forward_recorded(data, attn_mask, _need_weights, _valid_length,
                                          _final_seq_mask)
        m = T.as_tensor(attn_mask)
        vlt, vlp, vlc = T.valid_length_arg(_valid_length)
        dev = T.require_device(x, m, vlt, self.attn_qkv.weight.data)
        x = x.contiguous()
        if x.dtype != self.attn_qkv.weight.dtype:
            raise TypeError('data dtype {} != parameter dtype {}'.format(
                x.dtype, self.attn_qkv.weight.dtype))
        if self._layout == 'NT':
            B, L = x.shape[0], x.shape[1]
        else:
            L, B = x.shape[0], x.shape[1]
        mptr = None
        if m is not None and vlp is None:
            pred = mask_predicate(m)
            if pred is not None and tuple(m.shape) == (B, L, L):
                if pred[0] == 'full':
                    vlt, vlp, vlc = T.valid_length_arg(pred[1])
            else:
                if m.dtype != torch.bool:
                    m = m != 0
                m = torch.broadcast_to(m, (B, L, L)).contiguous()
                mptr = T.ptr(m)
        out = torch.empty_like(x)
        w = torch.empty((B, self._num_heads, L, L), dtype=x.dtype, device=dev) \
            if _need_weights else None
        cfg = self.cfg_struct(T.code(x))
        p = self.params_struct()
        lib = _lib.load()
        ws = T.workspace(dev, lib.gnlp_encoder_workspace_bytes(ctypes.byref(cfg), B, L,
                                                               int(_need_weights)))
        _lib.check(lib.gnlp_encoder_layer_forward(
            T.ptr(out), T.ptr(x), ctypes.byref(p), ctypes.byref(cfg), B, L, mptr, vlp, vlc,
            T.ptr(w), int(bool(_final_seq_mask)), T.ptr(ws), 0 if ws is None else ws.numel(),
            T.stream(dev)))
        return T.like_caller(out, data), (T.like_caller(w, data) if w is not None else None)

    def _forward_recorded(self, data, attn_mask, need_weights, valid_length, final_seq_mask):
        """The same layer op by op, every op recording its backward (training; the reference
        differentiates exactly this composition under mx.autograd, models/transformer.py:259-271):
        Dense (+ residual epilogue), the attention cell, LayerNorm and the FFN are the autograd
        functions of autograd.py over the HIP kernels."""
        x = T.as_tensor(data)
        if self._pre_norm:                      # :259-260 (data is REBOUND to the normalised tensor)
            x = self.layer_norm(x)
        U, H = self._units, self._num_heads
        qkv = self.attn_qkv(x)                  # :261
        q, k, v = [qkv[..., i * U:(i + 1) * U].reshape(x.shape[0], x.shape[1], H, U // H)
                   for i in range(3)]           # :262-264
        ctx, (_, w) = self.attention_cell(q, k, v, attn_mask, None, _need_scores=False,
                                          _need_weights=need_weights,
                                          _valid_length=valid_length)          # :265
        out = self.attention_proj(ctx, other=x, combine='add')                 # :266-268
        if not self._pre_norm:
            out = self.layer_norm(out)          # :269-270
        out = self.ffn(out)                     # :271
        if final_seq_mask:
            out = sequence_mask(out, valid_length, self._layout)
        return T.like_caller(T.as_tensor(out), data), w


# ---- LayerNorms folded into the GEMMs around them (bf16 post-norm stacks) ---------------------
# gnlp_encoder_stack_forward_folded takes, per layer, the folded parameters of attn_qkv (with the
# PREVIOUS layer's ffn.layer_norm) and of ffn.ffn_1 (with this layer's layer_norm).  They are a
# function of the parameters only, computed on the device once per parameter version and kept with
# the first layer of the stack.  The key is (data_ptr, _version) of every tensor involved: an
# optimizer step, load_state_dict or `p.data = ...` change it; an in-place write through `.data`
# (`p.data.copy_(...)`) changes neither -- call drop_folded_parameters(layers) after such an edit
# (or set GNLP_LN_FOLD=0).
_FOLD_ATTR = '_gnlp_folded_params'


def drop_folded_parameters(layers):
    """Forget the folded LayerNorm parameters of a stack (a list of TransformerEncoderLayers, or a
    module that contains them): the next forward recomputes them."""
    mods = layers.modules() if isinstance(layers, nn.Module) else layers
    for m in mods:
        if hasattr(m, _FOLD_ATTR):
            delattr(m, _FOLD_ATTR)


def _fold_sources(layers):
    for ly in layers:
        for t in (ly.attn_qkv.weight, ly.attn_qkv.bias, ly.layer_norm.gamma, ly.layer_norm.beta,
                  ly.ffn.ffn_1.weight, ly.ffn.ffn_1.bias, ly.ffn.layer_norm.gamma,
                  ly.ffn.layer_norm.beta):
            yield t


def _fold_key(layers):
    return tuple(None if t is None else (t.data_ptr(), t._version) for t in _fold_sources(layers))


def _folded_params(layers, dev):
    """(ctypes array, tensors kept alive) or None when the stack cannot fold at all."""
    l0 = layers[0]
    if l0.attn_qkv.weight.dtype != torch.bfloat16 or l0._pre_norm:
        return None
    for ly in layers:
        if ly._pre_norm or ly.ffn._pre_norm or ly.ffn._use_gated_activation or \
                not isinstance(ly.ffn.layer_norm, LayerNorm):
            return None
        if ly.layer_norm.beta is None or ly.ffn.layer_norm.beta is None:
            return None
    key = _fold_key(layers)
    cached = getattr(l0, _FOLD_ATTR, None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    lib = _lib.load()
    stream = T.stream(dev)
    keep = []
    n = len(layers)
    arr = (_lib.EncoderLayerFolded * n)()

    def fold(dense, norm):
        w = dense.weight.data.contiguous()
        N, K = w.shape
        wf = torch.empty_like(w)
        c = torch.empty(N, dtype=torch.float32, device=dev)
        bf = torch.empty(N, dtype=torch.float32, device=dev)
        _lib.check(lib.gnlp_fold_norm_dense(
            T.ptr(wf), T.ptr(c), T.ptr(bf), T.ptr(w),
            T.ptr(dense.bias.data) if dense.bias is not None else None, T.ptr(norm.gamma.data),
            T.ptr(norm.beta.data), _lib.BF16, N, K, 0, 1.0, stream))
        keep.extend([wf, c, bf])
        return T.ptr(wf), T.ptr(c), T.ptr(bf)

    for i, ly in enumerate(layers):
        if i > 0:
            arr[i].qkv_wf, arr[i].qkv_c, arr[i].qkv_bf = fold(ly.attn_qkv, layers[i - 1].ffn.layer_norm)
        arr[i].ffn1_wf, arr[i].ffn1_c, arr[i].ffn1_bf = fold(ly.ffn.ffn_1, ly.layer_norm)
    # (the entry also holds the source storages: while it lives, no other tensor can be handed
    # their addresses, so an equal data_ptr means the same memory)
    keep.extend(t.data for t in _fold_sources(layers) if t is not None)
    setattr(l0, _FOLD_ATTR, (key, arr, keep))
    return arr, keep


def sequence_mask(data, valid_length, layout):
    """npx.sequence_mask(data, valid_length, axis=time) (models/bert.py:311-312); records its
    backward (the cotangent masked alike) under autograd."""
    x = T.as_tensor(data).contiguous()
    vlt, vlp, vlc = T.valid_length_arg(valid_length)
    dev = T.require_device(x, vlt)
    B, L = (x.shape[0], x.shape[1]) if layout == 'NT' else (x.shape[1], x.shape[0])

    def call(t):
        out = torch.empty_like(t)
        _lib.check(_lib.load().gnlp_sequence_mask(T.ptr(out), T.ptr(t), vlp, vlc, T.code(t), B, L,
                                                  t.shape[-1], int(layout == 'TN'), T.stream(dev)))
        return out
    return SeqMaskFn.apply(x, call) if wants_grad(x) else call(x)


def run_encoder_stack(layers, data, valid_length, layout, final_seq_mask, all_encodings=False):
    """Run a list of post-/pre-norm TransformerEncoderLayers back to back in ONE C call
    (gnlp_encoder_stack_forward); the mask is the valid_length predicate evaluated in-kernel."""
    x = T.as_tensor(data)
    if wants_grad(x, *[p for ly in layers for p in ly.parameters()]):
        # training: layer by layer, every op recording its backward (bert.py:295-312)
        outs, h = [], data
        for i, ly in enumerate(layers):
            seqm = bool(final_seq_mask) and valid_length is not None and \
                (all_encodings or i == len(layers) - 1)
            h, _ = ly(h, None, _need_weights=False, _valid_length=valid_length,
                      _final_seq_mask=seqm)
            outs.append(h)
        return outs if all_encodings else h
    vlt, vlp, vlc = T.valid_length_arg(valid_length)
    l0 = layers[0]
    dev = T.require_device(x, vlt, l0.attn_qkv.weight.data)
    x = x.contiguous()
    if x.dtype != l0.attn_qkv.weight.dtype:
        raise TypeError('data dtype {} != parameter dtype {}'.format(x.dtype,
                                                                     l0.attn_qkv.weight.dtype))
    if layout == 'NT':
        B, L = x.shape[0], x.shape[1]
    else:
        L, B = x.shape[0], x.shape[1]
    n = len(layers)
    cfg = l0.cfg_struct(T.code(x))
    arr = (_lib.EncoderLayerParams * n)(*[ly.params_struct() for ly in layers])
    lib = _lib.load()
    ws = T.workspace(dev, lib.gnlp_encoder_workspace_bytes(ctypes.byref(cfg), B, L, 0))
    if all_encodings:
        outs = [torch.empty_like(x) for _ in range(n)]
        ptrs = (ctypes.c_void_p * n)(*[T.ptr(o) for o in outs])
        out_ptr = None
    else:
        outs = torch.empty_like(x)
        ptrs = None
        out_ptr = T.ptr(outs)
    folded = None
    if not all_encodings and x.dtype == torch.bfloat16 and (B * L) % 256 == 0:
        folded = _folded_params(layers, dev)
    if folded is not None:   # (the C side falls back to the plain stack for shapes it cannot fold)
        _lib.check(lib.gnlp_encoder_stack_forward_folded(
            out_ptr, T.ptr(x), arr, folded[0], n, ctypes.byref(cfg), B, L, vlp, vlc,
            int(bool(final_seq_mask) and vlp is not None), T.ptr(ws),
            0 if ws is None else ws.numel(), T.stream(dev)))
    else:
        _lib.check(lib.gnlp_encoder_stack_forward(
            out_ptr, T.ptr(x), arr, n, ctypes.byref(cfg), B, L, vlp, vlc,
            int(bool(final_seq_mask) and vlp is not None), ptrs, T.ptr(ws),
            0 if ws is None else ws.numel(), T.stream(dev)))
    if all_encodings:
        return [T.like_caller(o, data) for o in outs]
    return T.like_caller(outs, data)


class TransformerEncoder(nn.Module):
    """models/transformer.py:276-385."""

    def __init__(self, num_layers=6, recurrent=False, units=512, hidden_size=2048, num_heads=8,
                 activation_dropout=0.0, dropout=0.1, use_qkv_bias=True, attention_dropout=0.1,
                 layer_norm_eps=1E-5, data_norm=False, pre_norm=False, weight_initializer=None,
                 bias_initializer='zeros', activation='relu', dtype='float32', layout='NT'):
        super().__init__()
        self._dtype = dtype
        self.num_layers = num_layers
        self._recurrent = recurrent
        self._data_norm = data_norm
        self._pre_norm = pre_norm
        self._layout = layout
        assert layout in ['TN', 'NT'], 'Invalid layout received = {}. ' \
                                       'Only "TN" and "NT" are accepted!'.format(layout)
        if self._pre_norm:
            self.ln_final = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        if self._data_norm:
            self.ln_data = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        real_num_layers = 1 if recurrent else num_layers
        self.layers = nn.ModuleList([
            TransformerEncoderLayer(units=units, hidden_size=hidden_size, num_heads=num_heads,
                                    hidden_dropout_prob=dropout,
                                    attention_dropout_prob=attention_dropout,
                                    activation_dropout_prob=activation_dropout,
                                    use_qkv_bias=use_qkv_bias, layer_norm_eps=layer_norm_eps,
                                    weight_initializer=weight_initializer, pre_norm=pre_norm,
                                    activation=activation, layout=self._layout, dtype=dtype)
            for _ in range(real_num_layers)])

    @property
    def layout(self) -> str:
        return self._layout

    def forward(self, data, valid_length):
        """models/transformer.py:346-385."""
        out = data
        if self._data_norm:
            out = self.ln_data(out)
        seq = [self.layers[0] if self._recurrent else self.layers[i]
               for i in range(self.num_layers)]
        out = run_encoder_stack(seq, out, valid_length, self._layout, final_seq_mask=False)
        if self._pre_norm:
            out = self.ln_final(out)
        return out


class TransformerDecoderLayer(nn.Module):
    """models/transformer.py:389-598: causal self-attention, attention over the memory, FFN.
    Built from the encoder path's kernels: packed-QKV GEMM, the fused attention kernel with a byte
    mask (causal / memory masks are arbitrary masks to it; Lq != Lm for the memory), projection
    GEMMs with the residual fused, LayerNorm, PositionwiseFFN."""

    def __init__(self, units: int = 512, mem_units=None, hidden_size: int = 2048,
                 num_heads: int = 8, activation_dropout: float = 0.0, dropout: float = 0.1,
                 attention_dropout: float = 0.1, layer_norm_eps: float = 1E-5,
                 activation: str = 'relu', pre_norm: bool = False, use_qkv_bias: bool = True,
                 weight_initializer=None, bias_initializer='zeros', dtype='float32', layout='NT'):
        super().__init__()
        self._dtype = dtype
        self._units = units
        self._mem_units = units if mem_units is None else mem_units
        self._pre_norm = pre_norm
        self._num_heads = num_heads
        self._layout = layout
        assert layout in ['TN', 'NT'], 'Invalid layout received = {}. ' \
                                       'Only "TN" and "NT" are accepted!'.format(layout)
        attention_layout = 'NTK' if layout == 'NT' else 'TNK'
        if units % num_heads:
            raise ValueError('In Transformer, units should be divided exactly by the number of '
                             'heads. Received units={}, num_heads={}'.format(units, num_heads))
        kw = dict(weight_initializer=weight_initializer, dtype=dtype)
        self.attn_in_qkv = Dense(3 * units, in_units=units, use_bias=use_qkv_bias, **kw)
        self.self_attention = MultiHeadAttentionCell(query_units=units, num_heads=num_heads,
                                                     attention_dropout=attention_dropout,
                                                     dtype=dtype, layout=attention_layout)
        self.proj_in = Dense(units=units, in_units=units, use_bias=True, **kw)
        self.attn_inter_q = Dense(units, in_units=units, use_bias=use_qkv_bias, **kw)
        self.attn_inter_k = Dense(units, in_units=self._mem_units, use_bias=use_qkv_bias, **kw)
        self.attn_inter_v = Dense(units, in_units=self._mem_units, use_bias=use_qkv_bias, **kw)
        self.inter_attention = MultiHeadAttentionCell(query_units=units, num_heads=num_heads,
                                                      attention_dropout=attention_dropout,
                                                      dtype=dtype, layout=attention_layout)
        self.proj_inter = Dense(units=units, in_units=units, use_bias=True, **kw)
        self.ln_in = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        self.ln_inter = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        self.ffn = PositionwiseFFN(units=units, hidden_size=hidden_size, dropout=dropout,
                                   activation_dropout=activation_dropout,
                                   weight_initializer=weight_initializer,
                                   layer_norm_eps=layer_norm_eps, activation=activation,
                                   pre_norm=pre_norm, dtype=dtype)

    @property
    def layout(self) -> str:
        return self._layout

    def _heads(self, x):
        return x.reshape(x.shape[0], x.shape[1], self._num_heads, -1)

    def forward(self, data, mem, self_causal_mask, mem_attn_mask):
        """data (B, L, C) / (L, B, C); mem (B, Lm, C_mem) / (Lm, B, C_mem); self_causal_mask
        (B, L, L); mem_attn_mask (B, L, Lm) -> out like data.  models/transformer.py:513-598
        (with pre_norm the residual is the NORMALISED input, as the MXNet text has it)."""
        x = T.as_tensor(data)
        m = T.as_tensor(mem)
        if self._pre_norm:
            x = self.ln_in(x)
        qkv = T.as_tensor(self.attn_in_qkv(x))
        U = self._units
        q, k, v = qkv[..., :U], qkv[..., U:2 * U], qkv[..., 2 * U:]
        out, _ = self.self_attention(self._heads(q), self._heads(k), self._heads(v),
                                     self_causal_mask, _need_scores=False, _need_weights=False)
        out = self.proj_in(out, other=x)            # + data, fused in the GEMM epilogue
        if not self._pre_norm:
            out = self.ln_in(out)
        x = T.as_tensor(out)
        if self._pre_norm:
            x = self.ln_inter(x)
        out, _ = self.inter_attention(self._heads(T.as_tensor(self.attn_inter_q(x))),
                                      self._heads(T.as_tensor(self.attn_inter_k(m))),
                                      self._heads(T.as_tensor(self.attn_inter_v(m))),
                                      mem_attn_mask, _need_scores=False, _need_weights=False)
        out = self.proj_inter(out, other=x)
        if not self._pre_norm:
            out = self.ln_inter(out)
        return T.like_caller(T.as_tensor(self.ffn(out)), data)

    @property
    def state_batch_axis(self):
        return (0, 0) if self.layout == 'NT' else (1, 1)

    def init_states(self, batch_size, ctx, dtype='float32'):
        """Empty key / value caches (B, 0, H, C) / (0, B, H, C).  models/transformer.py:607-635."""
        d = self._units // self._num_heads
        shape = ((batch_size, 0, self._num_heads, d) if self.layout == 'NT'
                 else (0, batch_size, self._num_heads, d))
        dt = T.torch_dtype(dtype)
        return (torch.zeros(shape, dtype=dt, device=ctx), torch.zeros(shape, dtype=dt, device=ctx))

    def incremental_decode(self, data, states, mem, mem_valid_length, mem_attn_mask=None):
        """One target position: data (B, C_in); states = (keys, values) of the previous positions
        -> (out (B, C_out), (keys, values) with this position appended).
        models/transformer.py:637-734.  The self-attention needs no mask: every cached position
        precedes the new one."""
        x = T.as_tensor(data)
        if self._pre_norm:
            x = self.ln_in(x)
        time_axis = 1 if self.layout == 'NT' else 0
        x = T.as_tensor(x).unsqueeze(time_axis)
        prev_key, prev_value = states
        if mem_attn_mask is None:
            mem_attn_mask = gen_mem_attn_mask(mem, mem_valid_length, x, None, dtype=self._dtype,
                                              layout=self.layout)
        qkv = T.as_tensor(self.attn_in_qkv(x))
        U = self._units
        q, k, v = qkv[..., :U], qkv[..., U:2 * U], qkv[..., 2 * U:]
        new_key = torch.cat([T.as_tensor(prev_key), self._heads(k)], dim=time_axis)
        new_value = torch.cat([T.as_tensor(prev_value), self._heads(v)], dim=time_axis)
        out, _ = self.self_attention(self._heads(q), new_key, new_value, None,
                                     _need_scores=False, _need_weights=False)
        out = self.proj_in(out, other=x)
        if not self._pre_norm:
            out = self.ln_in(out)
        x = T.as_tensor(out)
        if self._pre_norm:
            x = self.ln_inter(x)
        mem_k, mem_v, kept = memory_kv(prev_key, mem, self.attn_inter_k, self.attn_inter_v)
        if kept is not None:
            new_key._gnlp_memory_kv = kept       # the projections travel with the decoding state
        out, _ = self.inter_attention(self._heads(T.as_tensor(self.attn_inter_q(x))),
                                      self._heads(mem_k), self._heads(mem_v),
                                      mem_attn_mask, _need_scores=False, _need_weights=False)
        out = self.proj_inter(out, other=x)
        if not self._pre_norm:
            out = self.ln_inter(out)
        out = T.as_tensor(self.ffn(out))
        return T.like_caller(out.reshape(-1, out.shape[-1]), data), (new_key, new_value)


class TransformerDecoder(nn.Module):
    """models/transformer.py:738-959."""

    def __init__(self, num_layers=6, recurrent=False, units=512, mem_units=None, hidden_size=2048,
                 num_heads=8, max_shift=None, activation_dropout=0.0, dropout=0.1,
                 attention_dropout=0.1, layer_norm_eps=1E-5, data_norm=False, pre_norm=False,
                 weight_initializer=None, bias_initializer='zeros', activation='relu',
                 dtype='float32', layout='NT'):
        super().__init__()
        self._dtype = dtype
        self._units = units
        self._mem_units = mem_units
        self.num_layers = num_layers
        self.recurrent = recurrent
        self.max_shift = max_shift
        self._data_norm = data_norm
        self._pre_norm = pre_norm
        self._layout = layout
        assert layout in ['TN', 'NT'], 'Invalid layout received = {}. ' \
                                       'Only "TN" and "NT" are accepted!'.format(layout)
        if self._data_norm:
            self.ln_data = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        if self._pre_norm:
            self.ln_final = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        real_num_layers = 1 if recurrent else num_layers
        self.layers = nn.ModuleList([
            TransformerDecoderLayer(units=units, mem_units=mem_units, hidden_size=hidden_size,
                                    num_heads=num_heads, activation_dropout=activation_dropout,
                                    dropout=dropout, attention_dropout=attention_dropout,
                                    layer_norm_eps=layer_norm_eps,
                                    weight_initializer=weight_initializer,
                                    bias_initializer=bias_initializer, activation=activation,
                                    pre_norm=pre_norm, layout=layout, dtype=dtype)
            for _ in range(real_num_layers)])

    @property
    def units(self):
        return self._units

    @property
    def layout(self) -> str:
        return self._layout

    def forward(self, data, valid_length, mem_data, mem_valid_length):
        """models/transformer.py:789-840."""
        out = data
        if self._data_norm:
            out = self.ln_data(out)
        self_causal_mask = gen_self_attn_mask(data, valid_length, dtype=self._dtype,
                                              attn_type='causal', layout=self._layout)
        mem_attn_mask = gen_mem_attn_mask(mem_data, mem_valid_length, data, valid_length,
                                          dtype=self._dtype, layout=self._layout)
        for i in range(self.num_layers):
            layer = self.layers[0] if self.recurrent else self.layers[i]
            out = layer(out, mem_data, self_causal_mask, mem_attn_mask)
        if self._pre_norm:
            out = self.ln_final(out)
        return out

    @property
    def state_batch_axis(self):
        return [(self.layers[0] if self.recurrent else self.layers[i]).state_batch_axis
                for i in range(self.num_layers)]

    def init_states(self, batch_size, ctx, dtype='float32'):
        """models/transformer.py:852-882."""
        return [(self.layers[0] if self.recurrent else self.layers[i]).init_states(
            batch_size=batch_size, ctx=ctx, dtype=dtype) for i in range(self.num_layers)]

    def incremental_decode(self, data, states, mem, mem_valid_length):
        """data (B, C_in), one cache pair per layer -> (out (B, C_out), new caches).
        models/transformer.py:884-959: the memory mask (B, 1, mem_length) is built once.
        The memory's key / value projections are kept in the states between steps
        (layers.memory_kv): they follow optimizer steps and copy_() of the weights, not writes
        through ``.data`` -- call layers.drop_memory_kv(states) after those."""
        out = data
        if self._data_norm:
            out = self.ln_data(out)
        time_axis = 0 if self.layout == 'TN' else 1
        step = T.as_tensor(data).unsqueeze(time_axis)
        mem_attn_mask = gen_mem_attn_mask(mem, mem_valid_length, step, None, dtype=self._dtype,
                                          layout=self._layout)
        new_states = []
        for i in range(self.num_layers):
            layer = self.layers[0] if self.recurrent else self.layers[i]
            out, new_state = layer.incremental_decode(out, states[i], mem, mem_valid_length,
                                                      mem_attn_mask)
            new_states.append(new_state)
        if self._pre_norm:
            out = self.ln_final(out)
        return out, new_states
