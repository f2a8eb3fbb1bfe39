"""T5Block / T5Encoder / T5Decoder / T5Model -- mirrors ``gluonnlp/models/t5.py`` (reference
src/gluonnlp/models/t5.py:173-350, 374-572, 576-711, 715-964, 968-1189).  The encoder stack is one
C call; the decoder runs block by block on the same kernels: causal and memory
masks are byte masks to the fused attention kernel, Lq != Lm for the memory, Lq = 1 when decoding
incrementally."""
import ctypes

import torch
from torch import nn

from .. import _lib
from .. import _tensor as T
from ..attention_cell import (MultiHeadAttentionCell, RelAttentionScoreCell, gen_mem_attn_mask,
                              gen_rel_position, gen_self_attn_mask, mask_predicate)
from ..layers import Dense, Embedding, PositionwiseFFN, RMSNorm, ffn_params_struct, memory_kv

__all__ = ['T5Block', 'T5Encoder', 'T5Decoder', 'T5Model']


def _normal(std):
    def init(t):
        t.normal_(0.0, std)
    return init


# developer knob (A/B only): GNLP_T5_PACK=0 issues the three projections as three GEMMs
_PACK_QKV = __import__('os').environ.get('GNLP_T5_PACK', '1') != '0'


class T5Block(nn.Module):
    """Pre-RMSNorm block with bias-free q/k/v/proj, ``scaled=False`` attention and the shared
    relative-position bias as ``edge_scores``.  models/t5.py:173-350, forward :492-572."""

    def __init__(self, d_model, d_kv, d_ff, is_decoder, num_heads=12, dropout_prob=0.1,
                 layer_norm_eps=1E-6, activation='relu', init_factor=1.0, layout='NT',
                 dtype='float32'):
        super().__init__()
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
        if self._is_decoder:
            self.cross_attn_layer_norm = RMSNorm(in_channels=d_model, center=False, scale=True,
                                                 variance_epsilon=layer_norm_eps, dtype=dtype)
            self.cross_attn_q = Dense(self._inner_dim, d_model, use_bias=False, dtype=dtype,
                                      weight_initializer=_normal((d_model * d_kv) ** -0.5 * init_factor))
            self.cross_attn_k = Dense(self._inner_dim, d_model, use_bias=False, dtype=dtype,
                                      weight_initializer=_normal(d_model ** -0.5 * init_factor))
            self.cross_attn_v = Dense(self._inner_dim, d_model, use_bias=False, dtype=dtype,
                                      weight_initializer=_normal(d_model ** -0.5 * init_factor))
            self.cross_attn = MultiHeadAttentionCell(
                query_units=self._inner_dim, num_heads=num_heads, attention_dropout=dropout_prob,
                scaled=False, normalized=False, dtype=dtype,
                layout='NTK' if layout == 'NT' else 'TNK', use_einsum=False)
            self.cross_attn_proj = Dense(d_model, self._inner_dim, use_bias=False, dtype=dtype,
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

    def packed_qkv(self):
        """[q; k; v] weights packed row-wise into one (3 * inner, d_model) buffer, so that the
        block's three projections are ONE GEMM of N = 3 * inner (the reference keeps three Dense
        layers, models/t5.py:296-318, and so do the parameter names here).  The buffer is not a
        copy that could go stale: the three parameters are re-seated as row slices OF it, so
        every in-place edit (load_state_dict, ``p.data.copy_``, an optimizer step -- version
        counters do not see edits through ``.data``) lands in the packed storage itself; a
        parameter whose storage was replaced (``.to()``, ``.cuda()``, ``p.data = ...``) no longer
        points into the buffer, which is then rebuilt."""
        ps = [self.self_attn_q.weight, self.self_attn_k.weight, self.self_attn_v.weight]
        buf = getattr(self, '_qkv_packed', None)
        if buf is not None:
            row_bytes = buf.shape[1] * buf.element_size()
            at, ok = buf.data_ptr(), True
            for p in ps:
                ok = ok and (p.data_ptr() == at and p.dtype == buf.dtype and p.device == buf.device
                             and p.is_contiguous() and p.shape[1] == buf.shape[1])
                at += p.shape[0] * row_bytes
            if ok:
                return buf
        buf = torch.cat([p.data for p in ps], dim=0).contiguous()
        at = 0
        for p in ps:
            p.data = buf[at:at + p.shape[0]]
            at += p.shape[0]
        object.__setattr__(self, '_qkv_packed', buf)
        return buf

    def params_struct(self):
        p = _lib.T5BlockParams()
        p.attn_ln_g = T.ptr(self.self_attn_layer_norm.gamma.data)
        p.attn_ln_b = T.ptr(self.self_attn_layer_norm.beta.data)
        p.q_w = T.ptr(self.self_attn_q.weight.data)
        p.k_w = T.ptr(self.self_attn_k.weight.data)
        p.v_w = T.ptr(self.self_attn_v.weight.data)
        if _PACK_QKV:
            p.qkv_w = T.ptr(self.packed_qkv())
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

    @property
    def state_batch_axis(self):
        assert self._is_decoder, 'state_batch_axis() is available for decoder only.'
        return (0, 0) if self.layout == 'NT' else (1, 1)

    def _init_key_value(self, batch_size, ctx, dtype='float32'):
        """models/t5.py:374-381: empty (B, 0, H, d_kv) / (0, B, H, d_kv) key and value."""
        assert self._is_decoder, '_init_key_value() is available for decoder only.'
        shape = ((batch_size, 0, self._num_heads, self._d_kv) if self.layout == 'NT'
                 else (0, batch_size, self._num_heads, self._d_kv))
        dt = T.torch_dtype(dtype)
        return (torch.zeros(shape, dtype=dt, device=ctx), torch.zeros(shape, dtype=dt, device=ctx))

    def transpose_for_scores(self, x):
        """(B, L, inner) -> (B, L, H, d_kv) or (L, B, inner) -> (L, B, H, d_kv).  t5.py:383-389."""
        x = T.as_tensor(x)
        return x.reshape(x.shape[0], x.shape[1], self._num_heads, -1)

    def _cross_and_ffn(self, hidden_states, mem_states, mem_attn_mask, states=None):
        out = self.cross_attn_layer_norm(hidden_states)
        q = self.transpose_for_scores(self.cross_attn_q(out))
        if states is not None:   # incremental decoding: (previous, new) key cache; the memory's
            # projections travel with the decoding state instead of being recomputed per step
            k, v, kept = memory_kv(states[0], mem_states, self.cross_attn_k, self.cross_attn_v)
            if kept is not None:
                states[1]._gnlp_memory_kv = kept
        else:
            k, v = self.cross_attn_k(mem_states), self.cross_attn_v(mem_states)
        k, v = self.transpose_for_scores(k), self.transpose_for_scores(v)
        out, _ = self.cross_attn(q, k, v, mem_attn_mask, _need_scores=False, _need_weights=False)
        hidden_states = self.cross_attn_proj(out, other=hidden_states, combine='add')
        return self.ffn(hidden_states)

    def incremental_decode(self, step_hidden_states, step_position_embeddings, past_key_value,
                           mem_states, step_mem_attn_mask):
        """One target position.  step_hidden_states (B, 1, d_model) / (1, B, d_model);
        step_position_embeddings (H, 1, L_past + 1); past_key_value two (B, L_past, H, d_kv) /
        (L_past, B, H, d_kv); step_mem_attn_mask (B, 1, L_src) -> (step_hidden_states,
        (key, value)) with the step appended.  models/t5.py:391-490."""
        assert self._is_decoder, 'incremental_decode() is available for decoder only.'
        x = T.as_tensor(step_hidden_states)
        out = self.self_attn_layer_norm(x)
        q = self.transpose_for_scores(self.self_attn_q(out))
        k = self.transpose_for_scores(self.self_attn_k(out))
        v = self.transpose_for_scores(self.self_attn_v(out))
        self_key = torch.cat([T.as_tensor(past_key_value[0]), k], dim=self._time_axis)
        self_value = torch.cat([T.as_tensor(past_key_value[1]), v], dim=self._time_axis)
        out, _ = self.self_attn(q, self_key, self_value, None, step_position_embeddings,
                                _need_scores=False, _need_weights=False)
        x = self.self_attn_proj(out, other=x, combine='add')
        x = self._cross_and_ffn(x, mem_states, step_mem_attn_mask,
                                states=(past_key_value[0], self_key))
        return T.like_caller(T.as_tensor(x), step_hidden_states), (self_key, self_value)

    def forward(self, hidden_states, self_attn_mask, position_embeddings, mem_states=None,
                mem_attn_mask=None):
        """models/t5.py:492-572."""
        m = T.as_tensor(self_attn_mask)
        if not self._is_decoder:
            pred = mask_predicate(m) if m is not None else ('ones', None)
            if pred is not None:
                vl = pred[1] if pred[0] == 'full' else None
                return run_t5_blocks([self], hidden_states, vl, position_embeddings, None,
                                     self._layout)
        # arbitrary dense mask / decoder: op by op through the same kernels (t5.py:537-571)
        out = self.self_attn_layer_norm(hidden_states)
        q = self.transpose_for_scores(self.self_attn_q(out))
        k = self.transpose_for_scores(self.self_attn_k(out))
        v = self.transpose_for_scores(self.self_attn_v(out))
        out, _ = self.self_attn(q, k, v, m, position_embeddings, _need_scores=False,
                                _need_weights=False)
        hidden_states = self.self_attn_proj(out, other=hidden_states, combine='add')
        if self._is_decoder:
            return T.like_caller(T.as_tensor(self._cross_and_ffn(hidden_states, mem_states,
                                                                 mem_attn_mask)), hidden_states)
        return self.ffn(hidden_states)


# ---- RMSNorms folded into the GEMMs that consume them (bf16 encoder blocks with a relu FFN) -------
# gnlp_t5_encoder_forward_folded takes, per block, W' = bf16(gamma * W) (+ c, b') of the packed
# q | k | v weights (with the block's self_attn_layer_norm) and of ffn.ffn_1 (with ffn.layer_norm):
# functions of the parameters only, computed on the device once per parameter version and kept with
# the first block.  Key and limit as for the encoder stacks (models/transformer.py): an in-place
# write through ``.data`` is not seen -- drop_folded_parameters(model) after such an edit.
_T5_FOLD_ATTR = '_gnlp_folded_params'   # (the attribute drop_folded_parameters deletes)


def _t5_fold_sources(blocks):
    for b in blocks:
        yield b.packed_qkv()
        for t in (b.self_attn_layer_norm.gamma, getattr(b.self_attn_layer_norm, 'beta', None),
                  b.ffn.ffn_1.weight, b.ffn.layer_norm.gamma,
                  getattr(b.ffn.layer_norm, 'beta', None)):
            yield t


def _t5_folded_params(blocks, dev):
    """(ctypes array, tensors kept alive), or None when the blocks cannot fold at all."""
    b0 = blocks[0]
    if not _PACK_QKV or b0.self_attn_q.weight.dtype != torch.bfloat16:
        return None
    for b in blocks:
        if b._is_decoder or b._activation != 'relu' or b.ffn.ffn_1.bias is not None:
            return None
    src = list(_t5_fold_sources(blocks))
    key = tuple(None if t is None else (t.data_ptr(), t._version) for t in src)
    cached = getattr(b0, _T5_FOLD_ATTR, None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    lib = _lib.load()
    stream = T.stream(dev)
    keep = []
    arr = (_lib.T5BlockFolded * len(blocks))()

    def fold(w, norm):
        w = w.data.contiguous()
        N, K = w.shape
        wf = torch.empty_like(w)
        c = torch.empty(N, dtype=torch.float32, device=dev)
        bf = torch.empty(N, dtype=torch.float32, device=dev)
        beta = getattr(norm, 'beta', None)
        _lib.check(lib.gnlp_fold_norm_dense(
            T.ptr(wf), T.ptr(c), T.ptr(bf), T.ptr(w), None, T.ptr(norm.gamma.data),
            T.ptr(beta.data) if beta is not None else None, _lib.BF16, N, K, 0, 1.0, stream))
        keep.extend([wf, c, bf, w])
        return T.ptr(wf), T.ptr(c), T.ptr(bf)

    for i, b in enumerate(blocks):
        arr[i].qkv_wf, arr[i].qkv_c, arr[i].qkv_bf = fold(b.packed_qkv(), b.self_attn_layer_norm)
        arr[i].ffn1_wf, arr[i].ffn1_c, arr[i].ffn1_bf = fold(b.ffn.ffn_1.weight, b.ffn.layer_norm)
    keep.extend(t.data for t in src if t is not None)
    object.__setattr__(b0, _T5_FOLD_ATTR, (key, arr, keep))
    return arr, keep


def run_t5_blocks(blocks, hidden_states, valid_length, position_bias, final_ln, layout,
                  toeplitz=False):
    """All blocks (+ the final RMSNorm) in ONE C call (gnlp_t5_encoder_forward).  ``toeplitz``:
    ``position_bias`` is the (H, 2L - 1) table of RelAttentionScoreCell.self_attn_bias_table."""
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
        if toeplitz:
            if tuple(pb.shape) != (b0._num_heads, 2 * L - 1):
                raise ValueError('bias table {} != (heads, 2L - 1) = {}'.format(
                    tuple(pb.shape), (b0._num_heads, 2 * L - 1)))
            pb = pb.contiguous()
        else:
            pb = torch.broadcast_to(pb, (b0._num_heads, L, L)).contiguous()
    n = len(blocks)
    cfg = b0.cfg_struct(T.code(x))
    cfg.bias_toeplitz = int(bool(toeplitz))
    arr = (_lib.T5BlockParams * n)(*[b.params_struct() for b in blocks])
    lib = _lib.load()
    ws = T.workspace(dev, lib.gnlp_t5_workspace_bytes(ctypes.byref(cfg), B, L))
    out = torch.empty_like(x)
    fin_g = T.ptr(final_ln.gamma.data) if final_ln is not None else None
    fin_b = T.ptr(final_ln.beta.data) if final_ln is not None and \
        getattr(final_ln, 'beta', None) is not None else None
    folded = _t5_folded_params(blocks, dev) if (B * L) % 256 == 0 else None
    if folded is not None:
        _lib.check(lib.gnlp_t5_encoder_forward_folded(
            T.ptr(out), T.ptr(x), arr, folded[0], n, fin_g, fin_b, ctypes.byref(cfg), B, L, vlp,
            vlc, T.ptr(pb), T.ptr(ws), 0 if ws is None else ws.numel(), T.stream(dev)))
    else:
        _lib.check(lib.gnlp_t5_encoder_forward(
            T.ptr(out), T.ptr(x), arr, n, fin_g, fin_b, ctypes.byref(cfg), B, L, vlp, vlc, T.ptr(pb),
            T.ptr(ws), 0 if ws is None else ws.numel(), T.stream(dev)))
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
        """models/t5.py:666-711.  The reference materialises the (H, L, L) bias and every layer
        and batch element re-reads it; it depends on j - i only, so the stack is given the
        (H, 2L - 1) table instead (one fused rel-position + bucket + gather launch) and the
        attention kernel indexes it with a row stride of -1."""
        x = T.as_tensor(hidden_states)
        L = x.shape[self._time_axis]
        table = self.relative_position_encoder.self_attn_bias_table(L)
        return run_t5_blocks(list(self.layers), hidden_states, valid_length, table,
                             self.final_layer_norm, self._layout, toeplitz=True)


class T5Decoder(nn.Module):
    """models/t5.py:715-964: unidirectional relative-position buckets, causal self-attention,
    attention over the encoder memory."""

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
            query_units=self._inner_dim, num_heads=num_heads, method='t5', bidirectional=False,
            embed_initializer=_normal(d_model ** -0.5 * init_factor),
            layout='NTK' if layout == 'NT' else 'TNK', dtype=dtype)
        self.layers = nn.ModuleList([
            T5Block(d_model=d_model, d_kv=d_kv, d_ff=d_ff, is_decoder=True, num_heads=num_heads,
                    dropout_prob=dropout_prob, layer_norm_eps=layer_norm_eps,
                    activation=activation, init_factor=init_factor, layout=layout, dtype=dtype)
            for _ in range(num_layers)])
        self.final_layer_norm = RMSNorm(in_channels=d_model, center=False, scale=True,
                                        variance_epsilon=layer_norm_eps, dtype=dtype)

    @property
    def layout(self):
        return self._layout

    @property
    def state_batch_axis(self):
        return list(layer.state_batch_axis for layer in self.layers)

    def _init_key_values(self, batch_size, ctx, dtype='float32'):
        return list(layer._init_key_value(batch_size, ctx, dtype) for layer in self.layers)

    def incremental_decode(self, step_hidden_states, position, past_key_values, mem_states,
                           mem_valid_length):
        """models/t5.py:812-893.  ``position`` (B,) is accepted as in the reference, which derives
        the step's relative positions from the length of the cached keys instead (:863-870)."""
        past = T.as_tensor(past_key_values[0][0])
        L = past.shape[self._time_axis] + 1
        # last row of the (L, L) bias = relative positions (mem_j - (L - 1)) of the new query =
        # the first L entries of the bias' Toeplitz table (bias[h, i, j] = table[h, L - 1 + j - i]):
        # O(L) per step where the (H, L, L) tensor is O(L^2)
        step_position_embeddings = self.relative_position_encoder.self_attn_bias_table(L)[
            :, :L].unsqueeze(1).contiguous()
        step_mem_attn_mask = gen_mem_attn_mask(mem_states, mem_valid_length, step_hidden_states,
                                               dtype=self._dtype, layout=self.layout)
        present_key_values = []
        for i, layer in enumerate(self.layers):
            step_hidden_states, present = layer.incremental_decode(
                step_hidden_states, step_position_embeddings, past_key_values[i], mem_states,
                step_mem_attn_mask)
            present_key_values.append(present)
        return self.final_layer_norm(step_hidden_states), present_key_values

    def forward(self, hidden_states, valid_length, mem_states, mem_valid_length):
        """models/t5.py:895-964."""
        x = T.as_tensor(hidden_states)
        L = x.shape[self._time_axis]
        position_embeddings = self.relative_position_encoder.self_attn_bias(L)
        self_attn_mask = gen_self_attn_mask(hidden_states, valid_length, dtype=self._dtype,
                                            attn_type='causal', layout=self.layout)
        mem_attn_mask = gen_mem_attn_mask(mem_states, mem_valid_length, hidden_states,
                                          valid_length, dtype=self._dtype, layout=self.layout)
        for layer in self.layers:
            hidden_states = layer(hidden_states, self_attn_mask, position_embeddings, mem_states,
                                  mem_attn_mask)
        return self.final_layer_norm(hidden_states)


class T5Model(nn.Module):
    """T5 encoder-decoder (models/t5.py:968-1189): one input embedding shared by both stacks."""

    def __init__(self, vocab_size=32128, d_model=768, d_kv=64, d_ff=3072, num_layers=12,
                 num_heads=12, dropout_prob=0.1, layer_norm_eps=1E-6, activation='relu',
                 init_factor=1.0, layout='NT', dtype='float32'):
        super().__init__()
        assert vocab_size > 0, 'Vocab size {} is not valid.'.format(vocab_size)
        self._vocab_size = vocab_size
        self._d_model, self._d_kv, self._d_ff = d_model, d_kv, d_ff
        self._num_layers = num_layers
        self._num_heads = num_heads
        self._inner_dim = num_heads * d_kv
        self._activation = activation
        self._init_factor = init_factor
        self._dtype = dtype
        assert layout in ['TN', 'NT'], \
            'Invalid layout: {}. Only "TN" and "NT" are supported.'.format(layout)
        self._layout = layout
        self._time_axis = 1 if self.layout == 'NT' else 0
        self.input_embedding_layer = Embedding(input_dim=vocab_size, output_dim=d_model,
                                               weight_initializer=_normal(1.0 * init_factor),
                                               dtype=dtype)
        kw = dict(d_model=d_model, d_kv=d_kv, d_ff=d_ff, num_layers=num_layers,
                  num_heads=num_heads, dropout_prob=dropout_prob, layer_norm_eps=layer_norm_eps,
                  activation=activation, init_factor=init_factor, layout=layout, dtype=dtype)
        self.encoder = T5Encoder(**kw)
        self.decoder = T5Decoder(**kw)

    @property
    def activation(self):
        return self._activation

    @property
    def layout(self):
        return self._layout

    @property
    def num_layers(self):
        return self._num_layers

    @property
    def vocab_size(self):
        return self._vocab_size

    def encode(self, src_data, src_valid_length):
        """token ids (B, L_src) / (L_src, B) -> (B, L_src, d_model) / (L_src, B, d_model)."""
        return self.encoder(self.input_embedding_layer(src_data), src_valid_length)

    def decode(self, tgt_data, tgt_valid_length, mem_states, mem_valid_length):
        return self.decoder(self.input_embedding_layer(tgt_data), tgt_valid_length, mem_states,
                            mem_valid_length)

    def forward(self, src_data, src_valid_length, tgt_data, tgt_valid_length):
        enc_out = self.encode(src_data, src_valid_length)
        return self.decode(tgt_data, tgt_valid_length, enc_out, src_valid_length)
