"""BertTransformer and BertModel -- mirror ``gluonnlp/models/bert.py`` (reference
src/gluonnlp/models/bert.py:213-315 the encoder stack, :318-529 the backbone: embedding front-end
and pooler).  MLM / NSP heads are out of scope."""
import torch
from torch import nn

from .. import _tensor as T
from ..layers import (Dense, Embedding, LayerNorm, PositionalEmbedding, embedding_forward)
from .transformer import TransformerEncoderLayer, run_encoder_stack

__all__ = ['BertTransformer', 'BertModel']


def cls_pooling(pooler, sequence, layout):
    """apply_pooling (models/bert.py:506-529): the first token of every sequence through the
    pooler Dense (its activation fused); ``pooler`` None returns the CLS rows themselves."""
    seq = T.as_tensor(sequence)
    if layout == 'NT':
        cls = seq[:, 0, :]
    elif layout == 'TN':
        cls = seq[0, :, :]
    else:
        raise NotImplementedError
    if pooler is None:
        return T.like_caller(cls, sequence)
    return T.like_caller(pooler(cls), sequence)


class _EncoderStack(nn.Module):
    """Shared control flow of BertTransformer / ElectraEncoder / RobertaEncoder."""
    _final_seq_mask = True
    _layers_attr = 'all_layers'

    def _build(self, units, hidden_size, num_layers, num_heads, attention_dropout_prob,
               hidden_dropout_prob, layer_norm_eps, weight_initializer, activation, layout, dtype):
        layers = nn.ModuleList([
            TransformerEncoderLayer(units=units, hidden_size=hidden_size, num_heads=num_heads,
                                    attention_dropout_prob=attention_dropout_prob,
                                    hidden_dropout_prob=hidden_dropout_prob,
                                    layer_norm_eps=layer_norm_eps,
                                    weight_initializer=weight_initializer, activation=activation,
                                    layout=layout, dtype=dtype)
            for _ in range(num_layers)])
        setattr(self, self._layers_attr, layers)

    @property
    def layout(self):
        return self._layout

    def _run(self, data, valid_length):
        layers = list(getattr(self, self._layers_attr))
        if not self._output_attention:
            # hot path: the whole stack is one C call, mask = valid_length predicate in-kernel
            res = run_encoder_stack(layers, data, valid_length, self._layout,
                                    final_seq_mask=self._final_seq_mask,
                                    all_encodings=self._output_all_encodings)
            return res, []
        # output_attention: layer by layer, materialising (B, H, L, L) weights
        out = data
        all_enc, extra = [], []
        for i, layer in enumerate(layers):
            last = (i == len(layers) - 1)
            seqm = self._final_seq_mask and (self._output_all_encodings or last) \
                and valid_length is not None
            out, w = layer(out, None, _need_weights=True, _valid_length=valid_length,
                           _final_seq_mask=seqm)
            if self._output_all_encodings:
                all_enc.append(out)
            extra.append(w)
        return (all_enc if self._output_all_encodings else out), extra


class BertTransformer(_EncoderStack):
    """models/bert.py:213-315: N post-norm encoder layers, mask built once from valid_length,
    final ``npx.sequence_mask`` (:311-312)."""

    def __init__(self, units: int = 512, hidden_size: int = 2048, num_layers: int = 6,
                 num_heads: int = 8, attention_dropout_prob: float = 0.,
                 hidden_dropout_prob: float = 0., output_attention: bool = False, dtype='float32',
                 output_all_encodings: bool = False, layer_norm_eps: float = 1E-12,
                 weight_initializer=None, bias_initializer='zeros', activation='gelu',
                 layout='NT'):
        super().__init__()
        assert units % num_heads == 0, \
            'In BertTransformer, The units should be divided exactly ' \
            'by the number of heads. Received units={}, num_heads={}'.format(units, num_heads)
        self._dtype = dtype
        self._num_layers = num_layers
        self._output_attention = output_attention
        self._output_all_encodings = output_all_encodings
        self._layout = layout
        self._build(units, hidden_size, num_layers, num_heads, attention_dropout_prob,
                    hidden_dropout_prob, layer_norm_eps, weight_initializer, activation, layout,
                    dtype)

    def forward(self, data, valid_length):
        """-> (out | [all encodings], [attention weights])   models/bert.py:259-315."""
        return self._run(data, valid_length)


class BertModel(nn.Module):
    """models/bert.py:318-529: token ids -> (contextual embeddings, pooled CLS output).
    Parameter paths as in the reference: ``word_embed.weight``, ``token_type_embed.weight``,
    ``token_pos_embed._embed.weight``, ``embed_layer_norm.{gamma,beta}``, ``encoder.all_layers.*``,
    ``pooler.{weight,bias}``."""

    def __init__(self, vocab_size=30000, units=768, hidden_size=3072, num_layers=12, num_heads=12,
                 max_length=512, hidden_dropout_prob=0., attention_dropout_prob=0.,
                 num_token_types=2, pos_embed_type='learned', activation='gelu',
                 layer_norm_eps=1E-12, embed_initializer=None, weight_initializer=None,
                 bias_initializer='zeros', dtype='float32', use_pooler=True, layout='NT',
                 compute_layout='auto'):
        super().__init__()
        self._dtype = dtype
        self.use_pooler = use_pooler
        self.pos_embed_type = pos_embed_type
        self.num_token_types = num_token_types
        self.vocab_size = vocab_size
        self.units = units
        self.max_length = max_length
        self.activation = activation
        self.layer_norm_eps = layer_norm_eps
        self._layout = layout
        if compute_layout is None or compute_layout == 'auto':
            self._compute_layout = layout
        else:
            self._compute_layout = compute_layout
        self.encoder = BertTransformer(
            units=units, hidden_size=hidden_size, num_layers=num_layers, num_heads=num_heads,
            attention_dropout_prob=attention_dropout_prob,
            hidden_dropout_prob=hidden_dropout_prob, output_attention=False,
            output_all_encodings=False, activation=activation, layer_norm_eps=layer_norm_eps,
            weight_initializer=weight_initializer, bias_initializer=bias_initializer, dtype=dtype,
            layout=self._compute_layout)
        self.word_embed = Embedding(vocab_size, units, weight_initializer=embed_initializer,
                                    dtype=dtype)
        self.embed_layer_norm = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        self.token_type_embed = Embedding(num_token_types, units,
                                          weight_initializer=weight_initializer, dtype=dtype)
        if pos_embed_type is not None:
            self.token_pos_embed = PositionalEmbedding(units=units, max_length=max_length,
                                                       dtype=dtype, method=pos_embed_type)
        if use_pooler:
            self.pooler = Dense(units=units, in_units=units, flatten=False, activation='tanh',
                                weight_initializer=weight_initializer,
                                bias_initializer=bias_initializer, dtype=dtype)

    @property
    def layout(self):
        return self._layout

    def get_initial_embedding(self, inputs, token_types=None):
        """models/bert.py:458-504 (one kernel: three gathers, two adds, LayerNorm)."""
        pos_w = (self.token_pos_embed.table(inputs, self._layout)
                 if self.pos_embed_type is not None else None)
        return embedding_forward(inputs, token_types, self.word_embed.weight.data,
                                 self.token_type_embed.weight.data, pos_w,
                                 self.embed_layer_norm.gamma.data, self.embed_layer_norm.beta.data,
                                 self.layer_norm_eps, self._layout)

    def apply_pooling(self, sequence):
        """models/bert.py:506-529."""
        return cls_pooling(self.pooler, sequence, self._layout)

    def _encode(self, embedding, valid_length):
        emb = T.as_tensor(embedding)
        if self._compute_layout != self._layout:
            ctx, extra = self.encoder(torch.swapaxes(emb, 0, 1), valid_length)
            ctx = torch.swapaxes(T.as_tensor(ctx), 0, 1)
        else:
            ctx, extra = self.encoder(emb, valid_length)
        return ctx, extra

    def forward(self, inputs, token_types, valid_length):
        """-> (contextual_embedding, pooled_output) or contextual_embedding
        (models/bert.py:404-456)."""
        ctx, _ = self._encode(self.get_initial_embedding(inputs, token_types), valid_length)
        if self.use_pooler:
            return ctx, self.apply_pooling(ctx)
        return ctx
