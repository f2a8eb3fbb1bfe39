"""ElectraEncoder and ElectraModel -- mirror ``gluonnlp/models/electra.py`` (reference
src/gluonnlp/models/electra.py:175-307 the stack, :310-510 the backbone)."""
import torch
from torch import nn

from .. import _tensor as T
from ..layers import Dense, Embedding, LayerNorm, PositionalEmbedding, embedding_forward
from .bert import _EncoderStack, cls_pooling

__all__ = ['ElectraEncoder', 'ElectraModel']


class ElectraEncoder(_EncoderStack):
    """Same control flow as BertTransformer, layers live under ``all_encoder_layers``."""
    _layers_attr = 'all_encoder_layers'

    def __init__(self, units=512, hidden_size=2048, num_layers=6, num_heads=8,
                 attention_dropout_prob=0., hidden_dropout_prob=0., output_attention=False,
                 dtype='float32', output_all_encodings=False, layer_norm_eps=1E-12,
                 weight_initializer=None, bias_initializer='zeros', activation='gelu',
                 layout='NT'):
        super().__init__()
        assert units % num_heads == 0, \
            'In ElectraEncoder, The units should be divisible ' \
            'by the number of heads. Received units={}, num_heads={}'.format(units, num_heads)
        self._dtype = dtype
        self._layout = layout
        self._num_layers = num_layers
        self._output_attention = output_attention
        self._output_all_encodings = output_all_encodings
        self._build(units, hidden_size, num_layers, num_heads, attention_dropout_prob,
                    hidden_dropout_prob, layer_norm_eps, weight_initializer, activation, layout,
                    dtype)

    def forward(self, data, valid_length):
        """models/electra.py:249-307."""
        return self._run(data, valid_length)


class ElectraModel(nn.Module):
    """models/electra.py:310-510: BERT's backbone with a factorised embedding (embed_size may
    differ from units: ``embed_factorized_proj``) and the raw CLS row as pooled output."""

    def __init__(self, vocab_size=30000, units=768, hidden_size=3072, embed_size=128,
                 num_layers=12, num_heads=12, max_length=512, hidden_dropout_prob=0.,
                 attention_dropout_prob=0., num_token_types=2, pos_embed_type='learned',
                 activation='gelu', layer_norm_eps=1E-12, embed_initializer=None,
                 weight_initializer=None, bias_initializer='zeros', dtype='float32',
                 use_pooler=True, layout='NT', compute_layout='auto'):
        super().__init__()
        self._dtype = dtype
        self.use_pooler = use_pooler
        self.pos_embed_type = pos_embed_type
        self.num_token_types = num_token_types
        self.vocab_size = vocab_size
        self.embed_size = embed_size
        self.units = units
        self.max_length = max_length
        self.layer_norm_eps = layer_norm_eps
        self._layout = layout
        if compute_layout is None or compute_layout == 'auto':
            self._compute_layout = layout
        else:
            self._compute_layout = compute_layout
        self.encoder = ElectraEncoder(
            units=units, hidden_size=hidden_size, num_layers=num_layers, num_heads=num_heads,
            attention_dropout_prob=attention_dropout_prob,
            hidden_dropout_prob=hidden_dropout_prob, output_attention=False,
            output_all_encodings=False, activation=activation, layer_norm_eps=layer_norm_eps,
            weight_initializer=weight_initializer, bias_initializer=bias_initializer, dtype=dtype,
            layout=self._compute_layout)
        self.word_embed = Embedding(vocab_size, embed_size, weight_initializer=embed_initializer,
                                    dtype=dtype)
        self.token_type_embed = Embedding(num_token_types, embed_size,
                                          weight_initializer=weight_initializer, dtype=dtype)
        if pos_embed_type is not None:
            self.token_pos_embed = PositionalEmbedding(units=embed_size, max_length=max_length,
                                                       dtype=dtype, method=pos_embed_type)
        self.embed_layer_norm = LayerNorm(epsilon=layer_norm_eps, in_channels=embed_size,
                                          dtype=dtype)
        if embed_size != units:
            self.embed_factorized_proj = Dense(units=units, in_units=embed_size, flatten=False,
                                               weight_initializer=weight_initializer,
                                               bias_initializer=bias_initializer, dtype=dtype)

    @property
    def layout(self):
        return self._layout

    def get_initial_embedding(self, inputs, token_types=None):
        """models/electra.py:466-510."""
        pos_w = (self.token_pos_embed.table(inputs, self._layout)
                 if self.pos_embed_type is not None else None)
        return embedding_forward(inputs, token_types, self.word_embed.weight.data,
                                 self.token_type_embed.weight.data, pos_w,
                                 self.embed_layer_norm.gamma.data, self.embed_layer_norm.beta.data,
                                 self.layer_norm_eps, self._layout)

    def forward(self, inputs, token_types, valid_length=None):
        """models/electra.py:403-464."""
        prev = T.as_tensor(self.get_initial_embedding(inputs, token_types))
        if self.embed_size != self.units:
            prev = T.as_tensor(self.embed_factorized_proj(prev))
        if self._compute_layout != self._layout:
            ctx, _ = self.encoder(torch.swapaxes(prev, 0, 1), valid_length)
            ctx = torch.swapaxes(T.as_tensor(ctx), 0, 1)
        else:
            ctx, _ = self.encoder(prev, valid_length)
        if self.use_pooler:
            return ctx, cls_pooling(None, ctx, self._layout)
        return ctx
