"""RobertaEncoder and RobertaModel -- mirror ``gluonnlp/models/roberta.py`` (reference
src/gluonnlp/models/roberta.py:121-188 the stack, :192-398 the backbone)."""
import torch
from torch import nn

from .. import _tensor as T
from ..layers import Dense, Embedding, LayerNorm, PositionalEmbedding, embedding_forward
from .bert import _EncoderStack, cls_pooling

__all__ = ['RobertaEncoder', 'RobertaModel']


class RobertaEncoder(_EncoderStack):
    """No final sequence_mask (roberta.py:184); with output_all_encodings the input is the first
    entry of the list (roberta.py:175)."""
    _final_seq_mask = False

    def __init__(self, units=768, hidden_size=3072, num_layers=12, num_heads=12,
                 attention_dropout_prob=0.1, hidden_dropout_prob=0.1, layer_norm_eps=1E-5,
                 weight_initializer=None, bias_initializer='zeros', activation='gelu',
                 dtype='float32', output_all_encodings=False, output_attention=False,
                 layout='NT'):
        super().__init__()
        self.units = units
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.attention_dropout_prob = attention_dropout_prob
        self.hidden_dropout_prob = hidden_dropout_prob
        self.layer_norm_eps = layer_norm_eps
        self.activation = activation
        self._dtype = dtype
        self._layout = layout
        self._output_all_encodings = output_all_encodings
        self._output_attention = output_attention
        self._build(units, hidden_size, num_layers, num_heads, attention_dropout_prob,
                    hidden_dropout_prob, layer_norm_eps, weight_initializer, activation, layout,
                    dtype)

    def forward(self, x, valid_length):
        """models/roberta.py:171-188."""
        res, extra = self._run(x, valid_length)
        if self._output_all_encodings:
            res = [x] + list(res)
        return res, extra


class RobertaModel(nn.Module):
    """models/roberta.py:192-398.  Parameter paths: ``word_embed.weight``,
    ``pos_embed._embed.weight``, ``embed_ln.{gamma,beta}``, ``encoder.all_layers.*``,
    ``pooler.{weight,bias}`` (only with classifier_activation)."""

    def __init__(self, vocab_size=50265, units=768, hidden_size=3072, num_layers=12,
                 num_heads=12, max_length=512, hidden_dropout_prob=0.1,
                 attention_dropout_prob=0.1, pos_embed_type='learned', activation='gelu',
                 pooler_activation='tanh', layer_norm_eps=1E-5, embed_initializer=None,
                 weight_initializer=None, bias_initializer='zeros', dtype='float32',
                 use_pooler=True, classifier_activation=False, encoder_normalize_before=True,
                 output_all_encodings=False, layout='NT', compute_layout='auto'):
        super().__init__()
        self._dtype = dtype
        self._output_all_encodings = output_all_encodings
        self.vocab_size = vocab_size
        self.units = units
        self.max_length = max_length
        self.pos_embed_type = pos_embed_type
        self.layer_norm_eps = layer_norm_eps
        self.use_pooler = use_pooler
        self.classifier_activation = classifier_activation
        self.encoder_normalize_before = encoder_normalize_before
        self._layout = layout
        if compute_layout == 'auto' or compute_layout is None:
            self._compute_layout = layout
        else:
            self._compute_layout = compute_layout
        self.word_embed = Embedding(vocab_size, units, weight_initializer=embed_initializer,
                                    dtype=dtype)
        if encoder_normalize_before:
            self.embed_ln = LayerNorm(epsilon=layer_norm_eps, in_channels=units, dtype=dtype)
        if pos_embed_type:
            self.pos_embed = PositionalEmbedding(units=units, max_length=max_length, dtype=dtype,
                                                 method=pos_embed_type)
        self.encoder = RobertaEncoder(
            units=units, hidden_size=hidden_size, num_layers=num_layers, num_heads=num_heads,
            attention_dropout_prob=attention_dropout_prob,
            hidden_dropout_prob=hidden_dropout_prob, layer_norm_eps=layer_norm_eps,
            weight_initializer=weight_initializer, bias_initializer=bias_initializer,
            activation=activation, dtype=dtype, output_all_encodings=output_all_encodings,
            layout=self._compute_layout)
        if use_pooler and classifier_activation:
            self.pooler = Dense(units=units, in_units=units, flatten=False,
                                activation=pooler_activation,
                                weight_initializer=weight_initializer,
                                bias_initializer=bias_initializer, dtype=dtype)

    @property
    def layout(self):
        return self._layout

    def get_initial_embedding(self, inputs):
        """models/roberta.py:337-372: no token types; LayerNorm only with
        encoder_normalize_before."""
        pos_w = self.pos_embed.table(inputs, self._layout) if self.pos_embed_type else None
        g = self.embed_ln.gamma.data if self.encoder_normalize_before else None
        b = self.embed_ln.beta.data if self.encoder_normalize_before else None
        return embedding_forward(inputs, None, self.word_embed.weight.data, None, pos_w, g, b,
                                 self.layer_norm_eps, self._layout)

    def apply_pooling(self, sequence):
        """models/roberta.py:374-398."""
        return cls_pooling(self.pooler if self.classifier_activation else None, sequence,
                           self._layout)

    def forward(self, tokens, valid_length):
        """models/roberta.py:319-335."""
        emb = T.as_tensor(self.get_initial_embedding(tokens))
        if self._layout != self._compute_layout:
            ctx, _ = self.encoder(torch.swapaxes(emb, 0, 1), valid_length)
            if isinstance(ctx, list):
                ctx = [torch.swapaxes(T.as_tensor(c), 0, 1) for c in ctx]
            else:
                ctx = torch.swapaxes(T.as_tensor(ctx), 0, 1)
        else:
            ctx, _ = self.encoder(emb, valid_length)
        if self.use_pooler:
            last = ctx[-1] if isinstance(ctx, list) else ctx
            return ctx, self.apply_pooling(last)
        return ctx
