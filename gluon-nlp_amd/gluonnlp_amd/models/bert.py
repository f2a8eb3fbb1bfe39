"""BertTransformer -- mirrors the encoder stack of ``gluonnlp/models/bert.py`` (reference
src/gluonnlp/models/bert.py:213-315).  Embeddings / pooler / MLM heads are "next" (SURVEY §8f)."""
from torch import nn

from .transformer import TransformerEncoderLayer, run_encoder_stack

__all__ = ['BertTransformer']


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
