"""ElectraEncoder -- mirrors ``gluonnlp/models/electra.py`` (reference
src/gluonnlp/models/electra.py:175-307)."""
from .bert import _EncoderStack

__all__ = ['ElectraEncoder']


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
