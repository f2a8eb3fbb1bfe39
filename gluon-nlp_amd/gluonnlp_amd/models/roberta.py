"""RobertaEncoder -- mirrors ``gluonnlp/models/roberta.py`` (reference
src/gluonnlp/models/roberta.py:121-188)."""
from .bert import _EncoderStack

__all__ = ['RobertaEncoder']


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
