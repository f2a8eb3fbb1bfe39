"""Encoder stacks that call the hot path -- mirrors the encoder classes of ``gluonnlp.models``."""
from . import transformer, bert, roberta, electra, t5  # noqa: F401
from .transformer import (TransformerEncoderLayer, TransformerEncoder,  # noqa: F401
                          TransformerDecoderLayer, TransformerDecoder)
from .bert import BertTransformer, BertModel  # noqa: F401
from .roberta import RobertaEncoder, RobertaModel  # noqa: F401
from .electra import ElectraEncoder, ElectraModel  # noqa: F401
from .t5 import T5Block, T5Encoder, T5Decoder, T5Model  # noqa: F401
