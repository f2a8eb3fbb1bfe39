"""gluonnlp_amd -- MI355X-native (gfx950) Transformer-encoder forward path behind GluonNLP's own
Python call signatures (attention_cell / layers / op / models.*).  Every computation is a
hand-written HIP kernel reached through the C ABI in include/gluonnlp_amd.h; there is no CPU
fallback (importing works without a GPU, calling does not)."""
from . import _lib  # noqa: F401
from . import op, attention_cell, layers, models, utils  # noqa: F401

__version__ = '0.1.0'
