"""Developer probe: the T5-shape attention with a (H, L, L) bias, full and ragged lengths: which
kernel serves it and how long it takes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import attn_bench
from gluonnlp_amd import _lib

for name, args in [('t5 bias full', (16, 1024, 12, 64, True, False, 5)),
                   ('t5 bias ragged', (16, 1024, 12, 64, True, True, 5)),
                   ('t5 bias full again', (16, 1024, 12, 64, True, False, 5))]:
    ms, tf = attn_bench.run(*args)
    print('%-22s %.1f us  kernel %d' % (name, ms * 1e3, _lib.load().gnlp_attn_last_kernel()), flush=True)
