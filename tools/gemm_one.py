"""Run one dense shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from gemm_bench import run
M, N, K = [int(v) for v in os.environ.get('SHAPE', '32768,1024,4096').split(',')]
ms, tf = run(M, N, K, os.environ.get('ACT', 'identity'), False, iters=int(os.environ.get('ITERS', 5)))
print('%.1f us %.0f TF' % (ms * 1e3, tf))
