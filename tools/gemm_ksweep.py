"""Per-tile fixed cost vs per-K-tile cost of the big-GEMM kernels: time(M, N, K) for several K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from gemm_bench import run
M = 32768
for ps in os.environ.get('PSVALS', '1,0').split(','):
    __import__('gluonnlp_amd._lib', fromlist=['x']).set_knob('GNLP_GEMM_W4', ps)
    for N, act, other in ((4096, 'identity', False), (4096, 'gelu', False), (1024, 'identity', True)):
        ts = []
        for K in (512, 1024, 2048, 4096):
            ms, tf = run(M, N, K, act, other, iters=10)
            ts.append((K, ms * 1e3, tf))
        tiles_per_cu = (M // 256) * (N // 256) / 256.0
        # least squares over K: t = a + b * (K/64)
        import numpy as np
        A = np.array([[1.0, k / 64.0] for k, _, _ in ts]); y = np.array([t for _, t, _ in ts])
        (a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
        print('PS=%s N=%d act=%s other=%d | %s | per-tile fixed %.2f us, per K-tile %.3f us (ideal 0.85 @2.4GHz)' % (
            ps, N, act, other, ' '.join('K%d %.0fus %.0fTF' % t for t in ts), a / tiles_per_cu, b / tiles_per_cu), flush=True)
