"""Developer tool: read the cycle stamps of the diagnostic W4_STAMP build of the one-wave-per-SIMD
GEMM (tools/build/stamp/libgluonnlp_amd.so, built with -DW4_STAMP; GNLP_LIB_PATH points at it).
Per steady-state K-tile and wave: cycles from the K-tile's start to barrier 1, inside barrier 1,
to barrier 2, inside barrier 2, to the vmcnt wait, inside it, inside barrier 3, to the end."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import numpy as np
import torch
from gluonnlp_amd import _lib
from gluonnlp_amd.layers import Dense

lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
NAMES = ['start->b1', 'in b1', 'b1->b2', 'in b2', 'b2->vmwait', 'in vmwait', 'in b3', 'b3->end']
for (M, N, K, act, other) in [(32768, 1024, 4096, 'identity', True), (32768, 4096, 1024, 'gelu', False),
                              (32768, 3072, 1024, 'identity', False)]:
    d = Dense(N, K, dtype='bfloat16').cuda()
    d.bias.data.normal_(0, 0.02)
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    o = torch.randn(M, N, device='cuda').to(torch.bfloat16) if other else None
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(int(os.environ.get('ITERS', 200))):     # long enough for the clock to settle
        d(x, act=act, other=o, out=out)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint * (256 * 4 * 16))()
    rc = raw.gnlp_debug_w4_read(buf, 256 * 4 * 16)
    a = np.frombuffer(buf, dtype=np.uint32).reshape(256, 4, 16).astype(np.float64)
    n, n7 = a[..., 8], a[..., 9]
    per = a[..., :8] / np.maximum(np.concatenate([np.repeat(n[..., None], 7, -1), n7[..., None]], -1), 1)
    print('M=%d N=%d K=%d %s other=%s rc=%d  K-tiles stamped per wave: %d' % (M, N, K, act, other, rc, int(np.median(n))))
    med = np.median(per.reshape(-1, 8), 0)
    p10 = np.percentile(per.reshape(-1, 8), 10, 0)
    p90 = np.percentile(per.reshape(-1, 8), 90, 0)
    for k in range(8):
        print('  %-12s median %7.0f   p10 %7.0f   p90 %7.0f cycles' % (NAMES[k], med[k], p10[k], p90[k]))
    print('  sum of medians %.0f cycles per K-tile (128 MFMAs = 2048 at 16 each)' % med.sum())
    clk = a[..., 10] / np.maximum(a[..., 11], 1) * 0.1
    print('  in-kernel clock (d memtime / d memrealtime x 100 MHz): median %.3f GHz, p10 %.3f, p90 %.3f; '
          'kernel %.1f us' % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90),
                              np.median(a[..., 11]) / 100.0))
    tiles = (M // 256) * (N // 256) / 256.0      # tiles per workgroup (256 workgroups)
    tot = np.median(a[..., 10])
    rest = tot - np.median(n) * med.sum()
    print('  whole kernel %.0f cycles per wave; %d steady K-tiles x %.0f = %.0f; the rest, %.0f cycles = '
          '%.0f per tile (%.1f tiles per workgroup): first K-tile (~%.0f) + drain + fill' %
          (tot, int(np.median(n)), med.sum(), np.median(n) * med.sum(), rest, rest / tiles, tiles, med.sum()))
    bw = per.mean(0)   # per wave index
    print('  per wave (mean over workgroups):')
    for w in range(4):
        print('    wave %d: ' % w + ' '.join('%6.0f' % v for v in bw[w]))
