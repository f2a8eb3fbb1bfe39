"""Developer tool: the vendor library's GEMM against this library's on the SAME tensors,
alternating in rounds of 100 launches (the chip is power-limited on these shapes: short bursts and
whichever runs first read high)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch
from gluonnlp_amd.layers import Dense

def timed(f, iters):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

M = 32768
for name, N, K in [('ffn2-shape', 1024, 4096), ('ffn1-shape', 4096, 1024), ('qkv-shape', 3072, 1024)]:
    d = Dense(N, K, dtype='bfloat16').cuda()
    d.weight.data.normal_(0, 0.02)
    d.bias.data.zero_()
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    w = d.weight.data
    fv = lambda: torch.matmul(x, w.t(), out=out)
    fo = lambda: d(x, out=out)
    for f in (fv, fo):
        for _ in range(5): f()
    line = []
    for rnd in range(3):
        tv = timed(fv, 100); to = timed(fo, 100)
        line.append('vendor %.0f us / ours %.0f us' % (tv, to))
    fl = 2.0 * M * N * K / 1e6
    print('%s M=%d N=%d K=%d (identity, bias, no residual): %s | last: vendor %.0f TF, ours %.0f TF' %
          (name, M, N, K, ' ; '.join(line), fl / tv, fl / to), flush=True)
