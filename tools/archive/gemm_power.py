"""Is the big GEMM power-limited?  Same launches on random-normal, constant and all-zero operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch
from gluonnlp_amd.layers import Dense

def run(M, N, K, kind, iters=30):
    d = Dense(N, K, dtype='bfloat16').cuda()
    x = torch.randn(M, K, device='cuda').bfloat16()
    if kind == 'zeros':
        x.zero_(); d.weight.data.zero_(); d.bias.data.zero_()
    elif kind == 'ones':
        x.fill_(1.0); d.weight.data.fill_(0.01); d.bias.data.zero_()
    elif kind == 'smallint':   # few mantissa bits toggling
        x = torch.randint(-2, 3, (M, K), device='cuda').bfloat16()
        d.weight.data = torch.randint(-2, 3, (N, K), device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(5):
        d(x, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        d(x, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9

for rnd in range(2):
    for kind in ('randn', 'zeros', 'ones', 'smallint'):
        print(kind, ' | '.join('%dx%dx%d %.0fus %.0fTF' % ((M, N, K) + run(M, N, K, kind))
                                for M, N, K in ((32768, 4096, 1024), (32768, 1024, 4096))), flush=True)
