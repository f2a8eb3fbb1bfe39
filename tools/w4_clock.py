"""Developer tool (run under rocprofv3 --pmc GRBM_GUI_ACTIVE by tools/w4_clock.sh): the FFN-2 shape
on the one-tile-per-workgroup kernel (GNLP_GEMM_W4=0), the one-wave-per-SIMD kernel and the vendor library's kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch
from gluonnlp_amd.layers import Dense
M, N, K = 32768, 1024, 4096
d = Dense(N, K, dtype='bfloat16').cuda()
d.bias.data.normal_(0, 0.02)
x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
o = torch.randn(M, N, device='cuda').to(torch.bfloat16)
out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for w4 in ('0', '1'):
    __import__('gluonnlp_amd._lib', fromlist=['x']).set_knob('GNLP_GEMM_W4', w4)
    for _ in range(25):
        d(x, other=o, out=out)
    torch.cuda.synchronize()
w = d.weight.data
for _ in range(25):
    torch.matmul(x, w.t(), out=out)
torch.cuda.synchronize()
