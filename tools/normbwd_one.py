"""One short run of the LayerNorm backward at 32768 x 1024 bf16 (for rocprofv3 --pmc)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch  # noqa: E402
from gluonnlp_amd.layers import LayerNorm  # noqa: E402

rows, C = 32768, 1024
ln = LayerNorm(epsilon=1e-12, in_channels=C, dtype='bfloat16').cuda()
ln.gamma.requires_grad_(True)
ln.beta.requires_grad_(True)
x = torch.randn(rows, C, device='cuda').to(torch.bfloat16).requires_grad_(True)
y = ln(x)
gy = torch.randn_like(y)
for _ in range(3):
    torch.autograd.grad(y, (x, ln.gamma, ln.beta), gy, retain_graph=True)
torch.cuda.synchronize()
print('done')
