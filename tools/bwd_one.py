"""One short run of the bf16 attention backward at the BERT-large shape (for rocprofv3 --pmc)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch  # noqa: E402
from gluonnlp_amd import attention_cell as ac  # noqa: E402

B, H, L, d = 64, 16, 512, 64
q, k, v = [torch.randn(B, L, H, d, device='cuda').to(torch.bfloat16).requires_grad_(True)
           for _ in range(3)]
vl = torch.full((B,), L, device='cuda')
mask = ac.gen_self_attn_mask(torch.zeros(B, L, 1, device='cuda'), vl)
out, _ = ac.multi_head_dot_attn(q, k, v, mask, scaled=True, query_head_units=d, layout='NTK',
                                _need_scores=False, _need_weights=False)
g = torch.randn_like(out)
for _ in range(3):
    torch.autograd.grad(out, (q, k, v), g, retain_graph=True)
torch.cuda.synchronize()
print('done')
