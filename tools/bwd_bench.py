"""Developer tool: time the backward kernels at the BERT-large shapes (B=64, H=16, L=512, d=64;
LayerNorm 32768 x 1024), bf16 and fp32, HIP events around the autograd call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch
from gluonnlp_amd import attention_cell as ac
from gluonnlp_amd.layers import LayerNorm


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for dtype in (torch.bfloat16, torch.float32):
    B, H, L, d = (64, 16, 512, 64) if dtype == torch.bfloat16 else (16, 16, 512, 64)
    q, k, v = [torch.randn(B, L, H, d, device='cuda').to(dtype).requires_grad_(True) for _ in range(3)]
    vl = torch.full((B,), L, device='cuda')
    mask = ac.gen_self_attn_mask(torch.zeros(B, L, 1, device='cuda'), vl)
    out, _ = ac.multi_head_dot_attn(q, k, v, mask, scaled=True, query_head_units=d, layout='NTK',
                                    _need_scores=False, _need_weights=False)
    g = torch.randn_like(out)
    fwd = timed(lambda: ac.multi_head_dot_attn(q.detach(), k.detach(), v.detach(), mask, scaled=True,
                                               query_head_units=d, layout='NTK', _need_scores=False,
                                               _need_weights=False))
    fwd_t = timed(lambda: ac.multi_head_dot_attn(q, k, v, mask, scaled=True, query_head_units=d,
                                                 layout='NTK', _need_scores=False,
                                                 _need_weights=False))
    bwd = timed(lambda: torch.autograd.grad(out, (q, k, v), g, retain_graph=True))
    fl = 4.0 * B * H * L * L * d
    print('attention %s B=%d: fwd %.1f us (%.0f TF; %.1f us when a backward will follow), bwd %.1f us'
          % (dtype, B, fwd, fl / fwd / 1e6, fwd_t, bwd))
    rows, C = 32768, 1024
    ln = LayerNorm(epsilon=1e-12, in_channels=C, dtype='bfloat16' if dtype == torch.bfloat16 else 'float32').cuda()
    ln.gamma.requires_grad_(True); ln.beta.requires_grad_(True)
    x = torch.randn(rows, C, device='cuda').to(dtype).requires_grad_(True)
    y = ln(x)
    gy = torch.randn_like(y)
    t = timed(lambda: torch.autograd.grad(y, (x, ln.gamma, ln.beta), gy, retain_graph=True))
    e = 2 if dtype == torch.bfloat16 else 4
    print('layer_norm backward %s: %.1f us, %.2f TB/s (reads g, x; writes dx)' % (
        dtype, t, 3.0 * rows * C * e / t / 1e6))
