"""Developer probe: fused attention, causal predicate against the full predicate over L."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch
from gluonnlp_amd import attention_cell as ac


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


H, d = 16, 64
for L in (128, 256, 512, 1024, 2048, 4096):
    B = max(1, 32768 // L)
    q, k, v = [torch.randn(B, L, H, d, device='cuda').to(torch.bfloat16) for _ in range(3)]
    vl = torch.full((B,), L, device='cuda')
    z = torch.zeros(B, L, 1, device='cuda')
    kw = dict(scaled=True, query_head_units=d, layout='NTK', _need_scores=False, _need_weights=False)
    full = ac.gen_self_attn_mask(z, vl)
    causal = ac.gen_self_attn_mask(z, vl, attn_type='causal')
    tf = timed(lambda: ac.multi_head_dot_attn(q, k, v, full, None, **kw))
    tc = timed(lambda: ac.multi_head_dot_attn(q, k, v, causal, None, **kw))
    print('B=%4d L=%5d  full %.1f us  causal %.1f us  ratio %.2f (tiles: %.2f)' % (
        B, L, tf, tc, tc / tf, (L / 128 + 1) / 2 / (L / 128) if L >= 128 else 1), flush=True)
