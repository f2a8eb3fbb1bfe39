"""Developer probe: the fused attention kernel's variants at one shape: predicate, dense byte
mask, bf16 bias, causal predicate, memory-mask predicate (HIP events around 10 calls each)."""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch
from gluonnlp_amd import _lib, attention_cell as ac


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    gc.collect()          # (a full collection inside the ten timed calls reads as a 30-80 ms call)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B, L, H in ((64, 512, 16), (16, 1024, 12)):
    d = 64
    q, k, v = [torch.randn(B, L, H, d, device='cuda').to(torch.bfloat16) for _ in range(3)]
    vl = torch.full((B,), L, device='cuda')
    z = torch.zeros(B, L, 1, device='cuda')
    pred = ac.gen_self_attn_mask(z, vl)
    dense = pred.clone()
    causal = ac.gen_self_attn_mask(z, vl, attn_type='causal')
    mvl = torch.randint(L // 2, L + 1, (B,), device='cuda')
    memp = ac.gen_mem_attn_mask(z, mvl, z, None)
    memd = memp.clone()
    cdense = causal.clone()
    e = torch.randn(H, L, L, device='cuda').to(torch.bfloat16)
    kw = dict(scaled=True, query_head_units=d, layout='NTK', _need_scores=False, _need_weights=False)
    for name, m, ee in (('predicate', pred, None), ('dense mask', dense, None), ('bias', pred, e),
                        ('bias + dense mask', dense, e), ('causal predicate', causal, None),
                        ('causal dense', cdense, None), ('causal + bias', causal, e),
                        ('memory predicate', memp, None), ('memory dense', memd, None)):
        t = timed(lambda: ac.multi_head_dot_attn(q, k, v, m, ee, **kw))
        print('B=%d L=%d %-20s %.1f us (kernel %d)' % (B, L, name, t, _lib.load().gnlp_attn_last_kernel()),
              flush=True)
