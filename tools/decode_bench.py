"""Developer probe: time per incremental-decode step of a T5-base-sized decoder stack (12 blocks,
d_model 768, 12 heads) against a 1024-token memory, with the memory's key / value projections kept
with the decoding state (layers.memory_kv) and recomputed per step as the reference does."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch
from gluonnlp_amd._tensor import as_tensor
from gluonnlp_amd.models import T5Decoder
import gluonnlp_amd.models.t5 as m5

B, LM, STEPS = 16, 1024, 32
m = T5Decoder(d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_heads=12, dtype='bfloat16',
              layout='NT').cuda()
for p in m.parameters():
    p.data.normal_(0, 0.02)
mem = torch.randn(B, LM, 768, device='cuda').to(torch.bfloat16)
mvl = torch.randint(LM // 2, LM + 1, (B,), device='cuda')
x = torch.randn(B, 1, 768, device='cuda').to(torch.bfloat16)
kept = m5.memory_kv


@torch.no_grad()
def run():
    past = m._init_key_values(B, 'cuda', 'bfloat16')
    for i in range(STEPS):
        pos = torch.full((B,), i, dtype=torch.int32, device='cuda')
        out, past = m.incremental_decode(x, pos, past, mem, mvl)
    return out


for name, fn in (('kept', kept), ('per step (reference)', lambda o, mm, k, v: (as_tensor(k(mm)), as_tensor(v(mm)), None))):
    m5.memory_kv = fn
    run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    run()
    torch.cuda.synchronize()
    print('%-22s %.2f ms per step' % (name, (time.perf_counter() - t) / STEPS * 1e3), flush=True)
