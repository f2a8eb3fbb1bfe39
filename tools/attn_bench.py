"""Micro-benchmark of the fused attention kernel at BERT-large / BERT-base / T5 shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import torch
from gluonnlp_amd.attention_cell import multi_head_dot_attn

def run(B, L, H, d=64, bias=False, ragged=False, iters=20):
    U = H * d
    qkv = torch.randn(B, L, 3 * U, device='cuda').to(torch.bfloat16)
    q, k, v = [qkv[..., i * U:(i + 1) * U].reshape(B, L, H, d) for i in range(3)]
    vl = torch.randint(1, L + 1, (B,), device='cuda') if ragged else torch.full((B,), L, device='cuda')
    e = torch.randn(H, L, L, device='cuda').to(torch.bfloat16) if bias else None
    f = lambda: multi_head_dot_attn(q, k, v, None, e, query_head_units=d, layout='NTK', scaled=not bias,
                                    _need_scores=False, _need_weights=False, _valid_length=vl)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    t.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(t) / iters
    return ms, 4.0 * B * H * L * L * d / ms / 1e9

if __name__ == '__main__':
  for name, args in [('bert_large B64 L512', (64, 512, 16)), ('bert_base B256 L128', (256, 128, 12)),
                     ('t5 B16 L1024 bias', (16, 1024, 12, 64, True)), ('t5 B16 L1024 no bias', (16, 1024, 12, 64, False)),
                     ('t5 bias ragged', (16, 1024, 12, 64, True, True)), ('t5 no bias ragged', (16, 1024, 12, 64, False, True)),
                     ('bert_large ragged', (64, 512, 16, 64, False, True))]:
    ms, tf = run(*args)
    print('%-22s %.1f us  %.0f TF (dense-equivalent)' % (name, ms * 1e3, tf), flush=True)
