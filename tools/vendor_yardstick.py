"""Yardstick only (developer tool, not on the product path): what the vendor libraries reach on
the same GEMM / attention shapes, next to this library's kernels.  torch's F.linear goes to
hipBLASLt / rocBLAS; F.scaled_dot_product_attention to whatever flash kernel torch ships."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import torch.nn.functional as F
from gemm_bench import run as ours_gemm
from attn_bench import run as ours_attn


def timed(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


if __name__ == '__main__':
    M = int(os.environ.get('M', 32768))
    for name, n, k, act, other in [('qkv', 3072, 1024, 'identity', False), ('proj', 1024, 1024, 'identity', True),
                                   ('ffn1', 4096, 1024, 'gelu', False), ('ffn2', 1024, 4096, 'identity', True)]:
        x = torch.randn(M, k, device='cuda').to(torch.bfloat16)
        w = (torch.randn(n, k, device='cuda') * 0.02).to(torch.bfloat16)
        b = torch.zeros(n, device='cuda', dtype=torch.bfloat16)
        out = torch.empty(M, n, device='cuda', dtype=torch.bfloat16)
        ms_nb = timed(lambda: torch.matmul(x, w.t(), out=out))
        ms_b = timed(lambda: F.linear(x, w, b))
        ms, tf = ours_gemm(M, n, k, act, other)
        fl = 2.0 * M * n * k / 1e9
        print('%-5s M=%d N=%d K=%d | vendor matmul %.0f us %.0f TF | vendor linear+bias %.0f us %.0f TF | ours (%s%s) %.0f us %.0f TF'
              % (name, M, n, k, ms_nb * 1e3, fl / ms_nb, ms_b * 1e3, fl / ms_b, act, '+res' if other else '', ms * 1e3, tf),
              flush=True)
    B, L, H, d = 64, 512, 16, 64
    q, k_, v = [torch.randn(B, H, L, d, device='cuda').to(torch.bfloat16) for _ in range(3)]
    ms = timed(lambda: F.scaled_dot_product_attention(q, k_, v))
    mo, tfo = ours_attn(B, L, H)
    print('attn B=%d L=%d H=%d | vendor sdpa %.0f us %.0f TF | ours %.0f us %.0f TF'
          % (B, L, H, ms * 1e3, 4.0 * B * H * L * L * d / ms / 1e9, mo * 1e3, tfo), flush=True)
