"""Developer tool: the one-wave-per-SIMD GEMM (GNLP_GEMM_W4=1) against the one-tile-per-workgroup
kernel, bit for bit, on the BERT shapes and a few awkward ones; then timings of both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from gluonnlp_amd.layers import Dense


def one(M, N, K, act, other):
    torch.manual_seed(M + N + K)
    d = Dense(N, K, dtype='bfloat16').cuda()
    d.weight.data.normal_(0, 0.05)
    d.bias.data.normal_(0, 0.5)
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    o = torch.randn(M, N, device='cuda').to(torch.bfloat16) if other else None
    outs = []
    for v in ('0', '1'):
        __import__('gluonnlp_amd._lib', fromlist=['x']).set_knob('GNLP_GEMM_W4', v)
        out = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
        d(x, act=act, other=o, out=out)
        torch.cuda.synchronize()
        outs.append(out)
    same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    ref = torch.nn.functional.linear(x.float(), d.weight.data.float(), d.bias.data.float())
    if act == 'gelu':
        ref = torch.nn.functional.gelu(ref)
    if act == 'relu':
        ref = torch.relu(ref)
    if other:
        ref = ref + o.float()
    err = ((outs[1].float() - ref).abs().max() / ref.abs().max()).item()
    print('M=%d N=%d K=%d %s%s: bit-identical=%s  rel err vs fp32 %.2e' %
          (M, N, K, act, '+res' if other else '', same, err), flush=True)
    return same and err < 2e-2


if __name__ == '__main__':
    ok = True
    for args in [(256, 256, 128, 'identity', False), (512, 768, 192, 'gelu', False),
                 (2048, 2304, 768, 'identity', True), (32768, 1024, 1024, 'identity', True),
                 (32768, 4096, 1024, 'gelu', False), (32768, 1024, 4096, 'identity', True),
                 (32768, 3072, 1024, 'identity', False), (16384 + 256, 512, 320, 'relu', True)]:
        ok = one(*args) and ok
    print('ALL OK' if ok else 'MISMATCH', flush=True)
    if ok and os.environ.get('BENCH', '1') == '1':
        from gemm_bench import run
        for v in ('0', '1', '0', '1'):
            __import__('gluonnlp_amd._lib', fromlist=['x']).set_knob('GNLP_GEMM_W4', v)
            line = []
            for name, m, n, k, act, other in [('qkv', 32768, 3072, 1024, 'identity', False),
                                              ('proj', 32768, 1024, 1024, 'identity', True),
                                              ('ffn1', 32768, 4096, 1024, 'gelu', False),
                                              ('ffn2', 32768, 1024, 4096, 'identity', True)]:
                ms, tf = run(m, n, k, act, other)
                line.append('%s %.0fus %.0fTF' % (name, ms * 1e3, tf))
            print('W4=%s | %s' % (v, ' | '.join(line)), flush=True)
