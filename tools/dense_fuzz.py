"""Developer tool: randomized check of gnlp_dense (through layers.Dense) against torch's float32
matmul + the same epilogue over random shapes (multiples of the tile sizes and odd ones), dtypes,
activations and fused operands.  Prints every case that disagrees."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import numpy as np
import torch
from gluonnlp_amd.layers import Dense

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ACTS = {None: lambda t: t, 'relu': torch.relu, 'gelu': lambda t: torch.nn.functional.gelu(t),
        'gelu(tanh)': lambda t: torch.nn.functional.gelu(t, approximate='tanh'), 'tanh': torch.tanh}
bad = 0
for it in range(ncase):
    dtype = torch.bfloat16 if rng.random() < 0.7 else torch.float32
    big = rng.random() < 0.5
    if rng.random() < 0.3:      # a few rows (decoding steps): the skinny kernel's shapes and edges
        M = int(rng.integers(1, 400))
        N = 4 * int(rng.integers(1, 800))
        K = 64 * int(rng.integers(1, 50))
    elif big:
        M = 256 * int(rng.integers(1, 20))
        N = int(rng.choice([192, 256, 384, 512, 768, 1024, 2304, 3072]))
        K = 64 * int(rng.integers(1, 40))
    else:
        M, N, K = int(rng.integers(1, 700)), int(rng.integers(1, 700)), int(rng.integers(1, 700))
    act = [None, 'relu', 'gelu', 'gelu(tanh)', 'tanh'][int(rng.integers(0, 5))]
    use_bias = bool(rng.random() < 0.8)
    other = rng.choice(['none', 'add', 'mul'])
    lyr = Dense(N, K, use_bias=use_bias, activation=act, dtype='bfloat16' if dtype == torch.bfloat16 else 'float32').cuda()
    with torch.no_grad():
        lyr.weight.copy_(torch.tensor(rng.standard_normal((N, K)) * 0.05))
        if use_bias:
            lyr.bias.copy_(torch.tensor(rng.standard_normal(N) * 0.5))
    x = torch.tensor(rng.standard_normal((M, K)), device='cuda').to(dtype)
    o = None if other == 'none' else torch.tensor(rng.standard_normal((M, N)), device='cuda').to(dtype)
    with torch.no_grad():
        y = lyr(x, o, combine='mul' if other == 'mul' else 'add')
        ref = x.float() @ lyr.weight.float().T
        if use_bias:
            ref = ref + lyr.bias.float()
        ref = ACTS[act](ref)
        if other == 'add':
            ref = ref + o.float()
        elif other == 'mul':
            ref = ref * o.float()
    fin = bool(torch.isfinite(y.float()).all())
    rms = float(ref.pow(2).mean().sqrt()) + 1e-9
    err = float(((y.float() - ref).abs() / (ref.abs() + rms)).max()) if fin else float('nan')
    lim = 2e-2 if dtype == torch.bfloat16 else 2e-4
    if not fin or err > lim:
        bad += 1
        print('case %d: %s M=%d N=%d K=%d act=%s bias=%s other=%s finite=%s err=%.3g'
              % (it, dtype, M, N, K, act, use_bias, other, fin, err), flush=True)
print('%d cases, %d flagged' % (ncase, bad))
