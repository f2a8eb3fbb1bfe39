"""Developer tool: randomized cross-check of the bf16-MFMA attention backward (with and without
the forward's statistics) against the general fp32-MFMA backward kernels on the same inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import numpy as np
import torch
from gluonnlp_amd import _lib, attention_cell as ac

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0


def grads(q, k, v, mask, e, go, kw, fused_fwd):
    qq, kk, vv = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    ee = None if e is None else e.detach().clone().requires_grad_(True)
    extra = dict(_need_scores=False, _need_weights=False) if fused_fwd else {}
    out, _ = ac.multi_head_dot_attn(qq, kk, vv, mask, ee, **kw, **extra)
    out.backward(go)
    return [t.grad.float() for t in (qq, kk, vv)] + ([ee.grad.float()] if ee is not None else [])


for it in range(ncase):
    B, H = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    L = int(rng.integers(1, 260))
    Lq, Lm = (L, L) if rng.random() < 0.6 else (int(rng.integers(1, 200)), int(rng.integers(1, 200)))
    d = 64
    q = torch.tensor(rng.standard_normal((B, Lq, H, d)), device='cuda').to(torch.bfloat16)
    k = torch.tensor(rng.standard_normal((B, Lm, H, d)), device='cuda').to(torch.bfloat16)
    v = torch.tensor(rng.standard_normal((B, Lm, H, d)), device='cuda').to(torch.bfloat16)
    go = torch.tensor(rng.standard_normal((B, Lq, H * d)), device='cuda').to(torch.bfloat16)
    kind = rng.choice(['none', 'random', 'upper', 'blocks', 'pred', 'causal', 'mem'])
    mask = None
    if kind == 'random':
        mask = torch.tensor(rng.random((B, Lq, Lm)) < rng.choice([0.05, 0.5, 0.95]), device='cuda')
    elif kind == 'upper':
        i, j = np.arange(Lq)[:, None], np.arange(Lm)[None, :]
        mask = torch.tensor(np.broadcast_to(j >= i, (B, Lq, Lm)).copy(), device='cuda')
    elif kind == 'blocks':
        m = np.kron(rng.random((B, (Lq + 31) // 32, (Lm + 63) // 64)) < 0.5, np.ones((32, 64), bool))
        mask = torch.tensor(np.ascontiguousarray(m[:, :Lq, :Lm]), device='cuda')
    elif kind in ('pred', 'causal') and Lq == Lm:
        vl = torch.tensor(rng.integers(0, Lq + 1, size=B), device='cuda')
        mask = ac.gen_self_attn_mask(torch.zeros((B, Lq, 1), device='cuda'), vl,
                                     attn_type='full' if kind == 'pred' else 'causal')
    elif kind == 'mem':
        mask = ac.gen_mem_attn_mask(torch.zeros((B, Lm, 1), device='cuda'),
                                    torch.tensor(rng.integers(0, Lm + 1, size=B), device='cuda'),
                                    torch.zeros((B, Lq, 1), device='cuda'),
                                    torch.tensor(rng.integers(0, Lq + 1, size=B), device='cuda')
                                    if rng.random() < 0.7 else None)
    e = None
    if rng.random() < 0.3:
        shape = [(H, Lq, Lm), (B, H, Lq, Lm), (1, 1, Lq, Lm)][int(rng.integers(0, 3))]
        e = torch.tensor(rng.standard_normal(shape), device='cuda').to(torch.bfloat16)
    kw = dict(scaled=bool(rng.random() < 0.7), query_head_units=d, layout='NTK')
    _lib.set_knob('GNLP_ATTN_BWD_BF16', '0')
    # (a memory mask: the reference reads its untagged bytes, and the fp32-MFMA kernels evaluating
    # the predicate are checked against that as well)
    ref = grads(q, k, v, mask.clone() if kind == 'mem' else mask, e, go, kw, False)
    runs = [(False, grads(q, k, v, mask, e, go, kw, False))] if kind == 'mem' else []
    _lib.set_knob('GNLP_ATTN_BWD_BF16', None)
    runs += [(f, None) for f in (False, True)]
    for fused_fwd, got in runs:
        if got is None:
            got = grads(q, k, v, mask, e, go, kw, fused_fwd)
        for name, a, b in zip(('dq', 'dk', 'dv', 'de'), got, ref):
            fin = bool(torch.isfinite(a).all())
            rms = float(b.pow(2).mean().sqrt()) + 1e-4 * float(go.float().abs().max())
            err = float(((a - b).abs() / (b.abs() + rms)).max()) if fin else float('nan')
            if not fin or err > 5e-2:
                bad += 1
                print('case %d: kind=%s B=%d H=%d Lq=%d Lm=%d scaled=%s bias=%s fused_fwd=%s %s finite=%s err=%.3g'
                      % (it, kind, B, H, Lq, Lm, kw['scaled'], None if e is None else tuple(e.shape),
                         fused_fwd, name, fin, err), flush=True)
print('%d cases, %d flagged' % (ncase, bad))
