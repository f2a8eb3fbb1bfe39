"""Developer tool: randomized cross-check of the fused bf16 attention kernel against the general
fp32-MFMA kernel (same inputs; the general kernel is forced by asking for the weights) over odd
mask structures, biases, length pairs and score scales.  Prints every case that disagrees."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import numpy as np
import torch
from gluonnlp_amd import _lib, attention_cell as ac

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for it in range(ncase):
    B, H = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    Lq, Lm = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    d = 64
    scale = float(rng.choice([0.25, 1.0, 3.0]))
    q = torch.tensor(rng.standard_normal((B, Lq, H, d)) * scale, device='cuda').to(torch.bfloat16)
    k = torch.tensor(rng.standard_normal((B, Lm, H, d)) * scale, device='cuda').to(torch.bfloat16)
    v = torch.tensor(rng.standard_normal((B, Lm, H, d)), device='cuda').to(torch.bfloat16)
    kind = rng.choice(['none', 'random', 'sparse', 'band', 'upper', 'cols', 'blocks', 'pred', 'causal', 'mem'])
    mask = None
    if kind == 'random':
        m = rng.random((B, Lq, Lm)) < rng.choice([0.02, 0.5, 0.98])
    elif kind == 'sparse':
        m = np.zeros((B, Lq, Lm), bool)
        for b in range(B):
            for i in range(Lq):
                m[b, i, rng.integers(0, Lm)] = True
    elif kind == 'band':
        i, j = np.arange(Lq)[:, None], np.arange(Lm)[None, :]
        w = int(rng.integers(1, 40))
        m = np.broadcast_to(np.abs(i - j) <= w, (B, Lq, Lm)).copy()
    elif kind == 'upper':
        i, j = np.arange(Lq)[:, None], np.arange(Lm)[None, :]
        m = np.broadcast_to(j >= i, (B, Lq, Lm)).copy()
    elif kind == 'cols':
        m = np.zeros((B, Lq, Lm), bool)
        m[:, :, rng.integers(0, Lm, size=max(1, Lm // 16))] = True
    elif kind == 'blocks':
        m = np.kron(rng.random((B, (Lq + 31) // 32, (Lm + 63) // 64)) < 0.5,
                    np.ones((32, 64), bool))[:, :Lq, :Lm]
    else:
        m = None
    if m is not None:
        mask = torch.tensor(np.ascontiguousarray(m), device='cuda')
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
    if rng.random() < 0.4:
        shape = [(H, Lq, Lm), (B, H, Lq, Lm), (1, 1, Lq, Lm)][int(rng.integers(0, 3))]
        e = torch.tensor(rng.standard_normal(shape) * 2, device='cuda').to(torch.bfloat16)
    scaled = bool(rng.random() < 0.6)
    kw = dict(scaled=scaled, query_head_units=d, layout='NTK')
    a, _ = ac.multi_head_dot_attn(q, k, v, mask, e, _need_scores=False, _need_weights=False, **kw)
    ka = _lib.load().gnlp_attn_last_kernel()
    b, _ = ac.multi_head_dot_attn(q, k, v, mask, e, **kw)
    kb = _lib.load().gnlp_attn_last_kernel()
    if kind == 'mem':   # the general kernel's predicate against the same mask's untagged bytes
        c, _ = ac.multi_head_dot_attn(q, k, v, mask.clone(), e, **kw)
        if not torch.equal(b, c):
            bad += 1
            print('case %d: general kernel, memory predicate != bytes (B=%d H=%d Lq=%d Lm=%d)'
                  % (it, B, H, Lq, Lm), flush=True)
    af, bf = a.float(), b.float()
    ok = bool(torch.isfinite(af).all()) and bool(torch.isfinite(bf).all())
    err = float((af - bf).abs().max() / (bf.abs().max() + 1e-6)) if ok else float('nan')
    if not ok or err > 3e-2 or ka != 1:
        bad += 1
        print('case %d: kind=%s B=%d H=%d Lq=%d Lm=%d scale=%g scaled=%s bias=%s kernels %d/%d finite=%s err=%.3g'
              % (it, kind, B, H, Lq, Lm, scale, scaled, None if e is None else tuple(e.shape), ka, kb, ok, err),
              flush=True)
print('%d cases, %d flagged' % (ncase, bad))
