"""Builder-run (not collected): HIP error and the measured floor (tests/util.py
attn_backward_floor) of the attention backward on the d = 64 goldens, per output."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'gluon-nlp_amd'))
from oracle import ref
from tests.util import attn_backward_floor, prep, rel_err
from tests.test_gpu_backward import run_attn
import gluonnlp_amd
ac = gluonnlp_amd.attention_cell
g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'grad.npz'))
q, k, v, gc, mask = [g['d64_' + n] for n in ('q', 'k', 'v', 'g_ctx', 'mask')]
md = torch.tensor(mask, device='cuda')
for dtype in (torch.float32, torch.bfloat16):
    for scaled in (1, 0):
        for ename in ('none', 'head'):
            e = None if ename == 'none' else g['d64_e_' + ename]
            key = 'd64_s{}_{}'.format(scaled, ename)
            if dtype == torch.float32:
                exp = [g[key + '_' + n] for n in ('ctx', 'dq', 'dk', 'dv')] + \
                    [g[key + '_de'] if e is not None else None]
                fl = attn_backward_floor(gc, q, k, v, mask, e, bool(scaled), 'f32')
            else:
                pq, pk, pv, pg = [prep(x, dtype) for x in (q, k, v, gc)]
                pe = None if e is None else prep(e, dtype)
                exp = [ref.multi_head_dot_attn(pq, pk, pv, mask, pe, scaled=bool(scaled), query_head_units=64)[0]] + \
                    list(ref.multi_head_dot_attn_backward(pg, pq, pk, pv, mask, pe, scaled=bool(scaled), query_head_units=64))
                fl = attn_backward_floor(pg, pq, pk, pv, mask, pe, bool(scaled), 'bf16')
            for kw, nm in (({}, 'weights'), (dict(_need_scores=False, _need_weights=False), 'fused')):
                got = run_attn(ac, 'NKT', None, q, k, v, md, e, gc, bool(scaled), dtype, 64, **kw)
                print(key, str(dtype)[6:], nm, ' '.join('{}: hip {:.2e} floor {:.2e}'.format(
                    n, float(rel_err(a, b)), float(rel_err(f, b)))
                    for n, a, b, f in zip(('ctx', 'dq', 'dk', 'dv', 'de'), got, exp, fl) if b is not None))
