"""Developer script (kept under tests/: it uses the oracle, which only tests may import; not
collected by pytest): error of the bf16 attention backward kernels (bf16-MFMA and fp32-MFMA paths)
against the float64 oracle on one causal / ragged case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gluonnlp_amd
from gluonnlp_amd import _lib, attention_cell as ac
from oracle import ref
import test_gpu_backward as tb

B, H, L, d = 2, 2, 200, 64
for attn_type in ('full', 'causal'):
    rng = np.random.default_rng(B * 100 + L)
    q, k, v = [rng.standard_normal((B, H, L, d)) for _ in range(3)]
    gc = rng.standard_normal((B, L, H * d))
    mask = ac.gen_self_attn_mask(torch.zeros((B, L, 1), device='cuda'),
                                 torch.tensor((200, 77), device='cuda'), attn_type=attn_type)
    dt = torch.bfloat16
    pq, pk, pv, pg = [tb.prep(x, dt) for x in (q, k, v, gc)]
    exp = ref.multi_head_dot_attn_backward(pg, pq, pk, pv, mask.cpu().numpy(), None, scaled=True,
                                           query_head_units=d)
    for knob in (None, '0'):
        _lib.set_knob('GNLP_ATTN_BWD_BF16', knob)
        a = tb.run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, mask, None, gc, True, dt, d)
        for name, x, y in zip(('dq', 'dk', 'dv'), a[1:4], exp[:3]):
            rms = np.sqrt((y ** 2).mean())
            err = np.abs(x - y)
            # what rounding the exact answer to bf16 alone would give
            yb = torch.tensor(y).to(torch.bfloat16).double().numpy()
            print(attn_type, 'bf16-mfma' if knob is None else 'fp32-mfma', name,
                  'max|d|/rms %.4f  rms|d|/rms %.5f  (output rounding alone: max %.4f rms %.5f)' % (
                      err.max() / rms, np.sqrt((err ** 2).mean()) / rms,
                      np.abs(yb - y).max() / rms, np.sqrt(((yb - y) ** 2).mean()) / rms))
    _lib.set_knob('GNLP_ATTN_BWD_BF16', None)
