"""Developer script (kept under tests/: it uses the oracle, which only tests may import; not
collected by pytest): error of the bf16 attention backward against the float64 oracle on un-scaled
(peaked) scores: recomputed statistics, the forward's statistics, and the fp32-MFMA kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from gluonnlp_amd import _lib, attention_cell as ac
from oracle import ref
import test_gpu_backward as tb

B, H, L, d = 2, 2, 160, 64
for scaled in (True, False):
    rng = np.random.default_rng(3)
    q, k, v = [rng.standard_normal((B, H, L, d)) for _ in range(3)]
    gc = rng.standard_normal((B, L, H * d))
    mask = ac.gen_self_attn_mask(torch.zeros((B, L, 1), device='cuda'), torch.tensor((160, 77), device='cuda'))
    dt = torch.bfloat16
    pq, pk, pv, pg = [tb.prep(x, dt) for x in (q, k, v, gc)]
    exp = ref.multi_head_dot_attn_backward(pg, pq, pk, pv, mask.cpu().numpy(), None, scaled=scaled,
                                           query_head_units=d)
    for name, knob, kw in (('bf16-mfma, recomputed statistics', None, {}),
                           ('bf16-mfma, forward statistics', None, dict(_need_scores=False, _need_weights=False)),
                           ('fp32-mfma', '0', {})):
        _lib.set_knob('GNLP_ATTN_BWD_BF16', knob)
        a = tb.run_attn(ac, 'NTK', (0, 2, 1, 3), q, k, v, mask, None, gc, scaled, dt, d, **kw)
        _lib.set_knob('GNLP_ATTN_BWD_BF16', None)
        out = []
        for nm, x, y in zip(('dq', 'dk', 'dv'), a[1:4], exp[:3]):
            rms = np.sqrt((y ** 2).mean())
            err = np.abs(x - y)
            out.append('%s rms %.5f max-metric %.4f' % (nm, np.sqrt((err ** 2).mean()) / rms,
                                                       (err / (np.abs(y) + rms)).max()))
        print('scaled=%s %-34s %s' % (scaled, name, ' | '.join(out)), flush=True)
