"""Developer tool: cycles per workgroup and in-kernel clock of the fused attention kernel, from the
diagnostic ATTN_STAMP build (tools/build/attnstamp/libgluonnlp_amd.so; GNLP_LIB_PATH points at it)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
import numpy as np
import torch
from gluonnlp_amd import _lib, attention_cell as ac

_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
B, H, L, d = 64, 16, 512, 64
q, k, v = [torch.randn(B, L, H, d, device='cuda').to(torch.bfloat16) for _ in range(3)]
for it in range(int(os.environ.get('ITERS', 50))):
    ac.multi_head_dot_attn(q, k, v, None, scaled=True, query_head_units=d, layout='NTK',
                           _need_scores=False, _need_weights=False)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for it in range(20):
    ac.multi_head_dot_attn(q, k, v, None, scaled=True, query_head_units=d, layout='NTK',
                           _need_scores=False, _need_weights=False)
e.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4096 * 4))()
raw.gnlp_debug_attn_read(buf, 4096 * 4)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.float64)
cyc, real, nt, start = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
clk = cyc / np.maximum(real, 1) * 0.1
print('kernel %.1f us; workgroups stamped %d; tiles per workgroup %d' % (s.elapsed_time(e) / 20 * 1e3, (nt > 0).sum(), np.median(nt)))
print('cycles per workgroup: median %.0f (p10 %.0f, p90 %.0f) = %.0f per KV tile; in-kernel clock %.3f GHz (p10 %.3f p90 %.3f)' % (
    np.median(cyc), np.percentile(cyc, 10), np.percentile(cyc, 90), np.median(cyc / np.maximum(nt, 1)),
    np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
print('workgroup duration %.2f us median; launch span (last start - first start) %.1f us' % (
    np.median(real) / 100.0, (start.max() - start.min()) / 100.0))
