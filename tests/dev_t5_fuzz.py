"""Developer script (kept under tests/: it uses the oracle; not collected by pytest): randomized
parity of the T5 encoder stack (RMSNorm pre-norm blocks, raw scores + relative-position bias)
against the float64 oracle over random widths, head sizes, lengths (multiples of 4 and not),
batch sizes, valid lengths, activations, layouts and dtypes.
usage: python tests/dev_t5_fuzz.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import ref
from tests.util import load_params, oracle_params, prep, rel_err, to_dev, to_np, tol
from gluonnlp_amd.models import T5Encoder

DT = {torch.float32: 'float32', torch.bfloat16: 'bfloat16'}
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(ncase):
    dtype = torch.bfloat16 if rng.random() < 0.65 else torch.float32
    H = int(rng.choice([1, 2, 4, 6]))
    d_kv = int(rng.choice([8, 16, 32, 64, 64]))
    d_model = int(rng.choice([32, 64, 128, 256]))
    d_ff = int(rng.choice([64, 128, 512]))
    NL = int(rng.integers(1, 3))
    act = str(rng.choice(['relu', 'gated-gelu']))
    layout = str(rng.choice(['NT', 'TN']))
    B, L = int(rng.integers(1, 5)), int(rng.integers(1, 140))
    if rng.random() < 0.25:
        B, L = [(1, 256), (2, 128), (1, 511), (2, 97)][int(rng.integers(0, 4))]
    vl = rng.integers(1, L + 1, size=B)
    lp = [ref.make_t5_block_params(rng, d_model, d_kv, d_ff, H, act != 'relu') for _ in range(NL)]
    relw = (rng.standard_normal((32, H)) * d_model ** -0.5).astype(np.float32)
    fin = {'gamma': (1 + 0.1 * rng.standard_normal(d_model)).astype(np.float32),
           'beta': np.zeros(d_model, np.float32)}
    x = rng.standard_normal((B, L, d_model))
    try:
        exp = ref.t5_encoder(prep(x, dtype), vl, prep(relw, dtype), [oracle_params(p, dtype) for p in lp],
                             oracle_params(fin, dtype), H, activation_name=act)
        enc = T5Encoder(d_model, d_kv, d_ff, num_layers=NL, num_heads=H, activation=act, dtype=DT[dtype],
                        layout=layout)
        for i in range(NL):
            load_params(enc.layers[i], lp[i], dtype)
        load_params(enc.final_layer_norm, fin, dtype)
        load_params(enc.relative_position_encoder, {'_rel_pos_embed.weight': relw}, dtype)
        enc.cuda()
        xin = to_dev(x, dtype)
        if layout == 'TN':
            xin = xin.transpose(0, 1).contiguous()
        got = to_np(enc(xin, torch.tensor(vl, device='cuda')))
        if layout == 'TN':
            got = np.swapaxes(got, 0, 1)
        valid = np.arange(L)[None, :] < vl[:, None]
        ok = bool(np.isfinite(got).all())
        err = float(rel_err(got[valid], exp[valid])) if ok else float('nan')
        # bf16: NL blocks and the final RMSNorm each hand a bf16 tensor on (a single block + norm
        # reached 2.1e-2 on 1 element of 2560 with a global figure of 6e-3: the printed breakdown)
        lim = tol(dtype) * ((NL + 1) ** 0.5 if dtype == torch.bfloat16 else 1.0)
        if not ok or err > lim:
            bad += 1
            print('case %d: %s d_model=%d d_kv=%d H=%d d_ff=%d NL=%d B=%d L=%d vl=%s act=%s layout=%s finite=%s err=%.3g'
                  % (it, DT[dtype], d_model, d_kv, H, d_ff, NL, B, L, vl.tolist(), act, layout, ok, err), flush=True)
            if ok:   # a rounding tail (a few elements, small global figure) or something systematic?
                dlt = np.abs(got[valid] - exp[valid])
                rel = dlt / (np.abs(exp[valid]) + np.sqrt(np.mean(exp[valid] ** 2)))
                print('   global max|d|/max|ref| = %.3g; elements above 1e-2: %d of %d; above 1.5e-2: %d; '
                      'rms of the element-wise figure %.3g'
                      % (dlt.max() / np.abs(exp[valid]).max(), int((rel > 1e-2).sum()), rel.size,
                         int((rel > 1.5e-2).sum()), float(np.sqrt(np.mean(rel ** 2)))), flush=True)
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print('case %d: %s d_model=%d d_kv=%d H=%d d_ff=%d NL=%d B=%d L=%d act=%s layout=%s raised %r'
              % (it, DT[dtype], d_model, d_kv, H, d_ff, NL, B, L, act, layout, ex), flush=True)
print('%d cases, %d flagged' % (ncase, bad))
