"""Developer script (kept under tests/: it uses the oracle; not collected by pytest): randomized
parity of the encoder stacks against the float64 oracle -- BertTransformer (post-norm, final
sequence mask, packed ragged path when the shape allows) and TransformerEncoder (pre / post norm)
over random widths, head counts, lengths, batch sizes, valid lengths, activations, layouts, dtypes.
usage: python tests/dev_stack_fuzz.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gluon-nlp_amd'))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import ref
from tests.util import load_params, oracle_params, prep, rel_err, to_dev, to_np, tol
from gluonnlp_amd.models import TransformerEncoder
from gluonnlp_amd.models.bert import BertTransformer

DT = {torch.float32: 'float32', torch.bfloat16: 'bfloat16'}
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(ncase):
    dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float32
    H = int(rng.choice([1, 2, 4, 8]))
    dh = int(rng.choice([8, 16, 32, 64]))
    U = H * dh
    F = int(rng.choice([U, 2 * U, 4 * U, 96]))
    NL = int(rng.integers(1, 4))
    kind = rng.choice(['bert', 'enc_post', 'enc_pre'])
    layout = str(rng.choice(['NT', 'TN']))
    if rng.random() < 0.3:          # shapes the packed ragged path takes (bf16, B * L % 256 == 0)
        B, L = [(4, 64), (2, 128), (8, 32), (1, 256)][int(rng.integers(0, 4))]
    else:
        B, L = int(rng.integers(1, 6)), int(rng.integers(1, 90))
    vl = rng.integers(0 if kind == 'bert' else 1, L + 1, size=B)
    act = str(rng.choice(['gelu', 'relu'])) if kind == 'bert' else str(rng.choice(['relu', 'gelu', 'gelu(tanh)']))
    eps = float(rng.choice([1e-12, 1e-5]))
    lp = [ref.make_encoder_layer_params(rng, U, F) for _ in range(NL)]
    x = rng.standard_normal((B, L, U))
    xin = x if layout == 'NT' else np.transpose(x, (1, 0, 2))
    try:
        if kind == 'bert':
            m = BertTransformer(units=U, hidden_size=F, num_layers=NL, num_heads=H, layer_norm_eps=eps,
                                activation=act, layout=layout, dtype=DT[dtype])
            for i in range(NL):
                load_params(m.all_layers[i] if hasattr(m, 'all_layers') else m.layers[i], lp[i], dtype)
            m.cuda()
            got = to_np(m(to_dev(xin, dtype), torch.tensor(vl, device='cuda'))[0])
            exp = ref.bert_transformer(prep(xin, dtype), vl, [oracle_params(p, dtype) for p in lp], H,
                                       eps, act, layout)[0]
        else:
            pre = kind == 'enc_pre'
            fin = {'gamma': (1 + 0.1 * rng.standard_normal(U)).astype(np.float32),
                   'beta': (0.1 * rng.standard_normal(U)).astype(np.float32)}
            m = TransformerEncoder(num_layers=NL, units=U, hidden_size=F, num_heads=H, pre_norm=pre,
                                   layer_norm_eps=eps, activation=act, dtype=DT[dtype], layout=layout)
            for i in range(NL):
                load_params(m.layers[i], lp[i], dtype)
            if pre:
                load_params(m.ln_final, fin, dtype)
            m.cuda()
            got = to_np(m(to_dev(xin, dtype), torch.tensor(vl, device='cuda')))
            exp = ref.transformer_encoder(prep(xin, dtype), vl, [oracle_params(p, dtype) for p in lp], H,
                                          eps, pre, act, layout, ln_final=oracle_params(fin, dtype))
        if layout == 'TN':
            got, exp = np.transpose(got, (1, 0, 2)), np.transpose(exp, (1, 0, 2))
        valid = np.arange(L)[None, :] < vl[:, None]
        ok = np.isfinite(got).all()
        err = rel_err(got[valid], exp[valid]) if valid.any() and ok else 0.0
        zero_ok = kind != 'bert' or float(np.abs(got[~valid]).max(initial=0.0)) == 0.0
        lim = tol(dtype) * (NL ** 0.5 if dtype == torch.bfloat16 else 1.0)
        if not ok or float(err) > lim or not zero_ok:
            bad += 1
            print('case %d: %s %s U=%d H=%d F=%d NL=%d B=%d L=%d vl=%s act=%s eps=%g layout=%s finite=%s err=%.3g zero=%s'
                  % (it, kind, DT[dtype], U, H, F, NL, B, L, vl.tolist(), act, eps, layout, ok, float(err), zero_ok),
                  flush=True)
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print('case %d: %s %s U=%d H=%d F=%d NL=%d B=%d L=%d act=%s layout=%s raised %r'
              % (it, kind, DT[dtype], U, H, F, NL, B, L, act, layout, ex), flush=True)
print('%d cases, %d flagged' % (ncase, bad))
