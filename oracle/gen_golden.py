"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own code.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (it needs /root/reference); the
fixtures it writes are plain data (.npz: inputs, weights, expected outputs) and are committed.
Nothing here is imported by the product, the GPU tests, smoke() or bench.py.

Source of truth: the reference's PyTorch sibling ``src/gluonnlp/torch/{attention_cell,layers,
models/transformer}.py``, imported with stub parent packages so that ``gluonnlp/__init__.py``
(which needs mxnet) never executes.  Everything runs in float64 on CPU.
For post-norm layers with ``normalized=False`` the sibling is mathematically identical to the
MXNet path; where the two differ (l2_normalize, pre_norm residual, RMSNorm,
buckets) the fixture is NOT taken from the sibling -- those cases are pinned by the reference's
test properties instead (tests/test_oracle_golden.py).

Usage:  python oracle/gen_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch as th

REF = '/root/reference/src/gluonnlp'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def load_sibling():
    for name, path in [('gluonnlp', REF), ('gluonnlp.torch', REF + '/torch'),
                       ('gluonnlp.torch.models', REF + '/torch/models')]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    ac = importlib.import_module('gluonnlp.torch.attention_cell')
    ly = importlib.import_module('gluonnlp.torch.layers')
    tr = importlib.import_module('gluonnlp.torch.models.transformer')
    return ac, ly, tr


def t(x):
    return th.from_numpy(np.ascontiguousarray(x))


def main():
    os.makedirs(OUT, exist_ok=True)
    th.set_default_dtype(th.float64)
    ac, ly, tr = load_sibling()

    # ---- masks (tests/test_attention_cell.py:202-249 shapes + docstring example vl=[8,6]) ----
    rng = np.random.default_rng(11)
    out = {}
    for tag, B, L, vl in [('t', 4, 8, rng.integers(1, 9, size=4)), ('doc', 2, 8, np.array([8, 6]))]:
        vl = vl.astype(np.int64)
        out[tag + '_vl'] = vl
        for layout in ['NT', 'TN']:
            data = th.zeros((B, L, 5) if layout == 'NT' else (L, B, 5))
            for attn_type in ['full', 'causal']:
                m = ac.gen_self_attn_mask(data, t(vl), attn_type=attn_type, layout=layout)
                out['{}_{}_{}'.format(tag, attn_type, layout)] = m.numpy()
            m = ac.gen_self_attn_mask(data, None, attn_type='causal', layout=layout)
            out['{}_causal_novl_{}'.format(tag, layout)] = m.numpy()
    np.savez_compressed(os.path.join(OUT, 'mask_self.npz'), **out)

    # ---- multi_head_dot_attn (tests/test_attention_cell.py:22-52 shapes) ----
    rng = np.random.default_rng(123)
    B, Lq, Lm, dk, dv = 5, 16, 32, 8, 6
    for H in [1, 2, 3]:
        q = rng.standard_normal((B, H, Lq, dk)).astype(np.float32)
        k = rng.standard_normal((B, H, Lm, dk)).astype(np.float32)
        v = rng.standard_normal((B, H, Lm, dv)).astype(np.float32)
        mask = rng.integers(0, 2, size=(B, Lq, Lm)).astype(np.int32)
        mask[0, 3, :] = 0          # a fully-masked query row
        mask[1, :, :] = 0          # a fully-masked sample
        e_share = rng.standard_normal((Lq, Lm)).astype(np.float32)
        e_head = rng.standard_normal((H, Lq, Lm)).astype(np.float32)
        out = dict(q=q, k=k, v=v, mask=mask, e_share=e_share, e_head=e_head)
        for scaled in [True, False]:
            for ename, e in [('none', None), ('share', e_share), ('head', e_head)]:
                ctx, (scores, w) = ac.multi_head_dot_attn(
                    t(q).double(), t(k).double(), t(v).double(), t(mask).bool(),
                    None if e is None else t(e).double(), scaled=scaled, layout='NKT',
                    use_einsum=False, training=False)
                key = 's{}_{}'.format(int(scaled), ename)
                out[key + '_ctx'] = ctx.numpy()
                out[key + '_scores'] = scores.numpy()
                out[key + '_w'] = w.numpy()
        # no mask at all
        ctx, (scores, w) = ac.multi_head_dot_attn(t(q).double(), t(k).double(), t(v).double(),
                                                  None, None, scaled=True, layout='NKT',
                                                  use_einsum=False, training=False)
        out['nomask_ctx'], out['nomask_w'] = ctx.numpy(), w.numpy()
        np.savez_compressed(os.path.join(OUT, 'attn_h{}.npz'.format(H)), **out)

    # ---- PositionwiseFFN (torch/layers.py:188-279) ----
    rng = np.random.default_rng(7)
    out = {}
    for name, kw in [('postnorm_relu', dict(activation='relu')),
                     ('postnorm_gelu', dict(activation='gelu', layer_norm_eps=1e-12)),
                     ('prenorm_gelu', dict(activation='gelu', pre_norm=True)),
                     ('gated_gelu_tanh', dict(activation='gelu(tanh)', gated_proj=True)),
                     ]:
        U, F = 48, 64
        ffn = ly.PositionwiseFFN(units=U, hidden_size=F, dropout=0.0, **kw).double().eval()
        with th.no_grad():
            for prm in ffn.parameters():
                prm.copy_(t(rng.standard_normal(tuple(prm.shape)) * 0.2))
            ffn.layer_norm.weight.add_(1.0)
        x = rng.standard_normal((4, 8, U)).astype(np.float32)
        with th.no_grad():
            y = ffn(t(x).double())
        out[name + '_x'] = x
        out[name + '_y'] = y.numpy()
        sd = ffn.state_dict()
        ren = {'ffn_1.weight': 'ffn_1.weight', 'ffn_1.bias': 'ffn_1.bias',
               'ffn_1_gate.weight': 'gated_ffn_1.weight', 'ffn_1_gate.bias': 'gated_ffn_1.bias',
               'ffn_2.weight': 'ffn_2.weight', 'ffn_2.bias': 'ffn_2.bias',
               'layer_norm.weight': 'layer_norm.gamma', 'layer_norm.bias': 'layer_norm.beta'}
        for kk, vv in sd.items():
            out[name + '/' + ren[kk]] = vv.numpy()
    np.savez_compressed(os.path.join(OUT, 'ffn.npz'), **out)

    # ---- TransformerEncoderLayer + 2-layer stacks (torch/models/transformer.py:11-111) ----
    def run_layers(cfgname, U, F, H, B, L, eps, nlayers, vl, seed, act='gelu'):
        rng = np.random.default_rng(seed)
        layers = [tr.TransformerEncoderLayer(units=U, hidden_size=F, num_heads=H,
                                             attention_dropout_prob=0.0, hidden_dropout_prob=0.0,
                                             layer_norm_eps=eps, activation=act,
                                             layout='NT').double().eval()
                  for _ in range(nlayers)]
        out = {}
        for li, layer in enumerate(layers):
            with th.no_grad():
                for n_, prm in layer.named_parameters():
                    if prm.ndim == 2:
                        w = np.clip(rng.standard_normal(tuple(prm.shape)) * 0.05, -0.1, 0.1)
                    elif n_.endswith('layer_norm.weight'):
                        w = 1.0 + 0.1 * rng.standard_normal(tuple(prm.shape))
                    else:
                        w = 0.1 * rng.standard_normal(tuple(prm.shape))
                    prm.copy_(t(w.astype(np.float32)).double())
            for kk, vv in layer.state_dict().items():
                kk = kk.replace('layer_norm.weight', 'layer_norm.gamma') \
                       .replace('layer_norm.bias', 'layer_norm.beta')
                out['L{}/{}'.format(li, kk)] = vv.numpy().astype(np.float32)
        x = rng.standard_normal((B, L, U)).astype(np.float32)
        vl = np.asarray(vl, dtype=np.int64)
        out['x'], out['valid_length'] = x, vl
        with th.no_grad():
            mask = ac.gen_self_attn_mask(t(x), t(vl), attn_type='full', layout='NT')
            h = t(x).double()
            for li, layer in enumerate(layers):
                h, w = layer(h, mask)
                out['out_L{}'.format(li)] = h.numpy()
                if li == 0:
                    out['attn_w_L0'] = w.numpy()
            # TN layout must give the transposed result (tests/torch/test_bert_torch.py)
            keep = (th.arange(L)[None, :] < t(vl)[:, None])[..., None]
            out['out_seqmask'] = th.where(keep, h, th.zeros_like(h)).numpy()   # bert.py:311-312
        np.savez_compressed(os.path.join(OUT, cfgname + '.npz'), **out)

    run_layers('enc_tiny_eps1e-12', 48, 64, 2, 4, 8, 1e-12, 2, [8, 5, 3, 7], 21)
    run_layers('enc_tiny_eps1e-5', 48, 64, 2, 4, 8, 1e-5, 2, [4, 8, 6, 3], 22)
    run_layers('enc_tiny_relu_vl1', 32, 64, 4, 3, 20, 1e-5, 2, [20, 1, 9], 23, act='relu')
    # head dim 64 (the MFMA fast path), ragged incl. vl=1 and vl=L
    run_layers('enc_d64', 128, 512, 2, 3, 128, 1e-12, 2, [128, 1, 77], 24)
    # ---- TransformerDecoderLayer x 2 (torch/models/transformer.py:114-275), post-norm.  The masks
    # are built here (causal self mask by the sibling; the memory mask by its definition
    # attention_cell.py:186-274 -- the sibling's gen_mem_attn_mask reads a shape wrongly) ----
    def run_decoder(cfgname, U, MU, F, H, B, L, LM, eps, vl, mvl, seed, act='relu'):
        rng = np.random.default_rng(seed)
        layers = [tr.TransformerDecoderLayer(units=U, mem_units=MU, hidden_size=F, num_heads=H,
                                             activation_dropout=0.0, dropout=0.0,
                                             attention_dropout=0.0, layer_norm_eps=eps,
                                             activation=act, layout='NT').double().eval()
                  for _ in range(2)]
        out = {}
        ren = {'ln_in.weight': 'ln_in.gamma', 'ln_in.bias': 'ln_in.beta',
               'ln_inter.weight': 'ln_inter.gamma', 'ln_inter.bias': 'ln_inter.beta',
               'ffn.layer_norm.weight': 'ffn.layer_norm.gamma',
               'ffn.layer_norm.bias': 'ffn.layer_norm.beta'}
        for li, layer in enumerate(layers):
            with th.no_grad():
                for n_, prm in layer.named_parameters():
                    if prm.ndim == 2:
                        w = np.clip(rng.standard_normal(tuple(prm.shape)) * 0.05, -0.1, 0.1)
                    elif n_.endswith(('ln_in.weight', 'ln_inter.weight', 'layer_norm.weight')):
                        w = 1.0 + 0.1 * rng.standard_normal(tuple(prm.shape))
                    else:
                        w = 0.1 * rng.standard_normal(tuple(prm.shape))
                    prm.copy_(t(w.astype(np.float32)).double())
            for kk, vv in layer.state_dict().items():
                out['L{}/{}'.format(li, ren.get(kk, kk))] = vv.numpy().astype(np.float32)
        x = rng.standard_normal((B, L, U)).astype(np.float32)
        mem = rng.standard_normal((B, LM, MU)).astype(np.float32)
        vl, mvl = np.asarray(vl, dtype=np.int64), np.asarray(mvl, dtype=np.int64)
        out.update(x=x, mem=mem, valid_length=vl, mem_valid_length=mvl)
        with th.no_grad():
            causal = ac.gen_self_attn_mask(t(x), t(vl), attn_type='causal', layout='NT')
            mm = ((th.arange(LM)[None, None, :] < t(mvl)[:, None, None]) &
                  (th.arange(L)[None, :, None] < t(vl)[:, None, None]))
            out['self_causal_mask'], out['mem_attn_mask'] = causal.numpy(), mm.numpy()
            h = t(x).double()
            for li, layer in enumerate(layers):
                h = layer(h, t(mem).double(), causal, mm)
                out['out_L{}'.format(li)] = h.numpy()
        np.savez_compressed(os.path.join(OUT, cfgname + '.npz'), **out)

    run_decoder('dec_tiny', 48, 40, 64, 2, 3, 7, 9, 1e-5, [7, 1, 4], [9, 3, 1], 31)
    run_decoder('dec_d64', 128, 128, 256, 2, 3, 40, 72, 1e-5, [40, 1, 23], [72, 64, 5], 32,
                act='gelu')
    print('wrote fixtures to', OUT)


if __name__ == '__main__':
    main()
