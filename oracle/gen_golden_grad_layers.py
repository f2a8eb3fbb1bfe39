"""Golden GRADIENTS of the composed blocks (second half: the backward pass of
the fused Dense epilogues, PositionwiseFFN and TransformerEncoderLayer) from the REFERENCE's
PyTorch sibling under torch autograd, float64, CPU.

TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).  The reference differentiates
these blocks as plain autograd-able compositions (src/gluonnlp/layers.py:619-647,
models/transformer.py:234-272); its torch sibling (torch/layers.py:188-279,
torch/models/transformer.py:11-110) states the same compositions in torch.  Each case contracts the
block's output with a fixed random cotangent and stores the gradients w.r.t. the input and every
parameter under the reference's (MX) parameter names.

Usage:  python oracle/gen_golden_grad_layers.py   ->  tests/golden/grad_layers.npz
"""
import os

import numpy as np
import torch as th

from gen_golden import OUT, load_sibling, t

FFN_REN = {'ffn_1.weight': 'ffn_1.weight', 'ffn_1.bias': 'ffn_1.bias',
           'ffn_1_gate.weight': 'gated_ffn_1.weight', 'ffn_1_gate.bias': 'gated_ffn_1.bias',
           'ffn_2.weight': 'ffn_2.weight', 'ffn_2.bias': 'ffn_2.bias',
           'layer_norm.weight': 'layer_norm.gamma', 'layer_norm.bias': 'layer_norm.beta'}


def main():
    th.set_default_dtype(th.float64)
    ac, ly, tr = load_sibling()
    out = {}
    rng = np.random.default_rng(77)
    # ---- PositionwiseFFN: post-norm gelu (BERT), pre-norm relu, gated gelu(tanh) (T5 v1.1 form) ----
    for name, kw in [('ffn_post_gelu', dict(activation='gelu', pre_norm=False)),
                     ('ffn_pre_relu', dict(activation='relu', pre_norm=True)),
                     ('ffn_gated_tanh', dict(activation='gelu(tanh)', pre_norm=False,
                                             gated_proj=True))]:
        U, F = 64, 128
        ffn = ly.PositionwiseFFN(units=U, hidden_size=F, dropout=0.0, **kw).double().eval()
        with th.no_grad():
            for prm in ffn.parameters():
                prm.copy_(t(rng.standard_normal(tuple(prm.shape)) * 0.2))
            ffn.layer_norm.weight.add_(1.0)
        x = rng.standard_normal((3, 8, U)).astype(np.float32)
        g = rng.standard_normal((3, 8, U)).astype(np.float32)
        xx = t(x).double().requires_grad_(True)
        y = ffn(xx)
        (y * t(g).double()).sum().backward()
        out[name + '_x'], out[name + '_g'], out[name + '_y'] = x, g, y.detach().numpy()
        out[name + '_dx'] = xx.grad.numpy()
        for kk, prm in ffn.named_parameters():
            out['{}/{}'.format(name, FFN_REN[kk])] = prm.detach().numpy()
            out['{}/d_{}'.format(name, FFN_REN[kk])] = prm.grad.numpy()
    # ---- TransformerEncoderLayer, post-norm (the pre-norm residual differs between the MX text and
    # the sibling: SURVEY 8c (3)); ragged valid_length incl. a length-1 sequence; 64-wide heads in
    # the second case so that the bf16 run takes the fused kernels ----
    for name, U, F, H, B, L, vl, act in [('enc_small', 48, 64, 2, 3, 8, [8, 5, 1], 'gelu'),
                                          ('enc_d64', 128, 256, 2, 2, 80, [80, 37], 'gelu')]:
        layer = tr.TransformerEncoderLayer(units=U, hidden_size=F, num_heads=H,
                                           attention_dropout_prob=0.0, hidden_dropout_prob=0.0,
                                           layer_norm_eps=1e-12, activation=act,
                                           layout='NT').double().eval()
        with th.no_grad():
            for n_, prm in layer.named_parameters():
                if prm.ndim == 2:
                    w = np.clip(rng.standard_normal(tuple(prm.shape)) * 0.08, -0.2, 0.2)
                elif n_.endswith('layer_norm.weight'):
                    w = 1.0 + 0.1 * rng.standard_normal(tuple(prm.shape))
                else:
                    w = 0.1 * rng.standard_normal(tuple(prm.shape))
                prm.copy_(t(w.astype(np.float32)).double())
        x = rng.standard_normal((B, L, U)).astype(np.float32)
        g = rng.standard_normal((B, L, U)).astype(np.float32)
        vl = np.asarray(vl, dtype=np.int64)
        g[np.arange(L)[None, :] >= vl[:, None]] = 0.0   # padded positions carry no cotangent
        mask = ac.gen_self_attn_mask(t(x), t(vl), attn_type='full', layout='NT')
        xx = t(x).double().requires_grad_(True)
        y, _ = layer(xx, mask)
        (y * t(g).double()).sum().backward()
        out[name + '_x'], out[name + '_g'], out[name + '_valid_length'] = x, g, vl
        out[name + '_heads'] = np.array(H)
        out[name + '_y'], out[name + '_dx'] = y.detach().numpy(), xx.grad.numpy()
        for kk, prm in layer.named_parameters():
            kk = kk.replace('layer_norm.weight', 'layer_norm.gamma') \
                   .replace('layer_norm.bias', 'layer_norm.beta')
            out['{}/{}'.format(name, kk)] = prm.detach().numpy()
            out['{}/d_{}'.format(name, kk)] = prm.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'grad_layers.npz'), **out)
    print('grad_layers.npz written:', len(out), 'arrays')


if __name__ == '__main__':
    main()
