"""Golden GRADIENTS (the backward pass of the attention and LayerNorm kernels)
from the REFERENCE's PyTorch sibling under torch autograd, float64, CPU.

TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).  What the reference's own test
checks (tests/test_attention_cell.py:85-151): ``out.backward()`` through
``multi_head_dot_attn`` and the gradients of query / key / value / rel_scores, across layouts.
Here ``out`` is contracted with a fixed random cotangent ``g_ctx`` (stronger than ones) and the
gradients are stored in the NKT layout; shapes: the reference test's (d_k = 8, d_v = 6, random 0/1
mask with a fully masked row, shared / per-head / no edge scores, scaled and un-scaled) and a
64-wide-head case with a valid_length mask.  LayerNorm: ``get_norm_layer('layer_norm')`` of the
sibling (torch/layers.py:148-177) under autograd.

Usage:  python oracle/gen_golden_grad.py   ->  tests/golden/grad.npz
"""
import os

import numpy as np
import torch as th

from gen_golden import OUT, load_sibling, t


def attn_case(ac, out, tag, rng, B, H, Lq, Lm, dk, dv, mask, biases):
    q = rng.standard_normal((B, H, Lq, dk)).astype(np.float32)
    k = rng.standard_normal((B, H, Lm, dk)).astype(np.float32)
    v = rng.standard_normal((B, H, Lm, dv)).astype(np.float32)
    g = rng.standard_normal((B, Lq, H * dv)).astype(np.float32)
    out.update({tag + '_q': q, tag + '_k': k, tag + '_v': v, tag + '_g_ctx': g,
                tag + '_mask': mask})
    for ename, e in biases.items():
        if e is not None:
            out['{}_e_{}'.format(tag, ename)] = e
        for scaled in (True, False):
            qq, kk, vv = [t(x).double().requires_grad_(True) for x in (q, k, v)]
            ee = None if e is None else t(e).double().requires_grad_(True)
            ctx, _ = ac.multi_head_dot_attn(qq, kk, vv, t(mask).bool(), ee, scaled=scaled,
                                            layout='NKT', use_einsum=False, training=False)
            (ctx * t(g).double()).sum().backward()
            key = '{}_s{}_{}'.format(tag, int(scaled), ename)
            out[key + '_ctx'] = ctx.detach().numpy()
            out[key + '_dq'], out[key + '_dk'], out[key + '_dv'] = \
                qq.grad.numpy(), kk.grad.numpy(), vv.grad.numpy()
            if ee is not None:
                out[key + '_de'] = ee.grad.numpy()


def main():
    th.set_default_dtype(th.float64)
    ac, ly, _ = load_sibling()
    out = {}
    rng = np.random.default_rng(2024)
    # reference test shapes (tests/test_attention_cell.py:22-52)
    B, H, Lq, Lm, dk, dv = 3, 2, 16, 32, 8, 6
    mask = rng.integers(0, 2, size=(B, Lq, Lm)).astype(np.int32)
    mask[0, 3, :] = 0
    attn_case(ac, out, 'small', rng, B, H, Lq, Lm, dk, dv, mask,
              {'none': None, 'share': rng.standard_normal((Lq, Lm)).astype(np.float32),
               'head': rng.standard_normal((H, Lq, Lm)).astype(np.float32)})
    # 64-wide heads, self-attention, valid_length mask (one length-1 sequence), several key tiles
    B, H, L, d = 2, 2, 72, 64
    vl = np.array([72, 37])
    mask = ((np.arange(L)[None, :, None] < vl[:, None, None]) &
            (np.arange(L)[None, None, :] < vl[:, None, None])).astype(np.int32)
    out['d64_vl'] = vl
    attn_case(ac, out, 'd64', rng, B, H, L, L, d, d, mask,
              {'none': None, 'head': rng.standard_normal((H, L, L)).astype(np.float32)})
    # LayerNorm under autograd
    for tag, rows, C, eps in [('ln48', 24, 48, 1e-12), ('ln1024', 6, 1024, 1e-5)]:
        ln = ly.get_norm_layer('layer_norm', epsilon=eps, in_channels=C).double()
        x = rng.standard_normal((rows, C)).astype(np.float32) * 2.0 + 0.3
        gam = (1.0 + 0.2 * rng.standard_normal(C)).astype(np.float32)
        bet = (0.1 * rng.standard_normal(C)).astype(np.float32)
        g = rng.standard_normal((rows, C)).astype(np.float32)
        with th.no_grad():
            ln.weight.copy_(t(gam).double())
            ln.bias.copy_(t(bet).double())
        xx = t(x).double().requires_grad_(True)
        y = ln(xx)
        (y * t(g).double()).sum().backward()
        out.update({tag + '_x': x, tag + '_gamma': gam, tag + '_beta': bet, tag + '_g': g,
                    tag + '_eps': eps, tag + '_y': y.detach().numpy(), tag + '_dx': xx.grad.numpy(),
                    tag + '_dgamma': ln.weight.grad.numpy(), tag + '_dbeta': ln.bias.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, 'grad.npz'), **out)
    print('grad.npz written:', len(out), 'arrays')


if __name__ == '__main__':
    main()
