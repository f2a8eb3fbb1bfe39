"""Minimax fit (a linear program) of the transcendental-free erf-GELU of the bf16 GEMM epilogues:
     gelu(x) ~= x * sat(0.5 + u * Q(y)),   u = x / C,  y = sat(u^2),  Q of degree d,  sat = clamp to [0, 1]
(2 v_pk_mul [one with the clamp modifier], d x v_pk_fma, v_pk_fma with clamp, v_pk_mul: no exp / rcp).
Inside |x| <= C the error is weighted by what the GPU test asserts after the bf16 rounding of the
output (|err| <= 4e-3 |ref| + 3e-4, of which the rounding takes up to 2^-8 |ref|); beyond C the
polynomial sees y = 1 and Q(1) >= 0.5 is asked, so the clamp gives exactly 1 / 0 there: gelu = x
for large x, -0 for very negative x.
Run in the build container:  python tools/fit/gelu_poly_fit.py [C] [degree]"""
import sys
import numpy as np
from scipy.special import erf
from scipy.optimize import linprog


def gelu(x):
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def fit(C, d, n=4001):
    x = np.linspace(0, C, n)[1:]          # the error is odd-symmetric: x > 0 is enough, weighted by
    u = x / C                             # the tighter (negative) side's bound
    y = u * u
    w = (4e-3 - 2.0 ** -8) * np.abs(gelu(-x)) + 3e-4
    A = (x * u)[:, None] * y[:, None] ** np.arange(d + 1)[None, :]
    b = gelu(x) - 0.5 * x
    nq = d + 1
    c = np.zeros(nq + 1); c[-1] = 1.0
    G = -np.ones((1, nq + 1)); G[0, -1] = 0.0     # -Q(1) <= -0.5
    Aub = np.vstack([np.hstack([A, -w[:, None]]), np.hstack([-A, -w[:, None]]), G])
    bub = np.concatenate([b, -b, [-0.5]])
    r = linprog(c, A_ub=Aub, b_ub=bub, bounds=[(None, None)] * nq + [(0, None)], method='highs')
    return (r.x[:nq], r.x[-1]) if r.status == 0 else (None, np.inf)


def eval_f32(x, q, C):
    """the device's float32 arithmetic, fma by fma"""
    x = x.astype(np.float32)
    xs = (x * np.float32(1.0 / C)).astype(np.float32)
    y = np.clip((xs * xs).astype(np.float32), 0, 1)
    t = np.full_like(y, np.float32(q[-1]))
    for k in range(len(q) - 2, -1, -1):
        t = (t.astype(np.float64) * y + np.float64(np.float32(q[k]))).astype(np.float32)
    phi = np.clip((xs.astype(np.float64) * t + 0.5).astype(np.float32), 0, 1)
    return x * phi


def check(q, C):
    import torch
    torch.manual_seed(0)
    worst, ok = 0.0, True
    for g in [torch.linspace(-30, 30, 2048), torch.linspace(-30, 30, 1 << 16),
              torch.randn(1 << 20) * 3, torch.linspace(-6, 6, 1 << 18)]:
        v = g.to(torch.bfloat16).float().numpy()
        got = torch.from_numpy(eval_f32(v, q, C)).to(torch.bfloat16).float().numpy()
        e = gelu(v.astype(np.float64))
        worst = max(worst, float((np.abs(got - e) / (4e-3 * np.abs(e) + 3e-4)).max()))
        ok &= bool(np.all(got[v > 9] == v[v > 9]) and np.all(np.abs(got[v < -9]) < 1e-10))
    v = np.linspace(-12, 12, 1 << 18).astype(np.float32)
    e = gelu(v.astype(np.float64))
    err = np.abs(eval_f32(v, q, C) - e)
    a = float((err / ((4e-3 - 2.0 ** -8) * np.abs(e) + 3e-4)).max())
    return worst, a, float(err.max()), ok


if __name__ == '__main__':
    Cs = [float(sys.argv[1])] if len(sys.argv) > 1 else [3.8, 3.9, 4.0, 4.25, 4.5]
    ds = [int(sys.argv[2])] if len(sys.argv) > 2 else [5, 6, 7]
    for C in Cs:
        for d in ds:
            q, t = fit(C, d)
            if q is None:
                print(f'C={C} d={d}: infeasible'); continue
            worst, a, amax, ok = check(q, C)
            print(f'C={C} d={d} lp_t={t:.3f}  test-bound ratio {worst:.3f}  approximation alone: '
                  f'{a:.3f} of its share, max abs {amax:.2e}  tails exact: {ok}')
            if len(Cs) == 1:
                print('  q =', ', '.join(f'{float(np.float32(v))!r}f' for v in q))
