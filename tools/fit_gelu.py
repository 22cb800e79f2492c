#!/usr/bin/env python3
"""Coefficients of `ahv_encoder.hip::gelu_pk` and their error, regenerated.

GELU(g) = g Phi(g) = max(g, 0) - |g| Phi(-|g|), and log2 Phi(-t) is smooth on t >= 0 (-1 at 0, ~ -t^2 / (2 ln 2) far
out), so  Phi(-t) = exp2(p(t)),  p(t) = -1 + c1 t + ... + c8 t^8:  eight fused multiply-adds, one v_exp_f32, no branch and
no second range.  The fit is weighted least squares on Chebyshev nodes of [0, 6], re-weighted towards the minimax solution;
the weight is the ABSOLUTE error of g Phi(g) the coefficient error causes (Phi(-t) ln 2 (1 + t)): where Phi(-t) is 1e-9
nobody needs log2 Phi to seven digits.  c8 < 0 and p decreases monotonically beyond the fitted range, so large |g| need
no clamp (exp2 -> 0).

Prints the coefficients and, evaluated with fp32 rounding after every operation, the largest error against the fp64 value
next to the same figure for the expression torch evaluates in fp32 (0.5 g (1 + erff(g / sqrt 2)), attention.py:81-88 via
F.gelu): 2.6e-7 against 4.5e-7 absolute on [-300, 300].
"""
import numpy as np
from scipy.special import erf, log_ndtr, ndtr

f32 = np.float32


def fit(deg=8, T=6.0, iters=60, n=4000):
    t = 0.5 * T * (1 - np.cos(np.pi * (np.arange(n) + 0.5) / n))
    y = log_ndtr(-t) / np.log(2.0) + 1.0                      # p(t) + 1 = t (c1 + c2 t + ...)
    base_w = np.exp(log_ndtr(-t)) * np.log(2.0) * (1 + t)
    w = base_w.copy()
    A = np.stack([t ** k for k in range(1, deg + 1)], 1)
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
        err = (A @ c - y) * base_w
        w = w * (1 + 4 * np.abs(err) / np.abs(err).max())
        w /= w.max()
    return c, np.abs(err).max()


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def gelu_pk(g, cs):
    g = g.astype(f32)
    t = np.abs(g)
    r = np.full_like(t, cs[-1])
    for k in range(len(cs) - 2, -1, -1):
        r = fma(r, t, cs[k])
    r = fma(r, t, f32(-1.0))
    with np.errstate(all="ignore"):
        e = np.exp2(r.astype(np.float64)).astype(f32)
    return fma(-t, e, np.maximum(g, f32(0)))


def gelu_torch_fp32(g):
    g = g.astype(f32)
    e = erf((g * f32(0.70710678118654752)).astype(np.float64)).astype(f32)
    return (f32(0.5) * g * (f32(1) + e)).astype(f32)


if __name__ == "__main__":
    c, e = fit()
    cs = [f32(x) for x in c]
    print("weighted fit error %.2e" % e)
    print("c1..c8 = " + ", ".join("%.9ef" % x for x in cs))
    g = np.concatenate([np.linspace(-12, 12, 4_000_001), np.linspace(-300, 300, 200_001)]).astype(f32)
    ex = g.astype(np.float64) * ndtr(g.astype(np.float64))
    y, yt = gelu_pk(g, cs), gelu_torch_fp32(g)
    sc = np.maximum(np.abs(ex), 1e-3)
    print("max |gelu_pk - fp64| %.3e   max |torch fp32 expression - fp64| %.3e   max |gelu_pk - torch fp32| %.3e"
          % (np.abs(y - ex).max(), np.abs(yt - ex).max(), np.abs(y - yt).max()))
    print("relative to max(|GELU|, 1e-3): gelu_pk %.3e   torch fp32 expression %.3e"
          % ((np.abs(y - ex) / sc).max(), (np.abs(yt - ex) / sc).max()))
    tt = np.linspace(6, 400, 100_000)
    p = sum(float(cs[k]) * tt ** (k + 1) for k in range(len(cs))) - 1
    print("beyond the fitted range: c8 = %.3e, p(6) = %.1f, monotonically decreasing: %s" % (cs[-1], p[0], bool(np.all(np.diff(p) < 0))))
