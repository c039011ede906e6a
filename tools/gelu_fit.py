#!/usr/bin/env python3
"""The GELU polynomials of gemm_epilogue.h (round 4): gelu(v) = v * clamp01(1/2 + v * P(w)), w = clamp01(1 - v^2 / 4.5^2).
P = fit of (Phi(c) - 1/2) / c on [0, 4.5], error weighted by c^2 (what gelu sees), reweighted towards the minimax solution,
rewritten in powers of w; the error is evaluated with fp32 Horner (fma) as the device runs it.  Prints the coefficient tables
(low -> high) of GeluW<DEG> and the maximum absolute error of gelu over all v.  numpy only; runs anywhere."""
import numpy as np
from math import erf, sqrt

c = np.linspace(1e-6, 4.5, 60001)
u = 2 * c * c / 4.5 ** 2 - 1
phi = np.array([0.5 * (1 + erf(x / sqrt(2))) for x in c])
f = (phi - 0.5) / c
fma32 = lambda a, b, cc: (a.astype(np.float64) * b.astype(np.float64) + np.float64(cc)).astype(np.float32)
for deg in (6, 8, 9, 10):
    wt = c * c + 1e-3
    for _ in range(80):
        coef = np.polynomial.chebyshev.chebfit(u, f, deg, w=wt)
        err = (np.polynomial.chebyshev.chebval(u, coef) - f) * c * c
        wt = wt * (1 + 2 * np.abs(err) / np.abs(err).max())
    a = np.polynomial.Polynomial(np.polynomial.chebyshev.cheb2poly(coef))(np.polynomial.Polynomial([1.0, -2.0])).coef.astype(np.float32)
    v = c.astype(np.float32)
    w = np.clip(fma32((v * v).astype(np.float32), np.full_like(v, np.float32(-1 / 20.25)), 1.0), 0, 1).astype(np.float32)
    p = np.full_like(w, a[-1])
    for k in range(deg - 1, -1, -1):
        p = fma32(p, w, a[k])
    worst = 0.0
    for sgn in (1, -1):
        vv = (sgn * v).astype(np.float32)
        g = (vv * np.clip(fma32(vv, p, 0.5), 0, 1).astype(np.float32)).astype(np.float64)
        worst = max(worst, np.abs(g - sgn * c * (phi if sgn > 0 else 1 - phi)).max())
    print(f"DEG {deg}: max |gelu error| {worst:.2e};  a = {{" + ", ".join(f"{x:.9e}f" for x in a) + "}")
