#!/usr/bin/env python3
"""Coefficients of the branch-free erf used by the GELU epilogue of gemm_f16x2.hip (csrc/common.h: fast_erff).
|x| <= 1: erf(x) = x * q(x^2), q of degree 6;  1 < |x| <= 4: erf(x) = sign(x) * (1 - 2^(-p(|x|))), p of degree 8;
|x| > 4 clamps to 4 (1 - erf(4) = 1.5e-8 < half an fp32 ulp of 1).  Chebyshev-node least squares, then checked in
float32 arithmetic against scipy's erf."""
import numpy as np
from scipy.special import erf, erfc
from numpy.polynomial import chebyshev as C, polynomial as P

nodes = (np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) + 1) / 2
s = nodes
q = C.Chebyshev.fit(s, erf(np.sqrt(s)) / np.sqrt(s), 6, domain=[0, 1]).convert(kind=P.Polynomial).coef
t = nodes * 3 + 1
p = C.Chebyshev.fit(t, -np.log2(erfc(t)), 8, domain=[1, 4]).convert(kind=P.Polynomial).coef
print("q:", ", ".join(f"{np.float32(c):.9e}f" for c in q))
print("p:", ", ".join(f"{np.float32(c):.9e}f" for c in p))

def fast_erff(x):
    x = x.astype(np.float32); f = np.float32
    t = np.abs(x); s2 = x * x
    r = np.full_like(x, f(q[6]))
    for c in q[5::-1]: r = r * s2 + f(c)
    r1 = x * r
    tc = np.minimum(t, f(4.0))
    r = np.full_like(x, f(p[8]))
    for c in p[7::-1]: r = r * tc + f(c)
    r2 = np.copysign(f(1.0) - np.exp2(-r).astype(np.float32), x)
    return np.where(t < f(1.0), r1, r2)

xs = np.linspace(-6, 6, 2000001)
err = np.abs(fast_erff(xs).astype(np.float64) - erf(xs.astype(np.float32).astype(np.float64)))
print("max abs err (float32 evaluation):", err.max(), "at", xs[err.argmax()])
