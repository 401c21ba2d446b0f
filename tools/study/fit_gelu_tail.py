"""The coefficients of rr_common.h gelu_erf_fast: gelu(x) = max(x, 0) - z 2^p(z), z = min(|x|, 5.7), p(z) ~ log2 Phi(-z).  A polynomial
of the requested degree fitted to minimise the maximum ABSOLUTE error of z 2^p(z) on [0, 5.7] (least squares with Lawson reweighting),
then checked as the device evaluates it: fp32 coefficients, fp32 Horner with fused multiply-adds, over [-12, 12].

    python tools/study/fit_gelu_tail.py [degree]       # degree 5 -> the coefficients in the tree (round 5); 7 ~ rounds 2-4
"""
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf, erfc

ZMAX = 5.7
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
z = np.linspace(0, ZMAX, 40001)
phi = 0.5 * erfc(z / np.sqrt(2))
target = z * phi


def horner32(c, zf):
    p = np.full_like(zf, np.float32(c[0]))
    for k in c[1:]:
        p = (p.astype(np.float64) * zf.astype(np.float64) + np.float64(np.float32(k))).astype(np.float32)      # one rounding per step: an fma
    return p


w = np.ones_like(z)
c = np.polyfit(z, np.log2(phi), deg, w=np.sqrt(np.maximum(target, 1e-12)))
best = None
for _ in range(30):
    c = least_squares(lambda c: w * (z * np.exp2(np.polyval(c, z)) - target), c, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=5000).x
    e = np.abs(z * np.exp2(np.polyval(c, z)) - target)
    if best is None or e.max() < best[1]:
        best = (c.copy(), e.max())
    w = w * (1 + 4 * e / e.max())
    w /= w.mean()
c32 = [float(np.float32(k)) for k in best[0]]
x = np.linspace(-12, 12, 480001)
xf = x.astype(np.float32)
zz = np.minimum(np.abs(xf), np.float32(ZMAX))
tail = (zz.astype(np.float64) * np.exp2(horner32(c32, zz).astype(np.float64))).astype(np.float32)
gel = np.maximum(xf, 0).astype(np.float64) - tail.astype(np.float64)
exact = 0.5 * x * (1 + erf(x / np.sqrt(2)))
err = np.abs(gel - exact)
big = np.abs(exact) >= 1e-3
print(f"degree {deg}: max |gelu - exact| {err.max():.3e}; max relative error where |gelu| >= 1e-3: {(err[big] / np.abs(exact[big])).max():.3e}")
print("coefficients, highest power first:", " ".join(f"{k:.9e}f" for k in c32))
