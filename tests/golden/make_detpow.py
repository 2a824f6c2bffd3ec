"""Coefficients of the machine-independent log2 / exp2 pair (oracle/ovr_oracle.c ovr_oracle_det_log2f / _exp2f and ovr_hip_device.h det_log2f / det_exp2f):
Chebyshev interpolants converted to monomials and rounded to float32, printed as C hexadecimal float literals.
   log2(1 + f) / f  on [sqrt(1/2) - 1 - 1e-3, sqrt(2) - 1 + 1e-3], degree 9 (max relative error of the float32 evaluation 1.1e-7 = 1.4 ulp)
   2^r              on [-0.5, 0.5], degree 7 (6.5e-8 = 1.1 ulp)
python tests/golden/make_detpow.py"""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as Pl


def fit(fun, lo, hi, deg):
    k = np.arange(deg + 1)
    t = np.cos(np.pi * (k + 0.5) / (deg + 1))
    c = Ch.chebfit(t, fun(0.5 * (hi - lo) * t + 0.5 * (hi + lo)), deg)
    pt = Ch.cheb2poly(c)
    a, b = 2 / (hi - lo), -(hi + lo) / (hi - lo)
    px = np.zeros(1)
    for i, ci in enumerate(pt):
        term = np.array([1.0])
        for _ in range(i):
            term = Pl.polymul(term, [b, a])
        px = Pl.polyadd(px, ci * term)
    return px


if __name__ == "__main__":
    lo, hi = np.sqrt(0.5) - 1 - 1e-3, np.sqrt(2) - 1 + 1e-3
    print("log2(1+f)/f, c0..c9:", ", ".join(float(np.float32(x)).hex() for x in fit(lambda f: np.log2(1 + f) / f, lo, hi, 9)))
    print("2^r, c0..c7:", ", ".join(float(np.float32(x)).hex() for x in fit(lambda r: 2.0 ** r, -0.5, 0.5, 7)))
