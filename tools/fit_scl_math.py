"""Coefficients of the float32 "specification arithmetic" of the Polar SC/SCL decoders (oracle/polar_scl.c and
sionna_amd/csrc/scl_math.h hold the same numbers): T(a) = log(1 + exp(-a)) for a >= 0 as

    t = a * (-log2 e);  r = rint(t);  f = t - r;  e = ldexp(1 + f * E(f), r);  T = e * Q(e)

with E of degree 5 (2^f on [-1/2, 1/2]) and Q of degree 8 (log1p(z) / z on [0, 1]), both Horner / fma.
Near-minimax fits (interpolation at Chebyshev nodes) rounded to float32; prints the hex constants and the
measured error of the float32 evaluation against float64."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P


def fit(fn, lo, hi, deg):
    c = Ch.Chebyshev.interpolate(fn, deg, domain=[lo, hi])
    return c.convert(kind=P.Polynomial, domain=[-1, 1], window=[-1, 1]).coef.astype(np.float32)


def coefficients():
    e = fit(lambda f: np.where(np.abs(f) < 1e-12, np.log(2.0), np.expm1(f * np.log(2.0)) / np.where(f == 0, 1, f)), -0.5, 0.5, 5)
    q = fit(lambda z: np.where(z < 1e-12, 1.0, np.log1p(z) / np.where(z == 0, 1, z)), 0.0, 1.0, 8)
    return e, q


def fma32(a, b, c):
    """Exactly rounded float32 fma of arrays (the double-rounding cases of float64 emulation are avoided by
    Boldo-Melquiond's round-to-odd trick being unnecessary here: only used to MEASURE the error)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def T32(a, e, q):
    a = a.astype(np.float32)
    t = a * np.float32(-1.4426950408889634)
    r = np.rint(t)
    f = t - r
    p = np.full_like(f, e[5])
    for k in (4, 3, 2, 1, 0):
        p = fma32(p, f, np.float32(e[k]))
    p = fma32(p, f, np.float32(1.0))
    ex = np.ldexp(p, r.astype(np.int32)).astype(np.float32)
    s = np.full_like(ex, q[8])
    for k in range(7, -1, -1):
        s = fma32(s, ex, np.float32(q[k]))
    return ex * s


if __name__ == "__main__":
    e, q = coefficients()
    print("E:", ", ".join(f"0x{v.view(np.uint32):08x}u /* {v:.9g} */" for v in e))
    print("Q:", ", ".join(f"0x{v.view(np.uint32):08x}u /* {v:.9g} */" for v in q))
    a = np.concatenate([np.linspace(0, 60, 2000001), np.logspace(-8, 1.78, 500001)])
    ref = np.log1p(np.exp(-a.astype(np.float32).astype(np.float64)))
    err = np.abs(T32(a, e, q).astype(np.float64) - ref)
    print("max abs error of T:", err.max(), "at a =", a[err.argmax()], " max rel:", np.max(err / ref))
