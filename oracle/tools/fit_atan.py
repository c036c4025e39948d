"""Derives the odd polynomial used by sr_atan (oracle/shader_oracle.cpp and
shader-ray_amd/csrc/trace_common.h) and measures its error.

atan(a) for a in [0, 1] is evaluated as
    a <= tan(pi/8):  P(a)
    a >  tan(pi/8):  pi/4 + P((a - 1) / (a + 1))
with P(z) = z + z*s*(c0 + s*(c1 + s*(c2 + s*c3))), s = z*z, all in float32, one rounding per
operation.  The coefficients are a weighted least-squares fit on Chebyshev nodes (float64),
rounded to float32; the script then evaluates the exact float32 operation sequence with
numpy.float32 on a dense sample and reports the error in ulps against float64 atan.
"""
import numpy as np

T = np.tan(np.pi / 8)


def fit():
    k = np.arange(4000)
    z = T * np.cos(np.pi * (k + 0.5) / 4000)            # Chebyshev nodes on [-T, T]
    z = z[z > 1e-6]
    s = z * z
    # (atan(z) - z) / (z*s) = c0 + c1 s + c2 s^2 + c3 s^3
    target = (np.arctan(z) - z) / (z * s)
    A = np.stack([np.ones_like(s), s, s * s, s ** 3], axis=1)
    w = 1.0 / np.abs(np.arctan(z)) * (z * s)              # relative error weighting
    c, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
    return c.astype(np.float32)


def sr_atan01(a, c):
    """float32 op sequence, a in [0, 1]"""
    f = np.float32
    a = a.astype(f)
    big = a > f(T)
    z = np.where(big, (a - f(1)) / (a + f(1)), a).astype(f)
    s = (z * z).astype(f)
    p = (c[3] * s).astype(f)
    p = ((p + c[2]).astype(f) * s).astype(f)
    p = ((p + c[1]).astype(f) * s).astype(f)
    p = (p + c[0]).astype(f)
    r = (z + ((z * s).astype(f) * p).astype(f)).astype(f)
    return np.where(big, (f(np.pi / 4) + r).astype(f), r)


if __name__ == "__main__":
    c = fit()
    print("coefficients:", ", ".join("%.9gf" % v for v in c))
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(0, 1, 4_000_000), np.linspace(0, 1, 1_000_001),
                        np.float32(T) + np.arange(-2000, 2000) * 3e-8, 10.0 ** rng.uniform(-30, 0, 200_000)]).astype(np.float32)
    got = sr_atan01(a, c).astype(np.float64)
    want = np.arctan(a.astype(np.float64))
    ulp = np.spacing(want.astype(np.float32)).astype(np.float64)
    err = np.abs(got - want) / np.where(ulp > 0, ulp, 1)
    print("max error %.3f ulp at a = %.9g; mean %.3f ulp" % (err.max(), a[err.argmax()], err.mean()))
