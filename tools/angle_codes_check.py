#!/usr/bin/env python
"""CPU check (numpy, fp32 emulation) of the division-free angle codes of dd_attention2.hip::angle_codes (DD_FAST_ANGLE):
all 11 AngularEncoding codes [th, sin{1,2,3}th, sin th/2, sin th/3, cos{1,2,3}th, cos th/2, cos th/3] against float64, next to
the error of the straightforward fp32 pipeline (atan2f, then sinf / cosf of fp32 products -- what the reference's
models/common.py:38-53 evaluates).  Also re-derives the atan polynomial.  No GPU needed.
usage: python tools/angle_codes_check.py"""
import numpy as np

f32 = np.float32
ATAN = [9.9999933550e-01, -3.3329860447e-01, 1.9946561855e-01, -1.3908611309e-01, 9.6421528094e-02, -5.5911747027e-02,
        2.1862573855e-02, -4.0544654832e-03]           # atan(x) = x * p(x^2) on [0, 1], ascending powers (kernel constants)


def fit_atan(deg=7):
    n = 4000
    k = np.arange(n)
    u = 0.5 * (1 - np.cos(np.pi * (k + 0.5) / n))
    x = np.sqrt(u)
    f = np.where(x > 0, np.arctan(x) / np.maximum(x, 1e-300), 1.0)
    w = np.ones(n)
    for _ in range(40):                                   # Lawson iteration towards the minimax fit
        A = np.vander(u, deg + 1, increasing=True)
        c, *_ = np.linalg.lstsq(A * w[:, None], f * w, rcond=None)
        err = (A @ c - f) * x
        w *= 1 + 2 * np.abs(err) / np.abs(err).max()
        w /= w.mean()
    return c, np.abs(err).max()


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def kernel_codes(a, b):
    ax, ay, az = a.T
    bx, by, bz = b.T
    dot = ((ax * bx + ay * by).astype(f32) + az * bz).astype(f32)
    cx = (ay * bz - az * by).astype(f32); cy = (az * bx - ax * bz).astype(f32); cz = (ax * by - ay * bx).astype(f32)
    nn = ((cx * cx + cy * cy).astype(f32) + cz * cz).astype(f32)
    n = np.sqrt(nn).astype(f32)
    n2 = fma(dot, dot, nn)
    r = (f32(1) / np.sqrt(n2)).astype(f32)
    s1 = (n * r).astype(f32); c1 = (dot * r).astype(f32)
    s2 = (f32(2) * s1 * c1).astype(f32); c2 = fma(c1, c1, -(s1 * s1).astype(f32))
    s3 = fma(s2, c1, (c2 * s1).astype(f32)); c3 = fma(c2, c1, -(s2 * s1).astype(f32))
    den = (f32(1) + np.abs(c1)).astype(f32)
    x = (s1 * (f32(1) / den).astype(f32)).astype(f32)
    u = (x * x).astype(f32)
    p = np.full_like(x, f32(ATAN[-1]))
    for ck in ATAN[-2::-1]:
        p = fma(p, u, np.full_like(x, f32(ck)))
    phi = (p * x).astype(f32)
    th = np.where(c1 >= 0, (f32(2) * phi).astype(f32), fma(np.full_like(phi, f32(-2)), phi, np.full_like(phi, f32(np.pi))))
    big = np.sqrt((f32(0.5) * den).astype(f32)).astype(f32)
    small = (f32(0.5) * s1 * (f32(1) / big).astype(f32)).astype(f32)
    sh = np.where(c1 >= 0, small, big); ch = np.where(c1 >= 0, big, small)
    a6 = (th * f32(1 / 6)).astype(f32); z = (a6 * a6).astype(f32)
    K = lambda v: np.full_like(z, f32(v))
    ps = fma(fma(fma(K(-1.9515295891e-4), z, K(8.3321608736e-3)), z, K(-1.6666654611e-1)) * z, a6, a6)
    pc = fma(fma(fma(K(2.443315711809948e-5), z, K(-1.388731625493765e-3)), z, K(4.166664568298827e-2)) * z, z, fma(K(-0.5), z, K(1)))
    st = (f32(2) * ps * pc).astype(f32); ct = fma(K(-2) * ps, ps, K(1))
    return np.stack([th, s1, s2, s3, sh, st, c1, c2, c3, ch, ct], 1)


def exact_codes(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    th = np.arctan2(np.linalg.norm(np.cross(a, b), axis=1), (a * b).sum(1))
    return np.stack([th, np.sin(th), np.sin(2 * th), np.sin(3 * th), np.sin(th / 2), np.sin(th / 3), np.cos(th), np.cos(2 * th),
                     np.cos(3 * th), np.cos(th / 2), np.cos(th / 3)], 1)


def libm_fp32_codes(a, b):
    ax, ay, az = a.T
    bx, by, bz = b.T
    dot = ((ax * bx + ay * by).astype(f32) + az * bz).astype(f32)
    cx = (ay * bz - az * by).astype(f32); cy = (az * bx - ax * bz).astype(f32); cz = (ax * by - ay * bx).astype(f32)
    n = np.sqrt(((cx * cx + cy * cy).astype(f32) + cz * cz).astype(f32)).astype(f32)
    th = np.arctan2(n, dot).astype(f32)
    fr = [1, 2, 3, 0.5, 1 / 3]
    return np.stack([th] + [np.sin((th * f32(f)).astype(f32)).astype(f32) for f in fr]
                    + [np.cos((th * f32(f)).astype(f32)).astype(f32) for f in fr], 1)


if __name__ == "__main__":
    c, e = fit_atan()
    print("atan fit (degree 7 in x^2): max error %.2e; coefficients" % e, ["%.10e" % v for v in c])
    assert np.allclose(c, ATAN, rtol=0, atol=2e-9)
    rng = np.random.default_rng(0)
    n = 2_000_000
    a = (rng.standard_normal((n, 3)) * 1.5).astype(f32); b = (rng.standard_normal((n, 3)) * 1.5).astype(f32)
    b[:n // 10] = a[:n // 10] * f32(1.3) + (rng.standard_normal((n // 10, 3)) * 1e-3).astype(f32)            # nearly parallel
    b[n // 10:n // 5] = -a[n // 10:n // 5] * f32(0.7) + (rng.standard_normal((n // 10, 3)) * 1e-3).astype(f32)  # nearly anti-parallel
    ex = exact_codes(a, b)
    names = "th s1 s2 s3 sh st c1 c2 c3 ch ct".split()
    ek = np.abs(kernel_codes(a, b).astype(np.float64) - ex).max(0)
    el = np.abs(libm_fp32_codes(a, b).astype(np.float64) - ex).max(0)
    for nm, x, y in zip(names, ek, el):
        print(f"{nm}: kernel {x:.2e}   fp32 libm pipeline {y:.2e}")
    assert ek.max() < 6e-7
