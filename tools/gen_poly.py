#!/usr/bin/env python3
"""Generates the polynomial coefficients used by the device math helpers in
nmrfit_amd/csrc/objective.hip (exp2 on [-1/2,1/2], sin/cos on [-pi/4,pi/4]) by Chebyshev
interpolation in 60-digit arithmetic, and reports the float64-evaluated max relative error."""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def cheb_fit(f, a, b, deg):
    """Interpolate f at deg+1 Chebyshev nodes on [a,b]; return monomial coeffs (low->high) in x."""
    n = deg + 1
    xs = [mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    ts = [(a + b) / 2 + (b - a) / 2 * x for x in xs]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, t in enumerate(ts):
        for j in range(n):
            A[i, j] = t ** j
        y[i] = f(t)
    c = mp.lu_solve(A, y)
    return [c[i] for i in range(n)]


def horner64(c, x):
    p = np.full_like(x, float(c[-1]))
    for ck in reversed(c[:-1]):
        p = p * x + float(ck)
    return p


def report(name, c, f, a, b):
    x = np.linspace(float(a), float(b), 20001)
    approx = horner64(c, x)
    exact = np.array([float(f(mp.mpf(float(t)))) for t in x])
    rel = np.max(np.abs(approx - exact) / np.maximum(np.abs(exact), 1e-300))
    print("// %s: degree %d, max rel err (float64 Horner) %.3g" % (name, len(c) - 1, rel))
    for i, ck in enumerate(c):
        print("    %s,  // x^%d" % (mp.nstr(ck, 20), i))


if __name__ == "__main__":
    half = mp.mpf(1) / 2
    for deg in (10, 11, 12):
        c = cheb_fit(lambda t: mp.power(2, t), -half, half, deg)
        report("exp2(f), f in [-1/2,1/2]", c, lambda t: mp.power(2, t), -half, half)
    # sin(x)/x and cos(x) as polynomials in y = x^2 on [0, (pi/4)^2]
    q = (mp.pi / 4) ** 2
    for deg in (6, 7):
        c = cheb_fit(lambda y: mp.sin(mp.sqrt(y)) / mp.sqrt(y) if y != 0 else mp.mpf(1), mp.mpf(0), q, deg)
        report("sin(x)/x in y=x^2", c, lambda y: mp.sin(mp.sqrt(y)) / mp.sqrt(y) if y != 0 else mp.mpf(1), mp.mpf(0), q)
        c = cheb_fit(lambda y: mp.cos(mp.sqrt(y)), mp.mpf(0), q, deg)
        report("cos(x) in y=x^2", c, lambda y: mp.cos(mp.sqrt(y)), mp.mpf(0), q)
    print("pio2_hi/lo split:")
    pio2 = mp.pi / 2
    h1 = float(pio2)
    # Cody-Waite 3-part split with trailing zeros: take 33 bits per part
    import math
    def trunc_bits(x, bits):
        m, e = math.frexp(float(x))
        return math.ldexp(math.floor(m * 2 ** bits) / 2 ** bits, e)
    p1 = trunc_bits(pio2, 33); r = pio2 - mp.mpf(p1)
    p2 = trunc_bits(r, 33); r2 = r - mp.mpf(p2)
    p3 = float(r2)
    print("    %r, %r, %r" % (p1, p2, p3))
    print("2/pi = %r" % float(2 / mp.pi))
