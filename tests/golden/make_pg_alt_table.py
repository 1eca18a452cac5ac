#!/usr/bin/env python
"""Switch points t(h) of the alternate Polya-gamma sampler (oracle/pg_oracle.c: pg_alt_trunc, pyglm_amd/csrc/pgl_rng.h: pgl_pg_alt_trunc) and
the check that makes any of them safe to use.

In the scale x = 4 omega the PG(h, 0) density is  f(x | h) = sum_n (-1)^n a_n(x | h),
    a_n(x | h) = 2^h Gamma(n + h) / (Gamma(n + 1) Gamma(h)) (2n + h) / sqrt(2 pi x^3) exp(-(2n + h)^2 / (2x))
(Windle, Polson & Scott 2014, arXiv 1405.0506).  The sampler proposes from  left(x) = a_0(x | h)  below t and from
right(x) = (pi/2)^h x^(h-1) exp(-pi^2 x / 8) / Gamma(h)  above it; it is exact as long as left >= f on (0, t] and right >= f on [t, inf) (the
tilt exp(-z^2 x / 2) multiplies all three).  t(h) is taken where the two pieces cross -- the smallest envelope -- for h = 1.00, 1.01, ..., 2.00
(a shape h uses the entry of floor((h - 1) * 100)), rounded to four decimals.

    python tests/golden/make_pg_alt_table.py          prints the table and the dominance margins (60-digit arithmetic, mpmath)
"""
import mpmath as mp

mp.mp.dps = 60


def a_n(n, x, h):
    return mp.power(2, h) * mp.gamma(n + h) / (mp.gamma(n + 1) * mp.gamma(h)) * (2 * n + h) / mp.sqrt(2 * mp.pi * x ** 3) * mp.exp(-(2 * n + h) ** 2 / (2 * x))


def density(x, h):
    s, n = mp.mpf(0), 0
    while True:
        t = a_n(n, x, h)
        s += (-1) ** n * t
        n += 1
        if (n > 10 and t < mp.mpf(10) ** -(mp.mp.dps - 5) * abs(s)) or n > 4000:
            return s


def left(x, h):
    return a_n(0, x, h)


def right(x, h):
    return mp.power(mp.pi / 2, h) * x ** (h - 1) * mp.exp(-mp.pi ** 2 * x / 8) / mp.gamma(h)


def crossing(h):
    return mp.findroot(lambda x: mp.log(left(x, h)) - mp.log(right(x, h)), 0.64 + 1.4 * (h - 1))


def table():
    return [round(float(crossing(mp.mpf(1) + mp.mpf(k) / 100)), 4) for k in range(101)]


def margins(h, t, nl=200, nr=600):
    """(min of left / f - 1 on (0, 1.3 t], min of right / f - 1 on [0.7 t, 30 t]): both pieces dominate well beyond the switch point"""
    h, t = mp.mpf(h), mp.mpf(t)
    ml = min(left(x, h) / density(x, h) - 1 for x in [t * mp.mpf("1.3") * i / nl for i in range(max(1, nl // 50), nl + 1)])
    mr = min(right(x, h) / density(x, h) - 1 for x in [t * (mp.mpf("0.7") + mp.mpf(i) / 20) for i in range(nr)])
    return float(ml), float(mr)


if __name__ == "__main__":
    tab = table()
    for i in range(0, 101, 10):
        print("    " + ", ".join("%.4f" % v for v in tab[i:i + 10]) + ",")
    for h in (1.001, 1.1, 1.3, 1.5, 1.7, 1.999):
        k = min(100, int((h - 1) * 100))
        print("h = %.3f  t = %.4f  min(left / f - 1), min(right / f - 1) = %.3e, %.3e" % ((h, tab[k]) + margins(h, tab[k])))
