"""Known-answer tests for oracle/pg_oracle.c (the PG sampler's source, pypolyagamma, is absent from
/root/reference -> pinned analytically). CPU only."""
import numpy as np
import pytest
from scipy import stats

from oracle import pyglm_oracle as orc


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        got = orc.philox_block(ctr, key)
        assert [int(x) for x in got] == want


def test_stream_layout():
    w = orc.philox_words(seed=0x1234567890ABCDEF, purpose=1, j=2, elem0=5, stream=orc.stream_id(7, 3), n=4)
    for i in range(4):
        ref = orc.philox_block([2 | (1 << 24), 5 + i, 7, 3], [0x90ABCDEF, 0x12345678])
        np.testing.assert_array_equal(w[i], ref)


def pg_mean(b, z):
    z = np.asarray(z, float)
    with np.errstate(invalid="ignore", divide="ignore"):
        m = b / (2 * z) * np.tanh(z / 2)
    return np.where(np.abs(z) < 1e-8, b / 4.0, m)


def pg_var(b, z):
    z = np.asarray(z, float)
    with np.errstate(invalid="ignore", divide="ignore"):
        v = b / (4 * z ** 3) * (np.sinh(z) - z) / np.cosh(z / 2) ** 2
    return np.where(np.abs(z) < 1e-4, b / 24.0, v)


@pytest.mark.parametrize("z", [0.0, 0.3, 1.0, -2.5, 6.0, 20.0])
def test_pg1_moments(z):
    n = 400000
    om = orc.pg_draw(None, np.full(n, z), seed=11, stream=orc.stream_id(0, 0))
    assert np.all(om > 0) and np.all(np.isfinite(om))
    m, v = pg_mean(1, abs(z)), pg_var(1, abs(z))
    assert abs(om.mean() - m) < 5 * np.sqrt(v / n)
    assert abs(om.var() - v) < 0.02 * v + 1e-12


@pytest.mark.parametrize("z,t", [(0.0, 1.0), (1.5, 0.7), (4.0, 3.0)])
def test_pg1_laplace_transform(z, t):
    # E exp(-t w) = cosh(z/2) / cosh(sqrt((z^2/2 + t)/2))
    n = 400000
    om = orc.pg_draw(None, np.full(n, z), seed=5, stream=orc.stream_id(3, 1))
    want = np.cosh(z / 2) / np.cosh(np.sqrt((z * z / 2 + t) / 2))
    got = np.exp(-t * om)
    assert abs(got.mean() - want) < 5 * got.std() / np.sqrt(n)


@pytest.mark.parametrize("z", [0.0, 2.0])
def test_pg1_ks_against_gamma_series(z):
    # PG(1,z) = 1/(2 pi^2) sum_k g_k / ((k-1/2)^2 + z^2/(4 pi^2)),  g_k ~ Exp(1); truncate at K terms + mean of tail
    rng = np.random.default_rng(0)
    n, K = 20000, 400
    k = np.arange(1, K + 1)
    denom = (k - 0.5) ** 2 + z * z / (4 * np.pi ** 2)
    series = (rng.exponential(size=(n, K)) / denom).sum(1) / (2 * np.pi ** 2)
    kk = np.arange(K + 1, 200000)
    series += (1.0 / ((kk - 0.5) ** 2 + z * z / (4 * np.pi ** 2))).sum() / (2 * np.pi ** 2)
    om = orc.pg_draw(None, np.full(n, z), seed=99, stream=orc.stream_id(1, 2))
    assert stats.ks_2samp(om, series).pvalue > 1e-3


def test_pg_integer_b_is_sum_and_streams_are_independent():
    n = 200000
    b = np.full(n, 3.0)
    om = orc.pg_draw(b, np.full(n, 1.2), seed=2, stream=orc.stream_id(0, 0))
    assert abs(om.mean() - pg_mean(3, 1.2)) < 5 * np.sqrt(pg_var(3, 1.2) / n)
    a = orc.pg_draw(None, np.full(1000, 0.5), seed=2, stream=orc.stream_id(0, 0))
    a2 = orc.pg_draw(None, np.full(1000, 0.5), seed=2, stream=orc.stream_id(0, 0))
    c = orc.pg_draw(None, np.full(1000, 0.5), seed=2, stream=orc.stream_id(1, 0))
    d = orc.pg_draw(None, np.full(500, 0.5), seed=2, stream=orc.stream_id(0, 0), elem0=500)
    np.testing.assert_array_equal(a, a2)                 # replayable
    assert not np.any(a == c)                            # other neuron -> other stream
    np.testing.assert_array_equal(a[500:], d)            # element index, not call order, keys the draw
    assert orc.pg_draw(np.zeros(4), np.zeros(4), 1, 0).tolist() == [0, 0, 0, 0]   # PG(0, z) = 0
    with pytest.raises(ValueError):
        orc.pg_draw(np.full(4, -0.5), np.zeros(4), 1, 0)


def pg_laplace(b, z, t):
    """E exp(-t w) for w ~ PG(b, z): (cosh(z/2) / cosh(sqrt((z^2/2 + t)/2)))^b, in logs (cosh overflows at z = 40)"""
    lc = lambda x: np.logaddexp(x, -x) - np.log(2.0)
    return np.exp(b * (lc(z / 2) - lc(np.sqrt((z * z / 2 + t) / 2))))


def gamma_series_sample(b, z, n, rng, K=600):
    """reference sample of PG(b, z) from its defining series, K terms + the mean of the remainder (3e-5 of the mean)"""
    k = np.arange(1, K + 1)
    denom = (k - 0.5) ** 2 + z * z / (4 * np.pi ** 2)
    out = np.empty(n)
    for i0 in range(0, n, 2000):
        m = min(2000, n - i0)
        out[i0:i0 + m] = (rng.gamma(b, size=(m, K)) / denom).sum(1)
    kk = np.arange(K + 1, 400000)
    out += b * (1.0 / ((kk - 0.5) ** 2 + z * z / (4 * np.pi ** 2))).sum()
    return out / (2 * np.pi ** 2)


REAL_B = [0.3, 1.0, 1.0 + 1e-7, 1.02, 1.5, 2.0, 2.5, 3.0 - 1e-6, 7.0, 12.0, 13.7, 50.0, 63.99, 70.5]
Z_GRID = [0.0, 0.3, 2.0, 6.0, 20.0, 40.0]


@pytest.mark.parametrize("b", REAL_B)
def test_pg_real_shape_moments_and_laplace(b):
    """PG(b, z) for real-valued b (regression.py:479-489 hands real shapes to pgdrawvpar): mean, variance and the Laplace transform at
    two arguments, over z from 0 to 40, for shapes on every branch (b < 1: the series; integers: exact Devroye sums; 1 < b < 64 with a
    fractional part: Devroye draws + one draw of the alternate sampler; b > 64: the series alone -- also in the cumulant tests below)"""
    n = 120000
    for iz, z in enumerate(Z_GRID):
        om = orc.pg_draw(np.full(n, b), np.full(n, z), seed=31, stream=orc.stream_id(iz, int(b * 10)))
        assert np.all(om > 0) and np.all(np.isfinite(om))
        m, v = pg_mean(b, z), pg_var(b, z)
        assert abs(om.mean() - m) < 5 * np.sqrt(v / n), (b, z)
        assert abs(om.var() - v) < 0.04 * v, (b, z)
        for t in (0.5 / m, 2.0 / m):                     # arguments on the scale of the distribution
            g = np.exp(-t * om)
            assert abs(g.mean() - pg_laplace(b, z, t)) < 5 * g.std() / np.sqrt(n), (b, z, t)


NEAR_INTEGER = [(1.0 + 2.2e-16, 2.0), (1.0 + 2.2e-16, 5.0), (3.0 + 4.4e-16, 5.0), (1.0 + 1e-14, 7.0), (1.0 + 1e-13, 33.0),
                (2.0 - 2.2e-16, 0.0), (1.0 + 1e-9, 60.0), (1.0 + 1.0001e-9, 60.0), (1.1 * 1.1 / 1.21 + 4.0, 1.0)]


@pytest.mark.timeout(60)
@pytest.mark.parametrize("b,z", NEAR_INTEGER)
def test_pg_shapes_at_rounding_distance_from_an_integer_terminate(b, z):
    """with h - 1 at rounding level the truncated-gamma proposal's 1 - c rounded to 0 and nothing was ever accepted (the draw
    spun).  Now 1 - c is formed without the subtraction, shapes within 1e-9 of an integer ARE that integer (same draws as the integer:
    asserted), and every loop is bounded."""
    n = 50000
    om = orc.pg_draw(np.full(n, b), np.full(n, z), seed=8, stream=orc.stream_id(1, 1))
    assert np.all(np.isfinite(om)) and np.all(om > 0)
    bi = np.round(b)
    m, v = pg_mean(bi, z), pg_var(bi, z)
    assert abs(om.mean() - m) < 5 * np.sqrt(v / n)
    if abs(b - bi) < 1e-9:
        np.testing.assert_array_equal(om, orc.pg_draw(np.full(n, bi), np.full(n, z), seed=8, stream=orc.stream_id(1, 1)))


@pytest.mark.timeout(60)
def test_pg_non_finite_activation_gives_nan_not_a_hang():
    for b in (0.5, 1.0, 1.5, 7.25, 80.0):
        for z in (np.nan, np.inf, -np.inf):
            assert np.all(np.isnan(orc.pg_draw(np.full(64, b), np.full(64, z), seed=1, stream=0)))
    assert orc.pg_draw(np.zeros(3), np.full(3, np.nan), seed=1, stream=0).tolist() == [0, 0, 0]      # PG(0, .) = 0 whatever z


@pytest.mark.parametrize("b,z", [(0.3, 0.0), (0.3, 6.0), (1.05, 1.0), (1.5, 0.0), (1.95, 8.0), (2.5, 2.0), (13.7, 0.3), (50.0, 20.0)])
def test_pg_real_shape_ks_against_gamma_series(b, z):
    rng = np.random.default_rng(int(b * 100 + z))
    n = 20000
    ref = gamma_series_sample(b, z, n, rng)
    om = orc.pg_draw(np.full(n, b), np.full(n, z), seed=77, stream=orc.stream_id(2, 5))
    assert stats.ks_2samp(om, ref).pvalue > 1e-3


# ----------------------------------------------------------------------------------------------- what the truncated series changes, exactly
def pg_cumulant(b, z, n, terms=200000):
    """n-th cumulant of PG(b, z) from its defining series w = sum_k g_k / d_k, g_k ~ Gamma(b, 1), d_k = 2 pi^2 ((k - 1/2)^2 + z^2 / 4 pi^2):
    kappa_n = b (n - 1)! sum_k d_k^-n  (a sum of independent gammas), the tail beyond `terms` by the midpoint rule"""
    from math import factorial
    c = z * z / (4 * np.pi ** 2)
    k = np.arange(1, terms + 1, dtype=np.float64)
    d = 2 * np.pi ** 2 * ((k - 0.5) ** 2 + c)
    s = np.sum(d ** -float(n))
    s += (2 * np.pi ** 2) ** -n * terms ** (1.0 - 2 * n) / (2 * n - 1)          # int_K^inf x^-2n dx (c is negligible there)
    return b * factorial(n - 1) * s


def series_sampler_cumulant(b, z, n, K=32):
    """the same cumulant for what the samplers draw for real-valued shapes (pgl_rng.h / pg_oracle.c): the first K terms of the series
    exactly, the remainder as ONE gamma variate with the remainder's exact mean m and variance v: shape m^2 / v, scale v / m, whose
    n-th cumulant is (n - 1)! m (v / m)^(n - 1)"""
    from math import factorial
    c = z * z / (4 * np.pi ** 2)
    k = np.arange(1, K + 1, dtype=np.float64)
    d = 2 * np.pi ** 2 * ((k - 0.5) ** 2 + c)
    head = b * factorial(n - 1) * np.sum(d ** -float(n))
    m = pg_cumulant(b, z, 1) - b * np.sum(1.0 / d)
    v = pg_cumulant(b, z, 2) - b * np.sum(d ** -2.0)
    return head + factorial(n - 1) * m * (v / m) ** (n - 1)


@pytest.mark.parametrize("b", [0.3, 13.7, 50.0, 170.0])
@pytest.mark.parametrize("z", [0.0, 2.0, 20.0])
def test_truncated_series_changes_cumulants_by_parts_per_million_at_most(b, z):
    """For real-valued shapes the samplers truncate the defining series at 32 terms and draw the rest as one moment-matched gamma variate
    (the third-party sampler the reference calls truncates the same series, uncorrected).  That is an approximation -- but every cumulant
    of a sum of independent terms is the sum of theirs, so what it changes can be written down: cumulants 1 and 2 not at all, and the
    n-th by [kappa_n(matched gamma) - kappa_n(true remainder)].  Relative to the distribution's own cumulant, for EVERY b (it cancels):
        z = 0:  -4e-11 (n = 3), -1e-14 (n = 4);   z = 2:  -1e-10, -5e-14;   z = 20:  -1.4e-6, -2e-8, -2e-10 (n = 5)
    (at large z the first terms of the series shrink and the remainder carries 6 % of the mean instead of 0.6 %).  A relative
    difference of 1e-6 in the third cumulant is not detectable on fewer than ~1e12 draws; the device tests (test_gpu_parity.py) check
    mean, variance, third cumulant and KS on 2e6."""
    tol3, tol4, tol5 = {0.0: (1e-10, 1e-13, 1e-13), 2.0: (3e-10, 2e-13, 1e-13), 20.0: (2e-6, 3e-8, 3e-10)}[z]
    for n, tol in ((1, 1e-12), (2, 1e-12), (3, tol3), (4, tol4), (5, tol5)):
        true, got = pg_cumulant(b, z, n), series_sampler_cumulant(b, z, n)
        assert abs(got - true) <= tol * true, (b, z, n, got / true - 1)
    # the series itself against the closed forms of the mean and the variance
    assert abs(pg_cumulant(b, z, 1) - pg_mean(b, z)) < 1e-13 * pg_mean(b, z) and abs(pg_cumulant(b, z, 2) - pg_var(b, z)) < 1e-13 * pg_var(b, z)
    # share of the remainder in the mean and in the variance
    c = z * z / (4 * np.pi ** 2)
    d = 2 * np.pi ** 2 * ((np.arange(1, 33) - 0.5) ** 2 + c)
    assert 1 - b * np.sum(1 / d) / pg_cumulant(b, z, 1) < (0.07 if z == 20.0 else 0.009)
    assert 1 - b * np.sum(d ** -2.0) / pg_cumulant(b, z, 2) < (5e-4 if z == 20.0 else 2e-6)


# ----------------------------------------------------------------------------------------------- the alternate sampler (1 < b < 2), exactly
def _alt_table_in_c():
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "pg_oracle.c")).read()
    body = re.search(r"pg_alt_trunc\[101\] = \{(.*?)\};", src, flags=re.S).group(1)
    return [float(v) for v in body.replace("\n", " ").split(",") if v.strip()]


def test_alt_sampler_switch_points_and_envelope():
    """the table of switch points is where the two proposal pieces cross (regenerated here for a few entries), both pieces dominate the
    density well beyond the switch point (60-digit arithmetic: any switch point in that region gives an exact sampler), and the device header
    carries the same table"""
    import importlib.util
    import os
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_pg_alt_table", os.path.join(here, "golden", "make_pg_alt_table.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    tab = _alt_table_in_c()
    assert len(tab) == 101 and abs(tab[0] - 2 / np.pi) < 1e-4 and all(b > a for a, b in zip(tab, tab[1:]))
    for k in (0, 1, 37, 50, 99, 100):
        assert abs(tab[k] - float(mk.crossing(mk.mp.mpf(1) + mk.mp.mpf(k) / 100))) < 6e-5, k
    for h in (1.004, 1.5, 1.996):          # shapes inside a bin use the bin's left edge
        ml, mr = mk.margins(h, tab[min(100, int((h - 1) * 100))], nl=60, nr=120)
        assert ml >= 0.0 and mr > 0.0, (h, ml, mr)
    dev = open(os.path.join(os.path.dirname(here), "pyglm_amd", "csrc", "pgl_rng.h")).read()
    body = re.search(r"pgl_pg_alt_trunc\[101\] = \{(.*?)\};", dev, flags=re.S).group(1)
    assert [float(v) for v in body.replace("\n", " ").split(",") if v.strip()] == tab


@pytest.mark.parametrize("b,z", [(1.5, 0.0), (1.5, 3.0), (1.07, 0.5), (1.93, 12.0)])
def test_alt_sampler_against_the_exact_distribution_function(b, z):
    """one-sample Kolmogorov-Smirnov of 2e5 draws against the EXACT distribution function of PG(b, z): the alternating density series of
    Windle et al. (in x = 4 omega: cosh^b(z/2) exp(-z^2 x / 8) sum_n (-1)^n a_n(x | b)) integrated with 30-digit arithmetic on a grid, linear
    in between (the grid is fine enough for 1e-4)"""
    import mpmath as mp
    mp.mp.dps = 30
    h, zz = mp.mpf(b), mp.mpf(z) / 2

    def a_n(n, x):
        return mp.power(2, h) * mp.gamma(n + h) / (mp.gamma(n + 1) * mp.gamma(h)) * (2 * n + h) / mp.sqrt(2 * mp.pi * x ** 3) * mp.exp(-(2 * n + h) ** 2 / (2 * x))

    def dens(x):
        s, n = mp.mpf(0), 0
        while True:
            t = a_n(n, x)
            s += (-1) ** n * t
            n += 1
            if (n > 8 and t < mp.mpf(10) ** -25 * abs(s)) or n > 2000:
                break
        return mp.cosh(zz) ** h * mp.exp(-zz * zz * x / 2) * s
    n = 200000
    om = orc.pg_draw(np.full(n, b), np.full(n, z), seed=123, stream=orc.stream_id(5, int(100 * b)))
    x = 4 * om
    probs = np.concatenate(([1e-4, 1e-3, 5e-3], np.linspace(0.01, 0.99, 99), [0.995, 0.999, 0.9999]))
    grid = np.concatenate(([0.0], np.quantile(x, probs), [x.max(), x.max() * 1.5 + 5]))
    cdf = [0.0]
    for lo, hi in zip(grid[:-1], grid[1:]):
        cdf.append(cdf[-1] + float(mp.quad(dens, [max(lo, 1e-9), hi])))
    assert abs(cdf[-1] - 1.0) < 1e-6
    res = stats.kstest(x, lambda q: np.interp(q, grid, cdf))
    assert res.pvalue > 1e-3, res
