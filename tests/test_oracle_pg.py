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


REAL_B = [0.3, 1.0, 2.0, 2.5, 7.0, 12.0, 13.7, 50.0]
Z_GRID = [0.0, 0.3, 2.0, 6.0, 20.0, 40.0]


@pytest.mark.parametrize("b", REAL_B)
def test_pg_real_shape_moments_and_laplace(b):
    """PG(b, z) for real-valued b (regression.py:479-489 hands real shapes to pgdrawvpar): mean, variance and the Laplace transform at
    two arguments, over z from 0 to 40, for shapes on every branch (series only, Devroye only, Devroye + series, series for b > 12)"""
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


@pytest.mark.parametrize("b,z", [(0.3, 0.0), (0.3, 6.0), (2.5, 2.0), (13.7, 0.3), (50.0, 20.0)])
def test_pg_real_shape_ks_against_gamma_series(b, z):
    rng = np.random.default_rng(int(b * 100 + z))
    n = 20000
    ref = gamma_series_sample(b, z, n, rng)
    om = orc.pg_draw(np.full(n, b), np.full(n, z), seed=77, stream=orc.stream_id(2, 5))
    assert stats.ks_2samp(om, ref).pvalue > 1e-3
