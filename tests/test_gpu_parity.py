"""GPU parity tests: the HIP path (through the C ABI, via pyglm_amd.engine) against the oracle and against the golden
vectors captured from the reference.  Integer work (Philox words, adjacency decisions) must be bit-exact; fp64 work
within the tolerances written at each assertion (north_star: weight posteriors within 1e-5 relative)."""
import ctypes

import numpy as np
import pytest

from oracle import pyglm_oracle as orc
from tests._pg_agree import assert_pg_agree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _engine(*a, **k):
    from pyglm_amd.engine import GibbsEngine
    return GibbsEngine(*a, **k)


# ----------------------------------------------------------------------------------------------- RNG + PG
def test_philox_words_bit_exact(torch_dev):
    torch = torch_dev
    from pyglm_amd._lib import call, ptr
    n = 5000
    out = torch.zeros(n, 4, dtype=torch.int32, device="cuda:0")
    seed, stream = 0x1234567890ABCDEF, orc.stream_id(77, 5)
    call("pgl_philox_words", seed, 1, 3, 10, stream, ptr(out), n, None)
    got = out.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, orc.philox_words(seed, 1, 3, 10, stream, n))


@pytest.mark.parametrize("zscale", [0.5, 3.0, 12.0])
def test_pg_draw_matches_oracle(torch_dev, zscale):
    torch = torch_dev
    from pyglm_amd._lib import call, ptr
    n = 200000
    rng = np.random.default_rng(1)
    z = rng.standard_normal(n) * zscale
    z[:5] = [0.0, 1e-12, -40.0, 40.0, 1.5625]      # 1/t boundary and extremes
    zd = torch.from_numpy(z).cuda()
    out = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    seed, stream = 42, orc.stream_id(3, 9)
    call("pgl_pg_draw", None, ptr(zd), ptr(out), n, seed, stream, 17, None)
    got = out.cpu().numpy()
    want = orc.pg_draw(None, z, seed, stream, 17)
    assert np.all(np.isfinite(got)) and np.all(got > 0)
    # libm vs OCML differ by ulps inside exp/log/erfc, which can flip an accept/reject on a knife edge: measured 0 of 3e7 (tests/_pg_agree.py)
    assert_pg_agree(got, want)
    # integer b (negative-binomial shape): sum of b PG(1) draws on the same stream
    b = rng.integers(0, 5, n).astype(np.float64)
    bd = torch.from_numpy(b).cuda()
    call("pgl_pg_draw", ptr(bd), ptr(zd), ptr(out), n, seed, stream, 0, None)
    got = out.cpu().numpy()
    want = orc.pg_draw(b, z, seed, stream, 0)
    assert_pg_agree(got, want)
    assert np.all(got[b == 0] == 0)


def _device_pg(b, z, n, seed, stream):
    import torch
    from pyglm_amd._lib import call, ptr
    zd = torch.full((n,), float(z), dtype=torch.float64, device="cuda:0")
    bd = torch.full((n,), float(b), dtype=torch.float64, device="cuda:0")
    out = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    call("pgl_pg_draw", ptr(bd), ptr(zd), ptr(out), n, seed, stream, 0, None)
    return out.cpu().numpy()


@pytest.mark.parametrize("b", [1.0, 2.0, 7.0, 50.0, 0.3, 1.5, 2.5, 13.7, 70.5])
def test_device_pg_analytic_checks(torch_dev, b):
    """the DEVICE sampler against known answers, not against its twin in the oracle: mean, variance and the Laplace transform
    E exp(-t w) = (cosh(z/2) / cosh(sqrt((z^2/2 + t)/2)))^b over z in {0, 0.3, 2, 6, 20, 40}, integer and real shapes b"""
    from tests.test_oracle_pg import pg_mean, pg_var, pg_laplace, Z_GRID
    n = 300000
    for iz, z in enumerate(Z_GRID):
        om = _device_pg(b, z, n, 17, orc.stream_id(iz, int(b * 10)))
        assert np.all(om > 0) and np.all(np.isfinite(om))
        m, v = float(pg_mean(b, z)), float(pg_var(b, z))
        assert abs(om.mean() - m) < 5 * np.sqrt(v / n), (b, z)
        assert abs(om.var() - v) < 0.03 * v, (b, z)
        for t in (0.5 / m, 2.0 / m):
            g = np.exp(-t * om)
            assert abs(g.mean() - pg_laplace(b, z, t)) < 5 * g.std() / np.sqrt(n), (b, z, t)


@pytest.mark.parametrize("b,z", [(1.0, 0.0), (1.0, 6.0), (2.0, 0.3), (7.0, 2.0), (50.0, 20.0), (0.3, 0.0), (1.5, 1.0), (2.5, 2.0), (13.7, 40.0)])
def test_device_pg_ks_against_gamma_series(torch_dev, b, z):
    """Kolmogorov-Smirnov of device draws against the defining sum-of-gammas series (600 terms, NumPy)"""
    from scipy import stats
    from tests.test_oracle_pg import gamma_series_sample
    n = 20000
    ref = gamma_series_sample(b, z, n, np.random.default_rng(int(b * 100 + z)))
    om = _device_pg(b, z, n, 5, orc.stream_id(1, 2))
    assert stats.ks_2samp(om, ref).pvalue > 1e-3


@pytest.mark.parametrize("b", [13.7, 50.0, 170.0])
@pytest.mark.parametrize("z", [0.0, 2.0, 20.0])
def test_device_pg_series_branch_on_two_million_draws(torch_dev, b, z):
    """large shapes of the device sampler (13.7 is 12 exact Devroye draws + one exact draw of PG(1.7, z) from the alternate sampler, 50 is
    exact, 170 the series alone: 32 terms + a moment-matched gamma remainder) at
    n = 2e6 per case: mean, variance and THIRD cumulant against the exact values of the defining series (each within 5 standard errors
    of the sample statistic), and a two-sample Kolmogorov-Smirnov test against an independent sampler of the same series -- 2000 terms,
    gamma variates from torch's generator on the GPU (n = 1e6).  tests/test_oracle_pg.py bounds analytically what the truncation changes
    (nothing in cumulants 1-2, 4e-11 .. 1.4e-6 of the third); this is the empirical side."""
    torch = torch_dev
    from scipy import stats
    from tests.test_oracle_pg import pg_cumulant
    n = 2000000
    om = _device_pg(b, z, n, 23, orc.stream_id(int(b), int(z) + 1))
    assert np.all(om > 0) and np.all(np.isfinite(om))
    k1, k2, k3 = (pg_cumulant(b, z, j) for j in (1, 2, 3))
    xc = om - om.mean()
    m2, m3, m4, m6 = (float(np.mean(xc ** j)) for j in (2, 3, 4, 6))
    assert abs(om.mean() - k1) < 5 * np.sqrt(k2 / n)
    assert abs(m2 - k2) < 5 * np.sqrt((m4 - m2 * m2) / n)
    se3 = np.sqrt(max(m6 - m3 * m3 - 6 * m4 * m2 + 9 * m2 ** 3, 0.0) / n)
    assert abs(m3 - k3) < 5 * se3, (m3, k3, se3)
    # independent series sampler on the GPU, in chunks: w = sum_k g_k / d_k over 2000 terms + the (deterministic) mean of what is left
    gen = torch.Generator(device="cuda:0").manual_seed(int(1000 * b + z))
    K, nref, chunk = 2000, 1000000, 50000
    c = z * z / (4 * np.pi ** 2)
    dk = torch.tensor(2 * np.pi ** 2 * ((np.arange(1, K + 1) - 0.5) ** 2 + c), device="cuda:0")
    tail = k1 - b * float((1.0 / dk).sum())
    conc = torch.full((chunk, K), float(b), dtype=torch.float64, device="cuda:0")
    ref = torch.cat([(torch._standard_gamma(conc, generator=gen) / dk).sum(1) + tail for _ in range(nref // chunk)]).cpu().numpy()
    res = stats.ks_2samp(om[:nref], ref)
    assert res.pvalue > 1e-3, (b, z, res)


@pytest.mark.parametrize("b", [1.5, 2.5, 13.7, 1.003, 1.997])
def test_device_pg_alternate_sampler_matches_oracle(torch_dev, b):
    """shapes with a fractional part in [1, 64] are exact on both sides: floor(b) - 1 Devroye draws + ONE draw of PG(1 + frac, z) from Windle's
    alternate sampler (pgl_rng.h: pgl_pg_alt; oracle/pg_oracle.c: pg_alt, written separately), on the shared stream: device = oracle at 1e-12
    but for knife-edge accept/reject decisions (libm vs OCML ulps); regression.py:479-489, 501-508 hand such shapes to pgdrawvpar"""
    torch = torch_dev
    from pyglm_amd._lib import call, ptr
    n = 100000
    rng = np.random.default_rng(int(b * 1000))
    z = rng.standard_normal(n) * 4.0
    z[:6] = [0.0, 1e-12, -40.0, 40.0, 2.0 * (1.0 + (b % 1.0)) / 1.2, 90.0]
    zd = torch.from_numpy(z).cuda()
    bd = torch.full((n,), b, dtype=torch.float64, device="cuda:0")
    out = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    call("pgl_pg_draw", ptr(bd), ptr(zd), ptr(out), n, 11, orc.stream_id(6, 2), 3, None)
    got = out.cpu().numpy()
    want = orc.pg_draw(np.full(n, b), z, 11, orc.stream_id(6, 2), 3)
    assert np.all(np.isfinite(got)) and np.all(got > 0)
    assert_pg_agree(got, want)


def test_device_pg_real_shapes_match_oracle(torch_dev):
    """real-valued shapes on a shared stream, every branch mixed: the series branch (b < 1, b > 64) consumes the stream identically on both
    sides, but the remainder's moments are computed by different formulas (closed form on the device, term-by-term sums in the oracle), hence
    1e-8 instead of 1e-12"""
    torch = torch_dev
    from pyglm_amd._lib import call, ptr
    n = 60000
    rng = np.random.default_rng(4)
    z = rng.standard_normal(n) * 4.0
    b = np.where(rng.random(n) < 0.5, rng.random(n) * 3.0, rng.random(n) * 40.0)
    b[:8] = [0.0, 1e-3, 12.0, 12.000001, 64.0, 64.000001, 63.5, 170.25]      # incl. both sides of the exact-sum limit
    zd, bd = torch.from_numpy(z).cuda(), torch.from_numpy(b).cuda()
    out = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    call("pgl_pg_draw", ptr(bd), ptr(zd), ptr(out), n, 9, orc.stream_id(2, 3), 5, None)
    got = out.cpu().numpy()
    want = orc.pg_draw(b, z, 9, orc.stream_id(2, 3), 5)
    assert got[0] == 0.0 and np.all(np.isfinite(got)) and np.all(got[1:] > 0)
    # (shapes below 0.05: the U^(1/b) boost of the gamma variates multiplies an ulp of log U by 1/b -- a rounding sensitivity, not a decision)
    small = b < 0.05
    assert_pg_agree(got[~small], want[~small], tol=1e-8)
    assert_pg_agree(got[small], want[small], tol=1e-5)


@pytest.mark.parametrize("b", [13.0, 50.0, 64.0])
def test_device_pg_integer_shapes_are_exact_sums(torch_dev, b):
    """integer shapes up to 64 (negative-binomial counts y + xi) are floor(b) EXACT Devroye draws of PG(1, z) on the shared stream -- no
    series, no approximation (pypolyagamma's samplers are exact rejection samplers too, regression.py:501-508): device = oracle at 1e-12.
    A draw is b accept/reject chains, so the knife-edge allowance of the PG(1) test scales with b."""
    torch = torch_dev
    from pyglm_amd._lib import call, ptr
    n = 40000
    rng = np.random.default_rng(int(b))
    z = rng.standard_normal(n) * 3.0
    zd = torch.from_numpy(z).cuda()
    bd = torch.full((n,), b, dtype=torch.float64, device="cuda:0")
    out = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    call("pgl_pg_draw", ptr(bd), ptr(zd), ptr(out), n, 3, orc.stream_id(4, 1), 2, None)
    got = out.cpu().numpy()
    want = orc.pg_draw(np.full(n, b), z, 3, orc.stream_id(4, 1), 2)
    assert_pg_agree(got, want)
    from tests.test_oracle_pg import pg_mean, pg_var
    om = _device_pg(b, 2.0, 200000, 29, orc.stream_id(int(b), 7))
    assert abs(om.mean() - pg_mean(b, 2.0)) < 5 * np.sqrt(pg_var(b, 2.0) / om.size)
    assert abs(om.var() - pg_var(b, 2.0)) < 0.03 * pg_var(b, 2.0)


def test_pg_moments_at_scale(torch_dev):
    torch = torch_dev
    from pyglm_amd._lib import call, ptr
    n = 4000000
    for z in (0.0, 2.0):
        zd = torch.full((n,), z, dtype=torch.float64, device="cuda:0")
        out = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        call("pgl_pg_draw", None, ptr(zd), ptr(out), n, 7, 1, 0, None)
        m = 0.25 if z == 0 else np.tanh(z / 2) / (2 * z)
        v = 1 / 24.0 if z == 0 else (np.sinh(z) - z) / (4 * z ** 3 * np.cosh(z / 2) ** 2)
        assert abs(out.mean().item() - m) < 5 * np.sqrt(v / n)
        assert abs(out.var().item() - v) < 0.01 * v


# ----------------------------------------------------------------------------------------------- design matrix
def test_design_matrix_golden(torch_dev, golden):
    S, basis = golden["G2_S"], golden["G2_basis"]
    T, N = S.shape
    eng = _engine(N, basis.shape[1])
    eng.add_data(S, basis=basis)
    np.testing.assert_allclose(eng.design_matrix(), golden["G2_F"], rtol=1e-12, atol=1e-14)
    ds = eng.datasets[0]
    X = ds.X.cpu().numpy()
    assert np.all(X[:T, eng.D] == 1.0) and np.all(X[T:] == 0) and np.all(X[:, eng.D + 1:] == 0)
    np.testing.assert_array_equal(ds.Xt.cpu().numpy()[:eng.D + 1, :T], X[:T, :eng.D + 1].T)
    # signed basis: no clipping
    Sb, bb = golden["G2b_S"], golden["G2b_basis"]
    eng = _engine(Sb.shape[1], bb.shape[1])
    eng.add_data(Sb, basis=bb)
    np.testing.assert_allclose(eng.design_matrix(), golden["G2b_F"], rtol=1e-12, atol=1e-13)


# ----------------------------------------------------------------------------------------------- one regression, golden
def _hyp(reg_list):
    from pyglm_amd.engine import prior_terms
    S_w = np.array([r.S_w for r in reg_list])
    mu_w = np.array([r.mu_w for r in reg_list])
    S_b = np.array([r.S_b[0, 0] for r in reg_list])
    mu_b = np.array([r.mu_b[0] for r in reg_list])
    rho = np.array([r.rho for r in reg_list])
    return (rho,) + prior_terms(S_w, mu_w, S_b, mu_b)


@pytest.mark.parametrize("gram", ["fp64", "int8"])
@pytest.mark.parametrize("tag", ["c0", "c1", "c2", "c3"])
def test_regression_resample_golden(torch_dev, golden, tag, gram):
    """the reference's own vectors (tests/golden), with the likelihood Gram on the fp64 kernel and as integer arithmetic on the int8 MFMA"""
    g = golden
    N, B = g[tag + "_mu_w"].shape
    r = orc.Regression(N, B, rho=g[tag + "_rho"], mu_w=g[tag + "_mu_w"], S_w=g[tag + "_S_w"], mu_b=g[tag + "_mu_b"], S_b=g[tag + "_S_b"])
    eng = _engine(N, B, 0, 1, gram=gram)
    datas = [(g[tag + "_X"], g[tag + "_y"]), (g[tag + "_X2"], g[tag + "_y2"])]
    for X, y in datas:
        Y = np.zeros((len(y), N))
        Y[:, 0] = y
        eng.add_data(Y, X=X)
    a0, W0, b0 = g[tag + "_a0"][None], g[tag + "_W0"][None], g[tag + "_b0"]
    # activation + per-bin log-likelihood sum (G9, G10)
    np.testing.assert_allclose(eng.psi(a0, W0, b0)[:, 0], g[tag + "_psi"], rtol=1e-12, atol=1e-13)
    ll = eng.log_likelihood(a0, W0, b0)
    r.a, r.W, r.b = a0[0].copy(), W0[0].copy(), b0.copy()
    want_ll = sum(r.log_likelihood(X, y).sum() for X, y in datas)
    np.testing.assert_allclose(ll[0], want_ll, rtol=1e-11)
    rho, Jw, hw, Jb, hb, c0 = _hyp([r])
    oms = [g[tag + "_om1"][:, None], g[tag + "_om2"][:, None]]
    a1, W1, b1, _ = eng.sweep(a0, W0, b0, rho, Jw, hw, Jb, hb, c0, g[tag + "_perm"][None], g[tag + "_u"][None], g[tag + "_z"][None],
                              seed=1, sweep=0, omega_override=oms)
    Jp, hp = eng.posterior(0)
    np.testing.assert_allclose(Jp, g[tag + "_J_prior"] + g[tag + "_J_lkhd"], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(hp, g[tag + "_h_prior"] + g[tag + "_h_lkhd"], rtol=1e-11, atol=1e-10)
    np.testing.assert_array_equal(a1[0], g[tag + "_a1"])                      # decisions: exact
    np.testing.assert_allclose(W1[0], g[tag + "_W1"], rtol=1e-8, atol=1e-10)   # posteriors: << 1e-5 rel
    np.testing.assert_allclose(b1, g[tag + "_b1"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("tag", ["c0", "c1", "c2"])
def test_flip_logodds_golden(torch_dev, golden, tag):
    """(a) the log-odds of the golden resample() against the oracle's replay of it (the oracle reproduces the reference's decisions and
    `_marginal_likelihood` values: tests/test_oracle_golden.py);  (b) against the REFERENCE's own `_marginal_likelihood` numbers
    (regression.py:343-378, fixture G6): walking the adjacency row from mask i to mask i+1 with forced decisions, the log-odds of the
    steps that flip telescope to ml(mask i+1) - ml(mask i)."""
    g = golden
    N, B = g[tag + "_mu_w"].shape
    r = orc.Regression(N, B, rho=g[tag + "_rho"], mu_w=g[tag + "_mu_w"], S_w=g[tag + "_S_w"], mu_b=g[tag + "_mu_b"], S_b=g[tag + "_S_b"])
    eng = _engine(N, B, 0, 1)
    datas = [(g[tag + "_X"], g[tag + "_y"]), (g[tag + "_X2"], g[tag + "_y2"])]
    for X, y in datas:
        Y = np.zeros((len(y), N))
        Y[:, 0] = y
        eng.add_data(Y, X=X)
    a0, W0, b0 = g[tag + "_a0"][None], g[tag + "_W0"][None], g[tag + "_b0"]
    rho, Jw, hw, Jb, hb, c0 = _hyp([r])
    oms = [g[tag + "_om1"][:, None], g[tag + "_om2"][:, None]]
    eng.keep_logodds = True
    a1, _, _, _ = eng.sweep(a0, W0, b0, rho, Jw, hw, Jb, hb, c0, g[tag + "_perm"][None], g[tag + "_u"][None], g[tag + "_z"][None],
                            seed=1, sweep=0, omega_override=oms)
    np.testing.assert_array_equal(a1[0], g[tag + "_a1"])
    r.a, r.W, r.b = a0[0].copy(), W0[0].copy(), b0.copy()
    trace = []
    r.resample(datas, [g[tag + "_om1"], g[tag + "_om2"]], g[tag + "_perm"], g[tag + "_u"], g[tag + "_z"], trace=trace)
    _check_logodds(eng, [(None, None, None, trace)])
    # (b) telescoping sums against the reference's marginal likelihoods; rho = 1/2 makes the prior-odds term vanish
    masks, mls = g[tag + "_ml_masks"].astype(bool), g[tag + "_ml"]
    half = np.full((1, N), 0.5)
    perm = np.arange(N, dtype=np.int32)[None]
    for i in range(len(masks) - 1):
        src, dst = masks[i], masks[i + 1]
        u = np.where(dst, 1.0 - 1e-15, 0.0)[None]                       # sample_discrete_from_log then returns dst[m] (if |log-odds| < 34)
        a2, _, _, _ = eng.sweep(src[None], W0, b0, half, Jw, hw, Jb, hb, c0, perm, u, g[tag + "_z"][None], seed=1, sweep=0, omega_override=oms)
        lo = eng.logodds.cpu().numpy()[0]
        assert np.all(np.abs(lo) < 34.0) and np.array_equal(a2[0], dst), (tag, i)
        total = lo[dst & ~src].sum() - lo[src & ~dst].sum()
        np.testing.assert_allclose(total, mls[i + 1] - mls[i], rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("gram", ["fp64", "int8"])
def test_model_sweep_golden(torch_dev, golden, gram):
    g = golden
    N, _, B = g["M_W0"].shape
    eng = _engine(N, B, batch=3, gram=gram)        # 4 neurons in batches of 3 + 1
    eng.add_data(g["M_Y"], basis=g["M_basis"])
    np.testing.assert_allclose(eng.design_matrix(), g["M_X"], rtol=1e-10, atol=1e-13)
    ll0 = eng.log_likelihood(g["M_A0"], g["M_W0"], g["M_b0"])
    np.testing.assert_allclose(ll0.sum(), g["M_ll0"], rtol=1e-11)
    regs = [orc.Regression(N, B, S_w=10.0, mu_b=-2.0) for _ in range(N)]
    rho, Jw, hw, Jb, hb, c0 = _hyp(regs)
    a1, W1, b1, llb = eng.sweep(g["M_A0"], g["M_W0"], g["M_b0"], rho, Jw, hw, Jb, hb, c0, g["M_perms"], g["M_us"], g["M_zs"], seed=5, sweep=0,
                                omega_override=[g["M_omegas"].T])
    np.testing.assert_allclose(llb.sum(), g["M_ll0"], rtol=1e-11)
    np.testing.assert_array_equal(a1, g["M_A1"])
    np.testing.assert_allclose(W1, g["M_W1"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(b1, g["M_b1"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(eng.log_likelihood(a1, W1, b1).sum(), g["M_ll1"], rtol=1e-10)


# ----------------------------------------------------------------------------------------------- bigger seeded cases vs the oracle
def _random_problem(N, B, T, seed, rho=0.5, p_spike=0.1, n_active_true=3):
    rng = np.random.default_rng(seed)
    basis = orc.cosine_basis(B, L=20) / 20
    Y = (rng.random((T, N)) < p_spike).astype(float)
    X = orc.convolve_with_basis(Y, basis)
    # make some targets depend on their inputs so that the sampler has something to find
    Wtrue = np.zeros((N, N, B))
    for n in range(N):
        for m in rng.choice(N, n_active_true, replace=False):
            Wtrue[n, m] = rng.standard_normal(B) * 3
    psi = X.reshape(T, -1) @ Wtrue.reshape(N, -1).T - 1.5
    Y2 = (rng.random((T, N)) < orc.logistic(psi)).astype(float)
    return basis, X, Y2, rng


def _check_logodds(eng, outs, rows=None):
    """the flip log-odds the kernel formed (lps[1] - lps[0] of regression.py:293-307, exported per proposal step) against the oracle's
    trace: bit-equal decisions only bound their error to ~1e-3, so the numbers themselves are compared (1e-9 absolute + 1e-10 relative)"""
    lo = eng.logodds.cpu().numpy()
    for n, out in enumerate(outs):
        if rows is not None and n not in rows:
            continue
        trace = out[3]
        want = np.array([t[1] for t in trace])
        got = lo[n, :len(trace)]
        assert np.array_equal(np.isnan(got), np.isnan(want)), n            # rho in {0, 1}: 0 log 0 = NaN on both sides
        ok = ~np.isnan(want)
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-10, atol=1e-9, err_msg="log-odds of row %d" % n)
        assert np.isnan(lo[n, len(trace):]).all()                          # no proposals beyond the trace


def _with_state(r, a, W, b):
    r.a, r.W, r.b = a, W, b
    return r


def _oracle_sweep(N, B, X, Y, a, W, b, kw, omegas, perm, u, z):
    outs = []
    for n in range(N):
        r = orc.Regression(N, B, **kw)
        r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
        trace = []
        r.resample([(X, Y[:, n])], [omegas[:, n]], perm[n], u[n], z[n], trace=trace)
        outs.append((r.a.copy(), r.W.copy(), r.b.copy(), trace))
    return outs


@pytest.mark.parametrize("N,B,T,rho,batch", [(12, 2, 700, 0.5, None), (60, 3, 1500, 0.5, 16), (40, 4, 900, 1.0, 7), (33, 5, 1200, 0.3, 33),
                                             (4, 1, 10000, 0.5, None), (10, 3, 6100, 0.5, 4), (25, 5, 2300, 0.4, 25)])   # (the last three: the fp64 Gram of a small model in time slices)
def test_sweep_vs_oracle(torch_dev, N, B, T, rho, batch):
    from pyglm_amd.engine import make_draws
    basis, X, Y, rng = _random_problem(N, B, T, seed=N * 7 + B)
    kw = dict(rho=rho, S_w=4.0, mu_w=0.0, mu_b=-1.5, S_b=2.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * a[:, :, None]
    b = rng.standard_normal(N) - 1.5
    eng = _engine(N, B, batch=batch)
    eng.add_data(Y, X=X)
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    rho_a, Jw, hw, Jb, hb, c0 = _hyp(regs)
    perm, u, z = make_draws(123, 4, range(N), N, N * B)
    eng.keep_logodds = True
    a1, W1, b1, _ = eng.sweep(a, W, b, rho_a, Jw, hw, Jb, hb, c0, perm, u, z, seed=123, sweep=4)
    omegas = eng.datasets[0].OK[:T, :N].cpu().numpy()
    # the PG draws themselves: the device's, from ITS activation (MFMA summation order), against the oracle's from NumPy's, on the same
    # stream -- every neuron (measured: 0 of 2.9e6 such draws differ although 72 % of the activations are not bit-equal, tests/_pg_agree.py)
    want = np.column_stack([orc.pg_draw(None, _with_state(regs[n], a[n], W[n], b[n:n + 1]).activation(X), 123, orc.stream_id(n, 4))
                            for n in range(N)])
    assert_pg_agree(omegas, want, what="omega of a sweep (N=%d, T=%d)" % (N, T))
    outs = _oracle_sweep(N, B, X, Y, a, W, b, kw, omegas, perm, u, z)
    for n, (ao, Wo, bo, trace) in enumerate(outs):
        np.testing.assert_array_equal(a1[n], ao, err_msg="adjacency row %d" % n)
        np.testing.assert_allclose(W1[n], Wo, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(b1[n], bo[0], rtol=1e-7, atol=1e-9)
    _check_logodds(eng, outs)
    # the flips really were exercised
    if rho < 1:
        assert (a1 != a).sum() > 0


def test_sweep_vs_oracle_large_initial_active_set(torch_dev):
    """initial active sets of ~380 rows: the tableau is initialised in chunks of 256 pivots (pivot-block inverse by 2 x 2 blocks
    of 128, rank-256 updates) plus a remainder chunk; neurons with fewer than 129 / 257 rows share the same launches"""
    from pyglm_amd.engine import make_draws
    N, B, T = 110, 4, 2500
    basis, X, Y, rng = _random_problem(N, B, T, seed=5)
    kw = dict(rho=0.6, S_w=4.0, mu_w=0.0, mu_b=-1.5, S_b=2.0)
    a = rng.random((N, N)) < 0.85
    a[3] = rng.random(N) < 0.2             # 23 blocks: a single sub-128 chunk
    a[7] = rng.random(N) < 0.45            # ~200 rows: one chunk with a partial second block
    a[11] = True                           # all 441 rows: 256 + 185
    W = rng.standard_normal((N, N, B)) * 0.3 * a[:, :, None]
    b = rng.standard_normal(N) - 1.5
    eng = _engine(N, B, batch=64)
    eng.add_data(Y, X=X)
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    rho_a, Jw, hw, Jb, hb, c0 = _hyp(regs)
    perm, u, z = make_draws(77, 1, range(N), N, N * B)
    eng.keep_logodds = True
    a1, W1, b1, _ = eng.sweep(a, W, b, rho_a, Jw, hw, Jb, hb, c0, perm, u, z, seed=77, sweep=1)
    omegas = eng.datasets[0].OK[:T, :N].cpu().numpy()
    outs = _oracle_sweep(N, B, X, Y, a, W, b, kw, omegas, perm, u, z)
    for n, (ao, Wo, bo, trace) in enumerate(outs):
        np.testing.assert_array_equal(a1[n], ao, err_msg="adjacency row %d" % n)
        np.testing.assert_allclose(W1[n], Wo, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(b1[n], bo[0], rtol=1e-7, atol=1e-9)
    _check_logodds(eng, outs)
    assert (a1 != a).sum() > 100


def test_visit_order_and_plain_tableau_agree(torch_dev):
    """the tableau in proposal order with trailing-only window updates (default) and the plain one (J's order, full updates) take
    the same decisions and draw the same weights; several windows (N = 150 > 64 blocks per window at B = 5)"""
    from pyglm_amd.engine import make_draws
    N, B, T = 150, 5, 2500
    basis, X, Y, rng = _random_problem(N, B, T, seed=9)
    kw = dict(rho=0.5, S_w=0.05, mu_w=0.0, mu_b=-1.5, S_b=2.0)          # a tight slab: many flips per window
    a = rng.random((N, N)) < 0.6
    W = rng.standard_normal((N, N, B)) * 0.1 * a[:, :, None]
    b = rng.standard_normal(N) - 1.5
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    rho_a, Jw, hw, Jb, hb, c0 = _hyp(regs)
    perm, u, z = make_draws(31, 2, range(N), N, N * B)
    outs = []
    for vo in (True, False):
        eng = _engine(N, B, batch=40, visit_order=vo)
        eng.add_data(Y, X=X)
        outs.append(eng.sweep(a, W, b, rho_a, Jw, hw, Jb, hb, c0, perm, u, z, seed=31, sweep=2)[:3])
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-9, atol=1e-12)
    assert (outs[0][0] != a).sum() > 1000
    # and against the oracle for a few rows
    omegas = eng.datasets[0].OK[:T, :N].cpu().numpy()
    for n in (0, 77, N - 1):
        r = orc.Regression(N, B, **kw)
        r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
        r.resample([(X, Y[:, n])], [omegas[:, n]], perm[n], u[n], z[n])
        np.testing.assert_array_equal(outs[0][0][n], r.a)
        np.testing.assert_allclose(outs[0][1][n], r.W, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("N,B,rho", [(12, 3, 0.5), (4, 1, 0.5), (19, 5, 0.3), (96, 1, 0.6), (6, 16, 0.5)])
def test_small_model_tail_in_one_launch_agrees_with_the_general_path(torch_dev, N, B, rho):
    """round 6: a model whose tableau has at most 98 rows takes flips + weight draw as ONE launch per batch with the tableau in LDS
    (pgl_small.hip); an engine with visit_order=False keeps the general path (pivot chunks, proposal windows, blocked Cholesky) at the same
    size.  Two implementations of regression.py:282-340: the same decisions, log-odds to 1e-9, weights to 1e-9 -- and both equal to the oracle."""
    from pyglm_amd.engine import make_draws
    T = 1500
    basis, X, Y, rng = _random_problem(N, B, T, seed=N + B, n_active_true=min(3, N))
    kw = dict(rho=rho, S_w=0.5, mu_w=0.0, mu_b=-1.5, S_b=2.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.3 * a[:, :, None]
    b = rng.standard_normal(N) - 1.5
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    hyp = _hyp(regs)
    perm, u, z = make_draws(13, 1, range(N), N, N * B)
    outs, los = [], []
    for vo in (True, False):
        eng = _engine(N, B, visit_order=vo, batch=5)
        eng.add_data(Y, X=X)
        eng.keep_logodds = True
        outs.append(eng.sweep(a, W, b, *hyp, perm, u, z, seed=13, sweep=1))
        los.append(eng.logodds.cpu().numpy())
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-9, atol=1e-11)
    np.testing.assert_array_equal(np.isnan(los[0]), np.isnan(los[1]))
    np.testing.assert_allclose(np.nan_to_num(los[0]), np.nan_to_num(los[1]), rtol=1e-10, atol=1e-9)
    assert (outs[0][0] != a).sum() > 0
    omegas = eng.datasets[0].OK[:T, :N].cpu().numpy()
    ref = _oracle_sweep(N, B, X, Y, a, W, b, kw, omegas, perm, u, z)
    for n, (ao, Wo, bo, trace) in enumerate(ref):
        np.testing.assert_array_equal(outs[0][0][n], ao)
        np.testing.assert_allclose(outs[0][1][n], Wo, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("gram,T", [("fp64", 600), ("int8", 600), ("fp64", 5000)])
def test_sharded_equals_unsharded(torch_dev, gram, T):
    """neuron sharding is invisible: two engines over [0,5) and [5,11) reproduce one engine over [0,11) bit for bit (with the fp64 Gram -- also
    where a small model's Gram is cut into time slices: their number follows from T and D alone -- and with the integer Gram, whose neurons
    travel in groups: scales, planes and residues are per neuron, so the grouping cannot matter)"""
    from pyglm_amd.engine import make_draws
    N, B = 11, 2
    basis, X, Y, rng = _random_problem(N, B, T, seed=5)
    kw = dict(rho=0.4, S_w=3.0, mu_w=0.0, mu_b=-1.0, S_b=1.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * a[:, :, None]
    b = rng.standard_normal(N)
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    hyp = _hyp(regs)
    res, post = {}, {}
    for (lo, hi) in [(0, N), (0, 5), (5, N)]:
        eng = _engine(N, B, lo, hi, gram=gram)
        eng.add_data(Y, basis=basis)
        perm, u, z = make_draws(9, 2, range(lo, hi), N, N * B)
        sl = slice(lo, hi)
        res[(lo, hi)] = eng.sweep(a[sl], W[sl], b[sl], *[h[sl] for h in hyp], perm, u, z, seed=9, sweep=2)
        post[(lo, hi)] = [eng.posterior(n - lo)[0][:N * B, :N * B] for n in ((0, 7) if (lo, hi) == (0, N) else (0,) if lo == 0 else (7,))]
    full = res[(0, N)]
    for k in range(4):
        joined = np.concatenate([res[(0, 5)][k], res[(5, N)][k]])
        np.testing.assert_array_equal(joined, full[k])
    # (the likelihood Gram on its own: the slices of a small model's Gram do not follow the shard either)
    low = np.tril(np.ones((N * B, N * B), dtype=bool))
    np.testing.assert_array_equal(post[(0, 5)][0][low], post[(0, N)][0][low])
    np.testing.assert_array_equal(post[(5, N)][0][low], post[(0, N)][1][low])


@pytest.mark.parametrize("N,B,T,rho", [(1, 3, 40, 0.5), (2, 1, 5, 0.5), (20, 10, 333, 0.5), (7, 32, 200, 0.6), (3, 2, 17, 0.0), (73, 1, 300, 1.0),
                                      (50, 5, 400, 1.0)])
def test_edge_shapes_vs_oracle(torch_dev, N, B, T, rho):
    """edges: a single neuron, T below one 16-row K tile / not a multiple of 16, the reference's default B = 10, the
    largest supported B (= 32: 10 blocks per proposal window), rho = 0 everywhere (deterministic: all connections off), and two all-on
    cases whose active dimension (74, 251) ends a few rows into a 64-row Cholesky sub-panel with the padded leading dimension (80, 256)
    shorter than the sub-panel: a strip update must not write past a neuron's own rows"""
    from pyglm_amd.engine import make_draws
    rng = np.random.default_rng(100 + N + B)
    X = np.abs(rng.standard_normal((T, N, B))) * 0.4
    Y = (rng.random((T, N)) < 0.3).astype(float)
    kw = dict(rho=rho, S_w=2.0, mu_w=0.0, mu_b=-0.5, S_b=1.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.5 * a[:, :, None]
    b = rng.standard_normal(N) * 0.3
    eng = _engine(N, B)
    eng.add_data(Y, X=X)
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    hyp = _hyp(regs)
    perm, u, z = make_draws(77, 1, range(N), N, N * B)
    eng.keep_logodds = True
    a1, W1, b1, ll = eng.sweep(a, W, b, *hyp, perm, u, z, seed=77, sweep=1)
    om = eng.datasets[0].OK[:T, :N].cpu().numpy()
    outs = _oracle_sweep(N, B, X, Y, a, W, b, kw, om, perm, u, z)
    for n, (ao, Wo, bo, _) in enumerate(outs):
        np.testing.assert_array_equal(a1[n], ao)
        np.testing.assert_allclose(W1[n], Wo, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(b1[n], bo[0], rtol=1e-7, atol=1e-9)
    _check_logodds(eng, outs)
    if rho == 0.0:
        assert not a1.any() and np.all(W1 == 0)


@pytest.mark.parametrize("gram", ["auto", "int8"])
def test_three_datasets_accumulate(torch_dev, gram):
    """data_list may hold several datasets whose sufficient statistics add (regression.py:237-260); the PG element index
    continues across datasets.  int8: every data set's product goes through its own planes, scales (batch norms) and CRT, the CRTs adding up in J"""
    from pyglm_amd.engine import make_draws
    rng = np.random.default_rng(9)
    N, B = 6, 2
    Ts = [150, 33, 400]
    Xs = [np.abs(rng.standard_normal((T, N, B))) * 0.4 for T in Ts]
    Ys = [(rng.random((T, N)) < 0.3).astype(float) for T in Ts]
    kw = dict(rho=0.5, S_w=2.0, mu_w=0.1, mu_b=-0.5, S_b=1.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.5 * a[:, :, None]
    b = rng.standard_normal(N) * 0.3
    eng = _engine(N, B, gram=gram)
    for X, Y in zip(Xs, Ys):
        ds = eng.add_data(Y, X=X)
        assert ds.int8 == (gram == "int8")
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    hyp = _hyp(regs)
    perm, u, z = make_draws(5, 3, range(N), N, N * B)
    a1, W1, b1, ll = eng.sweep(a, W, b, *hyp, perm, u, z, seed=5, sweep=3)
    oms = [ds.OK[:ds.T, :N].cpu().numpy() for ds in eng.datasets]
    off = 0
    for X, Y, om in zip(Xs, Ys, oms):              # the draws: element index = running time bin over the datasets
        r = regs[2]
        r.a, r.W, r.b = a[2], W[2], b[2:3]
        want = orc.pg_draw(None, r.activation(X), 5, orc.stream_id(2, 3), off)
        assert (np.abs(om[:, 2] - want) <= 1e-12 * want).mean() >= 0.99
        off += X.shape[0]
    for n in range(N):
        r = orc.Regression(N, B, **kw)
        r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
        np.testing.assert_allclose(ll[n], sum(r.log_likelihood(X, Y[:, n]).sum() for X, Y in zip(Xs, Ys)), rtol=1e-10)
        r.resample([(X, Y[:, n]) for X, Y in zip(Xs, Ys)], [om[:, n] for om in oms], perm[n], u[n], z[n])
        np.testing.assert_array_equal(a1[n], r.a)
        np.testing.assert_allclose(W1[n], r.W, rtol=1e-7, atol=1e-9)


def test_mixed_deterministic_and_sampled_rows_in_one_batch(torch_dev):
    """regression.py:153-155 / 274-275 is decided per regression: rows whose rho is all 0/1 take a = round(rho) and consume no
    uniforms, the others run the collapsed flips -- both kinds inside the same device batch"""
    from pyglm_amd.engine import make_draws
    rng = np.random.default_rng(31)
    N, B, T = 14, 3, 500
    X = np.abs(rng.standard_normal((T, N, B))) * 0.4
    Y = (rng.random((T, N)) < 0.3).astype(float)
    rho = np.full((N, N), 0.5)
    rho[2] = 1.0                                   # dense row
    rho[5] = (rng.random(N) < 0.5).astype(float)   # fixed 0/1 pattern
    rho[9] = 0.0                                   # empty row: bias only
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.5 * a[:, :, None]
    b = rng.standard_normal(N) * 0.3
    kw = dict(S_w=2.0, mu_w=0.0, mu_b=-0.5, S_b=1.0)
    regs = [orc.Regression(N, B, rho=rho[n], **kw) for n in range(N)]
    hyp = _hyp(regs)
    eng = _engine(N, B, batch=9)
    eng.add_data(Y, X=X)
    perm, u, z = make_draws(3, 0, range(N), N, N * B)
    a1, W1, b1, _ = eng.sweep(a, W, b, *hyp, perm, u, z, seed=3, sweep=0)
    om = eng.datasets[0].OK[:T, :N].cpu().numpy()
    for n, r in enumerate(regs):
        r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
        r.resample([(X, Y[:, n])], [om[:, n]], perm[n], u[n], z[n])
        np.testing.assert_array_equal(a1[n], r.a, err_msg="row %d" % n)
        np.testing.assert_allclose(W1[n], r.W, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(b1[n], r.b[0], rtol=1e-7, atol=1e-9)
    assert a1[2].all() and not a1[9].any() and np.array_equal(a1[5], rho[5].astype(bool))


@pytest.mark.parametrize("batch", [2, 12])
def test_split_border_contraction_two_datasets_nan_scratch(torch_dev, batch):
    """the border sums [Omega|Kappa]'[X, 1] (regression.py:253-260) are split over time slices whose partial sums pass through the batch's
    J buffer (pgl_sweep.hip): several slices with a ragged tail (Tp is not a multiple of 16 S), a second data set that accumulates, a
    small batch so that the buffer caps the number of slices -- with J and the tableau pre-filled with NaN, which must not leak into
    the border, the posterior or the draw."""
    import torch
    from pyglm_amd.engine import make_draws
    rng = np.random.default_rng(17)
    N, B = 12, 2
    Ts = [2500, 1310]
    Xs = [np.abs(rng.standard_normal((T, N, B))) * 0.4 for T in Ts]
    Ys = [(rng.random((T, N)) < 0.3).astype(float) for T in Ts]
    kw = dict(rho=0.5, S_w=2.0, mu_w=0.1, mu_b=-0.5, S_b=1.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.5 * a[:, :, None]
    b = rng.standard_normal(N) * 0.3
    eng = _engine(N, B, batch=batch)
    for X, Y in zip(Xs, Ys):
        eng.add_data(Y, X=X)
    eng.Jbuf.fill_(float("nan"))
    eng.Mtab.fill_(float("nan"))
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    hyp = _hyp(regs)
    perm, u, z = make_draws(5, 3, range(N), N, N * B)
    a1, W1, b1, ll = eng.sweep(a, W, b, *hyp, perm, u, z, seed=5, sweep=3)
    D = N * B
    want = sum(ds.OK[:ds.Tp].T.double() @ ds.X[:ds.Tp] for ds in eng.datasets)          # (2 ldn, Dp), rows = omega sums then kappa sums
    got = eng.border
    np.testing.assert_allclose(got[:, :D + 1].cpu().numpy(), want[:, :D + 1].cpu().numpy(), rtol=1e-12, atol=1e-12)
    assert np.isfinite(W1).all() and np.isfinite(b1).all()
    oms = [ds.OK[:ds.T, :N].cpu().numpy() for ds in eng.datasets]
    for n in range(N):
        r = orc.Regression(N, B, **kw)
        r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
        r.resample([(X, Y[:, n]) for X, Y in zip(Xs, Ys)], [om[:, n] for om in oms], perm[n], u[n], z[n])
        np.testing.assert_array_equal(a1[n], r.a)
        np.testing.assert_allclose(W1[n], r.W, rtol=1e-7, atol=1e-9)


def test_small_batches_of_a_narrow_model_give_the_bits_of_one_batch(torch_dev):
    """the number of time slices of the border sums was clamped by what THIS shard's scratch holds (nb ldj^2 doubles), so a shard swept
    in several small batches (batch << nloc at small D) got other last bits than the same neurons in one batch.  Now the slice count follows
    from T and D alone and a cramped scratch takes the neurons in column groups: batch = 2, 7 and 100 agree bit for bit (D = 100, T = 20 000:
    64 slices of the 200 x 112 border are 1.4 M doubles, two neurons' worth of scratch holds 0.5 M)."""
    from pyglm_amd.engine import make_draws
    N, B, T = 100, 1, 20000
    basis, X, Y, rng = _random_problem(N, B, T, seed=3)
    kw = dict(rho=0.5, S_w=3.0, mu_w=0.0, mu_b=-1.0, S_b=1.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * a[:, :, None]
    b = rng.standard_normal(N)
    hyp = _hyp([orc.Regression(N, B, **kw) for _ in range(N)])
    perm, u, z = make_draws(4, 1, range(N), N, N * B)
    outs = []
    for batch in (100, 2, 7):
        eng = _engine(N, B, batch=batch, gram="fp64")
        eng.add_data(Y, X=X)
        outs.append(eng.sweep(a, W, b, *hyp, perm, u, z, seed=4, sweep=1))
        border = eng.border.cpu().numpy().copy()
        outs[-1] = outs[-1] + (border,)
    for o in outs[1:]:
        for x, y in zip(o, outs[0]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("gram", ["fp64", "int8"])
def test_prefix_run_equals_the_same_neurons_of_a_full_sweep(torch_dev, gram):
    """engine.sweep(nrun=k) -- what bench.py's scaling_proxy times: one rank's share of a larger job -- sweeps the first k local neurons
    exactly as the full sweep does (bit for bit) and leaves the others as they came in"""
    from pyglm_amd.engine import make_draws
    N, B, T = 13, 2, 700
    basis, X, Y, rng = _random_problem(N, B, T, seed=8)
    kw = dict(rho=0.4, S_w=3.0, mu_w=0.0, mu_b=-1.0, S_b=1.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * a[:, :, None]
    b = rng.standard_normal(N)
    hyp = _hyp([orc.Regression(N, B, **kw) for _ in range(N)])
    perm, u, z = make_draws(9, 2, range(N), N, N * B)
    eng = _engine(N, B, gram=gram, batch=4)
    eng.add_data(Y, basis=basis)
    full = eng.sweep(a, W, b, *hyp, perm, u, z, seed=9, sweep=2)
    # ... and any range [f, f + k) of them (round 6: pgl_sweep_t.nfirst -- bench.py's scaling_proxy times EVERY shard of a G-rank job, not only
    # the first): batched from f on, as the rank owning that shard would batch it; the rows outside stay as they were
    for f, k in ((0, 5), (0, 8), (4, 6), (6, 7), (2, 11), (12, 1)):
        part = eng.sweep(a, W, b, *hyp, perm, u, z, seed=9, sweep=2, nrun=k, nfirst=f)
        for x, y, x0 in zip(part[:3], full[:3], (a, W, b)):
            np.testing.assert_array_equal(x[f:f + k], y[f:f + k])
            np.testing.assert_array_equal(x[:f], x0[:f])
            np.testing.assert_array_equal(x[f + k:], x0[f + k:])
        np.testing.assert_array_equal(part[3][f:f + k], full[3][f:f + k])
    with pytest.raises(Exception):
        eng.sweep(a, W, b, *hyp, perm, u, z, seed=9, sweep=2, nrun=3, nfirst=5)          # (odd start: refused by pgl_sweep)


def test_two_windows_per_pass_give_the_bits_of_one_pass_per_window(torch_dev):
    """pgl_k_flip_apply_pair stacks the panels of two proposal windows and passes over the trailing tableau once: every entry must see the same
    multiply-adds in the same order as with a pass per window (pgl_sweep_t.flip_single_pass = 1).  N = 200: four windows (64, 64, 64, 8
    blocks), one batch ragged."""
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    N, B, T = 200, 5, 1500
    D = N * B
    rng = np.random.default_rng(4)
    Y = (rng.random((T, N)) < 0.1).astype(float)
    X = rng.random((T, N, B)) * (rng.random((T, N, B)) < 0.3) * 0.3
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.1 * a[:, :, None]
    b = np.full(N, -2.0)
    eng = GibbsEngine(N, B, gram="fp64", batch=64)
    eng.keep_logodds = True
    eng.add_data(Y, X=X)
    perm, u, z = make_draws(9, 0, range(N), N, D)
    res = []
    for single in (False, True):
        eng.flip_single_pass = single
        out = {}
        for tag, S in (("few", 4.0), ("many", 0.02)):      # a loose slab: few flips per window; a tight one: most blocks flip (first panels too long to pair)
            hyp = prior_terms(np.tile(np.eye(B) * S, (N, N, 1, 1)), np.zeros((N, N, B)), np.ones(N), np.full(N, -2.0))
            a1, W1, b1, ll = eng.sweep(a, W, b, np.full((N, N), 0.5), *hyp, perm, u, z, seed=9, sweep=0)
            out.update({tag + "_a": a1, tag + "_W": W1, tag + "_b": b1, tag + "_lo": eng.logodds.cpu().numpy()})
        res.append(out)
    assert set(res[0]) == set(res[1])
    for k in res[0]:
        np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)
    flips_few = (res[0]["few_a"] != res[0]["many_a"]).sum()
    assert flips_few > 0            # (the two regimes are different chains)



@pytest.mark.parametrize("N,B,T,nds", [(4, 1, 10000, 1), (9, 3, 4100, 2), (24, 5, 33000, 1), (12, 10, 2048, 1), (40, 5, 6000, 1), (30, 10, 4096, 2)])
def test_small_model_gram_in_time_slices(torch_dev, N, B, T, nds):
    """pgl_sweep cuts the fp64 Gram of a small model (D <= 512: a few tiles per neuron) into time slices that run as separate work items and are
    added in slice order (pgl_gram_split): the posterior system against NumPy with the GPU's own omega, over one and two data sets"""
    from pyglm_amd.engine import make_draws
    basis, X, Y, rng = _random_problem(N, B, T, seed=N + T)
    D = N * B
    eng = _engine(N, B, gram="fp64")
    Ts = [T] if nds == 1 else [T - 1500, 1500]
    t0 = 0
    for Ti in Ts:
        eng.add_data(Y[t0:t0 + Ti], X=X[t0:t0 + Ti])
        t0 += Ti
    kw = dict(rho=0.5, S_w=4.0, mu_w=0.0, mu_b=-1.5, S_b=2.0)
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.3 * a[:, :, None]
    b = rng.standard_normal(N) - 1.5
    perm, u, z = make_draws(5, 0, range(N), N, D)
    eng.sweep(a, W, b, *_hyp(regs), perm, u, z, seed=5, sweep=0)
    Xf = X.reshape(T, D)
    t0 = 0
    om = []
    for ds in eng.datasets:
        om.append(ds.OK[:ds.T, :N].cpu().numpy())
    om = np.concatenate(om)
    for n in (0, N - 1):
        Jg, hg = eng.posterior(n)
        Jp0, hp0 = regs[n].prior_stats()
        want = Jp0[:D, :D] + (Xf * om[:, n:n + 1]).T @ Xf
        low = np.tril(np.ones((D, D), dtype=bool))
        np.testing.assert_allclose(Jg[:D, :D][low], want[low], rtol=1e-11, atol=1e-11 * np.abs(want).max())
