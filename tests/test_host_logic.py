"""Host-side logic of the product (no GPU): hyper-parameter plumbing, random-input keys, shard bounds, network prior,
basis construction -- against the oracle / the golden vectors."""
import numpy as np
import pytest

from oracle import pyglm_oracle as orc


def test_cosine_basis_matches_reference_vectors(golden):
    from pyglm_amd.utils.basis import cosine_basis, interpolate_basis
    for (B, L) in [(1, 100), (3, 10), (5, 100)]:
        np.testing.assert_allclose(cosine_basis(B, L=L), golden["G1_cosine_B%d_L%d" % (B, L)], rtol=1e-12, atol=1e-12)
    b = interpolate_basis(cosine_basis(3, L=10), 0.5, 10.0)
    assert b.shape == (21, 3) and np.all(b[0] == 0)
    np.testing.assert_allclose(0.5 * b.sum(0), 1.0)


def test_expand_helpers():
    from pyglm_amd.utils.utils import expand_scalar, expand_cov
    assert expand_scalar(2.0, (3, 2)).shape == (3, 2)
    c = expand_cov(3.0, (4, 2, 2))
    np.testing.assert_array_equal(c, orc.expand_cov(3.0, (4, 2, 2)))
    with pytest.raises(AssertionError):
        expand_scalar(np.zeros(3), (4,))


def test_prior_terms_match_oracle_prior_stats():
    from pyglm_amd.engine import prior_terms
    rng = np.random.default_rng(0)
    N, B = 5, 3
    A = rng.standard_normal((N, B, B))
    S_w = np.einsum("nij,nkj->nik", A, A) + 0.3 * np.eye(B)
    mu_w = rng.standard_normal((N, B))
    r = orc.Regression(N, B, S_w=S_w, mu_w=mu_w, S_b=2.0, mu_b=-1.0)
    Jw, hw, Jb, hb, c0 = prior_terms(S_w[None], mu_w[None], np.array([2.0]), np.array([-1.0]))
    J, h = r.prior_stats()
    for m in range(N):
        np.testing.assert_allclose(J[m * B:(m + 1) * B, m * B:(m + 1) * B], Jw[0, m], rtol=1e-12)
    np.testing.assert_allclose(h[:-1], hw[0].ravel(), rtol=1e-12)
    np.testing.assert_allclose([J[-1, -1], h[-1]], [Jb[0], hb[0]], rtol=1e-12)
    # c0[m] is the prior part of the marginal-likelihood change when block m switches on (regression.py:374,376)
    for m in range(N):
        L0 = np.linalg.cholesky(Jw[0, m])
        want = np.sum(np.log(np.diag(L0))) - 0.5 * hw[0, m].dot(np.linalg.solve(Jw[0, m], hw[0, m]))
        np.testing.assert_allclose(c0[0, m], want, rtol=1e-12)


def test_draws_are_keyed_by_global_neuron():
    from pyglm_amd.engine import make_draws
    p_all, u_all, z_all = make_draws(7, 3, range(0, 6), 9, 18)
    p_lo, u_lo, z_lo = make_draws(7, 3, range(0, 2), 9, 18)
    p_hi, u_hi, z_hi = make_draws(7, 3, range(2, 6), 9, 18)
    np.testing.assert_array_equal(np.concatenate([p_lo, p_hi]), p_all)
    np.testing.assert_array_equal(np.concatenate([u_lo, u_hi]), u_all)
    np.testing.assert_array_equal(np.concatenate([z_lo, z_hi]), z_all)
    assert sorted(p_all[0]) == list(range(9)) and z_all.shape == (6, 19)
    assert not np.array_equal(make_draws(7, 4, [0], 9, 18)[1], u_all[:1])


def test_shard_bounds_cover_and_balance():
    from pyglm_amd.models import shard_bounds
    for N in (1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(N, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == N
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_regression_hyperparameter_setters():
    from pyglm_amd.regression import SparseBernoulliRegression, BernoulliRegression
    np.random.seed(0)
    r = SparseBernoulliRegression(4, 2, rho=0.3, S_w=10.0, mu_b=-2.0)
    assert r.rho.shape == (4,) and r.S_w.shape == (4, 2, 2) and r.mu_w.shape == (4, 2) and r.S_b.shape == (1, 1) and r.mu_b.shape == (1,)
    assert r.a.shape == (4,) and r.W.shape == (4, 2) and r.b.shape == (1,) and np.all(r.W[~r.a] == 0)
    J_w, h_w, J_b, h_b = r.natural_params
    np.testing.assert_allclose(J_w[0], np.eye(2) / 10.0)
    assert not r.deterministic_sparsity
    d = BernoulliRegression(3, 1)
    assert d.deterministic_sparsity and np.all(d.rho == 1)
    r.S_w = 2.0
    np.testing.assert_allclose(r.S_w[1], 2 * np.eye(2))
    # same NumPy stream as the reference's constructor (regression.py:87-92)
    np.random.seed(5)
    r1 = SparseBernoulliRegression(3, 2, mu_b=-2, S_b=0.1)
    np.random.seed(5)
    a = np.random.rand(3) < 0.5
    W = np.array([a[n] * np.random.multivariate_normal(np.zeros(2), np.eye(2)) for n in range(3)])
    np.testing.assert_array_equal(r1.a, a)
    np.testing.assert_array_equal(r1.W, W)


def test_network_prior_shapes_and_niw_update():
    from pyglm_amd.networks import NIWSparseNetwork, NIWDenseNetwork, FixedMeanSparseNetwork, _NIW
    np.random.seed(1)
    N, B = 5, 2
    net = NIWSparseNetwork(N, B)
    assert net.rho.shape == (N, N) and np.all(net.rho == 0.5)
    assert net.mu_W.shape == (N, N, B) and net.sigma_W.shape == (N, N, B, B)
    np.testing.assert_array_equal(net.mu_W[1, 1], net._self_gaussian.mu)
    np.testing.assert_array_equal(net.mu_W[1, 2], net._gaussian.mu)
    np.testing.assert_array_equal(net.sigma_W_rows(2, 4), net.sigma_W[2:4])
    A = np.random.rand(N, N) < 0.5
    W = np.random.randn(N, N, B)
    net.resample((A, W))
    assert np.all(NIWDenseNetwork(N, B).rho == 1)
    from pyglm_amd import networks
    networks.set_reference_quirks(False)
    try:
        assert FixedMeanSparseNetwork(N, B, mu=0.5, sigma=2.0, rho=0.2).sigma_W[0, 1, 1, 1] == 2.0
    finally:
        networks.set_reference_quirks(True)
    with pytest.raises(AssertionError):
        net.resample((A.astype(float), W))
    # posterior parameters of the NIW update equal the oracle's restatement
    data = np.random.randn(40, B) + 1.0
    ours = _NIW(np.zeros(B), np.eye(B), 1.0, 4.0)
    ref = orc.NIWGaussian(np.zeros(B), np.eye(B), 1.0, 4.0, np.random.default_rng(0))
    mu_n, sig_n, k_n, nu_n = ref.posterior(data)
    draws, sigmas = [], []
    for _ in range(4000):
        ours.resample(data)
        draws.append(ours.mu)
        sigmas.append(ours.sigma)
    draws, sigmas = np.array(draws), np.array(sigmas)
    np.testing.assert_allclose(draws.mean(0), mu_n, atol=0.02)
    # second moments of the NIW posterior (published): E[Sigma] = Sigma_n / (nu_n - B - 1), Cov[mu] = E[Sigma] / kappa_n; and a
    # Bartlett check on the scale: tr(Sigma_n Sigma^-1) ~ chi^2 with nu_n B degrees of freedom (mean nu_n B, variance 2 nu_n B)
    ES = sig_n / (nu_n - B - 1)
    np.testing.assert_allclose(sigmas.mean(0), ES, rtol=0.04, atol=0.004)
    np.testing.assert_allclose(np.cov(draws.T), ES / k_n, rtol=0.1, atol=2e-3)
    tr = np.array([np.trace(sig_n @ np.linalg.inv(S)) for S in sigmas])
    assert abs(tr.mean() - nu_n * B) < 4 * np.sqrt(2 * nu_n * B / len(tr)) and abs(tr.var() / (2 * nu_n * B) - 1) < 0.15


def test_network_update_from_row_statistics_equals_the_update_from_the_data():
    """round 6: inside resample_model() the network prior's NIW updates (networks.py:132-149) take the count / sum / sum of outer products of the
    active weight vectors from per-row statistics (pgl_row_stats on the device, exchanged with the rows; host_row_stats here) instead of
    walking the (N, N, B) state on every rank.  Same posterior parameters, same random numbers consumed: the draws agree to rounding with the
    walk over W[A & ~eye] / W[A & eye] that a direct resample_network() call -- and the reference -- makes."""
    from pyglm_amd import networks
    from pyglm_amd.models import SparseBernoulliGLM, host_row_stats
    from tests._oracle_engine import OracleEngine
    rng = np.random.default_rng(3)
    N, B = 12, 3
    A = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.7 + 0.2
    st = host_row_stats(A, W, 0)
    eye = np.eye(N, dtype=bool)
    data = W[A & ~eye]
    np.testing.assert_allclose(st[:, 0].sum(), data.shape[0])
    np.testing.assert_allclose(st[:, 1:1 + B].sum(0), data.sum(0), rtol=1e-13)
    np.testing.assert_allclose(st[:, 1 + B:].sum(0).reshape(B, B), data.T @ data, rtol=1e-13)
    np.testing.assert_array_equal(host_row_stats(A[4:9], W[4:9], 4), st[4:9])            # a row's numbers do not depend on the shard
    g1 = networks._NIW(np.zeros(B), np.eye(B), 1.0, B + 2.0)
    g2 = networks._NIW(np.zeros(B), np.eye(B), 1.0, B + 2.0)
    np.random.seed(5)
    g1.resample(data)
    np.random.seed(5)
    g2.resample_stats(data.shape[0], data.sum(0), data.T @ data)
    np.testing.assert_allclose(g2.mu, g1.mu, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(g2.sigma, g1.sigma, rtol=1e-10, atol=1e-12)
    np.random.seed(5)
    g1.resample(np.zeros((0, B)))
    np.random.seed(5)
    g2.resample_stats(0, np.zeros(B), np.zeros((B, B)))                                   # no active connection: a draw from the prior
    np.testing.assert_array_equal(g2.mu, g1.mu)
    # model level: resample_model() (statistics from the rows) against resample_regressions() + resample_network() (the walk)
    for special in (True, False):
        ms = []
        for k in range(2):
            np.random.seed(2)
            m = SparseBernoulliGLM(6, B=2, regression_kwargs=dict(S_w=3.0, mu_b=-1.0), seed=9, engine_factory=OracleEngine)
            m.network.is_diagonal_weight_special = special
            np.random.seed(4)
            m.add_data((np.random.rand(300, 6) < 0.2).astype(float))
            if k == 0:
                m.resample_model()
            else:
                m.resample_regressions()
                m.resample_network()
            ms.append(m)
        np.testing.assert_array_equal(ms[0].weights, ms[1].weights)
        for name in ("_gaussian", "_self_gaussian"):
            np.testing.assert_allclose(getattr(ms[0].network, name).mu, getattr(ms[1].network, name).mu, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(getattr(ms[0].network, name).sigma, getattr(ms[1].network, name).sigma, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ms[0].regressions[3].S_w, ms[1].regressions[3].S_w, rtol=1e-9)


def test_network_constructors_match_the_reference(golden):
    """fixture G12: the NIW hyper-parameters and connection probabilities the REFERENCE's constructors end up with for a set of keyword
    arguments (most never arrive: networks.py:83, 180, 196), for B on both sides of nu_0 >= B; with set_reference_quirks(False) the
    keywords take effect"""
    import warnings
    from pyglm_amd import networks
    kw = dict(nu_0=7.0, kappa_0=3.0, mu_0=0.5, sigma_0=2.0, rho=0.2, rho_self=0.9)
    for name, cls in (("sparse", networks.NIWSparseNetwork), ("dense", networks.NIWDenseNetwork)):
        for B in (1, 2, 3, 5):
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                net = cls(3, B, **kw)
            assert any("never reach" in str(x.message) for x in w)
            g, s = net._gaussian, net._self_gaussian
            got = np.array([g.nu_0, g.kappa_0, g.mu_0[0], g.sigma_0[0, 0], s.nu_0, s.kappa_0, s.mu_0[0], s.sigma_0[0, 0]])
            want = golden["N_%s_B%d_niw" % (name, B)].copy()
            if B > 3:      # nu_0 = 3 < B: the reference cannot draw from this prior at all; floored (with a warning) so that cfg2/cfg3 run
                assert want[4] == 3.0 and got[4] == B + 2 and any("no proper inverse-Wishart" in str(x.message) for x in w)
                want[4] = B + 2
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(net.rho, golden["N_%s_B%d_rho" % (name, B)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = networks.FixedMeanSparseNetwork(3, 2, mu=0.7, sigma=4.0, rho=0.3)
    np.testing.assert_array_equal(net.mu_W, golden["N_fixed_mu"])
    np.testing.assert_array_equal(net.sigma_W, golden["N_fixed_sigma"])
    np.testing.assert_array_equal(net.rho, golden["N_fixed_rho"])
    networks.set_reference_quirks(False)
    try:
        net = networks.NIWSparseNetwork(3, 2, **kw)
        assert (net._gaussian.nu_0, net._self_gaussian.nu_0, net._gaussian.kappa_0, net._gaussian.mu_0[0], net.rho[0, 1], net.rho[1, 1]) == (7.0, 7.0, 3.0, 0.5, 0.2, 0.9)
        net = networks.FixedMeanSparseNetwork(3, 2, mu=0.7, sigma=4.0, rho=0.3)
        assert net.mu_W[0, 1, 0] == 0.7 and net.sigma_W[0, 1, 1, 1] == 4.0
    finally:
        networks.set_reference_quirks(True)


def test_network_resample_and_push_match_the_reference(golden):
    """fixture G11: the data the reference's resample_network hands to its two Gaussians (models.py:228-231, networks.py:132-149) and
    the hyper-parameters it then pushes into every regression (models.py:233-236), with the NIW draw held at the prior parameters as in
    the fixture (the draw itself is third-party code: see test_network_prior_shapes_and_niw_update for its distribution)"""
    from pyglm_amd import networks
    from pyglm_amd.models import SparseBernoulliGLM
    from tests._oracle_engine import OracleEngine
    g = golden
    N, _, B = g["M_W1"].shape
    seen = {}

    class Fixed(networks._NIW):
        def resample(self, data=()):
            self.mu, self.sigma = self.mu_0.copy(), self.sigma_0.copy()
            seen[id(self)] = np.array(data)
    old = networks._NIW
    networks._NIW = Fixed
    try:
        np.random.seed(3)
        m = SparseBernoulliGLM(N, basis=g["M_basis"], regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=1, engine_factory=OracleEngine)
    finally:
        networks._NIW = old
    for n, r in enumerate(m.regressions):
        r.a, r.W, r.b = g["M_A1"][n].copy(), g["M_W1"][n].copy(), g["M_b1"][n:n + 1].copy()
    m.resample_network()
    np.testing.assert_array_equal(seen[id(m.network._gaussian)].reshape(-1, B), g["M_net_offdiag_data"].reshape(-1, B))
    np.testing.assert_array_equal(seen[id(m.network._self_gaussian)].reshape(-1, B), g["M_net_diag_data"].reshape(-1, B))
    np.testing.assert_array_equal(np.array([r.S_w for r in m.regressions]), g["M_push_S_w"])
    np.testing.assert_array_equal(np.array([r.mu_w for r in m.regressions]), g["M_push_mu_w"])
    np.testing.assert_array_equal(np.array([r.rho for r in m.regressions]), g["M_push_rho"])
    # and the cached natural-parameter terms of that push are what the dense formulas give
    from pyglm_amd.engine import prior_terms
    _, (rho, prior, _, Jb, hb, _) = m._hyper_cache
    Jw, hw, c0 = prior.dense()
    want = prior_terms(g["M_push_S_w"], g["M_push_mu_w"], np.ones(N), np.full(N, -2.0))
    np.testing.assert_allclose(Jw, want[0], rtol=1e-14)
    np.testing.assert_allclose(c0, want[4], rtol=1e-13)
    np.testing.assert_array_equal(rho, g["M_push_rho"])


def test_pushed_hyper_cache_equals_dense_terms():
    """the broadcast fast path taken after resample_network equals prior_terms on the full arrays, and is dropped as soon
    as a hyper-parameter is set by hand"""
    from pyglm_amd.engine import prior_terms
    from pyglm_amd.models import SparseBernoulliGLM
    from tests._oracle_engine import OracleEngine
    np.random.seed(2)
    N, B = 7, 3
    m = SparseBernoulliGLM(N, B=B, regression_kwargs=dict(S_w=3.0, mu_b=-1.0), seed=1, engine_factory=OracleEngine)
    m.resample_network()
    assert type(m.regressions[0]._S_w).__name__ == "_BlockRows"           # pushed as (shared block, own block), expanded only when read
    versions, (rho, prior, _, Jb, hb, _) = m._hyper_cache
    assert versions == tuple(r._hyp_version for r in m.regressions)
    np.testing.assert_array_equal(m.regressions[4].S_w, m.network.sigma_W[4])
    np.testing.assert_array_equal(m.regressions[4].mu_w, m.network.mu_W[4])
    assert prior.Jw_u.shape[0] <= 3 and prior.label.shape == (N, N)      # a handful of distinct blocks, never expanded
    Jw, hw, c0 = prior.dense()
    regs = m.regressions
    want = prior_terms(np.array([r.S_w for r in regs]), np.array([r.mu_w for r in regs]), np.array([r.S_b[0, 0] for r in regs]),
                       np.array([r.mu_b[0] for r in regs]))
    for got, w in zip((Jw, hw, Jb, hb, c0), want):
        np.testing.assert_allclose(got, w, rtol=1e-13, atol=1e-15)
    np.testing.assert_array_equal(rho, np.array([r.rho for r in regs]))
    assert versions != tuple(r._hyp_version for r in regs)      # the public getters handed out live arrays: the cache counts as stale


def test_in_place_hyper_edits_reach_the_next_sweep():
    """`reg.rho[m] = x` edits the live array (as users of the reference do); the cached natural-parameter terms must not outlive it"""
    from pyglm_amd.models import SparseBernoulliGLM
    from tests._oracle_engine import OracleEngine
    np.random.seed(4)
    N, B, T = 4, 2, 300
    Y = (np.random.rand(T, N) < 0.2).astype(float)
    m = SparseBernoulliGLM(N, B=B, regression_kwargs=dict(S_w=3.0, mu_b=-1.0), seed=1, engine_factory=OracleEngine)
    m.add_data(Y)
    m.resample_model()
    m.resample_model()                         # second sweep runs from the cache of the network push
    assert m._hyper_cache[0] == tuple(r._hyp_version for r in m.regressions)
    for r in m.regressions:
        r.rho[:] = 0.0                         # in place, through the getter
        r.rho[1] = 1.0
    m.resample_regressions()
    np.testing.assert_array_equal(m.adjacency, np.tile(np.array([False, True, False, False]), (N, 1)))
    # a handle taken BEFORE earlier sweeps and edited afterwards (no getter call in between: nothing a version counter could see)
    m3 = SparseBernoulliGLM(N, B=B, regression_kwargs=dict(S_w=3.0, mu_b=-1.0), seed=1, engine_factory=OracleEngine)
    m3.add_data(Y)
    held = [r.rho for r in m3.regressions]
    m3.resample_regressions()
    m3.resample_regressions()
    for h in held:
        h[:] = 0.0
        h[2] = 1.0
    m3.resample_regressions()
    np.testing.assert_array_equal(m3.adjacency, np.tile(np.array([False, False, True, False]), (N, 1)))
    # ... and after a network push the arrays are new objects nobody holds: the cache is trusted again
    m3.resample_network()
    assert not any(r._handed_out for r in m3.regressions)
    m3.resample_regressions()
    assert m3._hyper_cache[0] == tuple(r._hyp_version for r in m3.regressions)
    m2 = SparseBernoulliGLM(N, B=B, regression_kwargs=dict(S_w=3.0, mu_b=-1.0), seed=1, engine_factory=OracleEngine)
    m2.add_data(Y)
    m2.resample_regressions()
    m2.regressions[0].S_w[2] *= 100.0          # in-place edit of a covariance block between sweeps
    m2.resample_regressions()
    assert m2._hyper_cache[1][1][0, 2, 0, 0] == pytest.approx(1.0 / 300.0)


def test_api_error_behaviour_matches_reference():
    """bare asserts as in the reference: models.py:68-70, 77 (add_data), regression.py:135 (S_b must be a scalar),
    networks.py:39-41 (resample input types)"""
    from pyglm_amd.models import SparseBernoulliGLM
    from pyglm_amd.regression import SparseBernoulliRegression
    from tests._oracle_engine import OracleEngine
    np.random.seed(0)
    m = SparseBernoulliGLM(3, B=2, seed=1, engine_factory=OracleEngine)
    with pytest.raises(AssertionError):
        m.add_data(np.zeros((10, 4)))                 # wrong number of neurons
    with pytest.raises(AssertionError):
        m.add_data([[0, 1, 0]])                       # not an ndarray
    with pytest.raises(AssertionError):
        m.add_data(np.zeros((10, 3)), X=np.zeros((10, 3, 5)))
    r = SparseBernoulliRegression(3, 2)
    with pytest.raises(AssertionError):
        r.S_b = np.eye(1)
    with pytest.raises(Exception):
        r._flatten_X(np.zeros(5))
    assert m.generate(T=0).shape == (0, 3)
    with pytest.raises(AssertionError):
        m.generate(T=2.5)


def test_draws_made_ahead_equal_draws_made_on_time():
    """the next sweep's permutations / uniforms / normals may be drawn while the GPU is busy (engine.sweep's host_overlap hook):
    the chain must not depend on when they were drawn"""
    from tests._oracle_engine import OracleEngine
    from pyglm_amd.models import SparseBernoulliGLM
    out = []
    for min_size in (1 << 30, 0):
        np.random.seed(4)
        N, B, T = 4, 2, 300
        Y = (np.random.rand(T, N) < 0.2).astype(float)
        m = SparseBernoulliGLM(N, B=B, regression_kwargs=dict(S_w=3.0, mu_b=-1.0), seed=21, engine_factory=OracleEngine)
        m.DRAW_AHEAD_MIN_SIZE = min_size
        m.add_data(Y)
        for _ in range(3):
            m.resample_model()
        assert (m._draws_ahead is not None) == (min_size == 0)
        out.append((m.adjacency.copy(), m.weights.copy(), m.biases.copy()))
    for x, y in zip(*out):
        np.testing.assert_array_equal(x, y)


def test_integer_gram_arithmetic_on_the_host():
    """the two pieces of exact arithmetic the integer-MFMA Gram rests on (pgl_i8gram.hip), replayed with NumPy / Python integers:
    (a) a residue by four fp64 operations -- q = (v / p + M) - M with M = 1.5 * 2^52, r = v - p q: congruent to v and a signed byte for
    every modulus and every |v| < 2^50;  (b) Garner's mixed-radix digits from unreduced partial sums reconstruct S exactly"""
    P = [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197]
    from math import gcd
    assert all(gcd(a, b) == 1 for i, a in enumerate(P) for b in P[i + 1:])
    prod = 1
    for p in P:
        prod *= p
    assert prod > 2 * 100000 * 2 ** 100 and prod > 2 * 112000 * 2 ** 100       # beta = 50 holds the integer Gram up to T = 112 000
    rng = np.random.default_rng(3)
    MAGIC = 6755399441055744.0
    v = np.rint(rng.uniform(-1, 1, 200000) * 2.0 ** 50)
    v[:6] = [0.0, 2.0 ** 50 - 1, -(2.0 ** 50 - 1), 127.5 * 255 // 1, 128.0 * 255, -128.0 * 255]
    vi = v.astype(np.int64)
    for p in P[1:]:
        q = (v * (1.0 / p) + MAGIC) - MAGIC                      # rint(v / p) up to an error of 2^-9.6
        assert np.all(q == np.rint(q))
        r = vi - p * q.astype(np.int64)                          # the fma is exact: the true result is a small integer
        assert np.all((r - vi) % p == 0) and r.min() >= -128 and r.max() <= 127
    low = (vi & 0xff).astype(np.uint8).view(np.int8).astype(np.int64)          # p = 256: the low byte of the integer
    assert np.all((low - vi) % 256 == 0)
    # (b) tables as in ModTable, digits with symmetric representatives, Horner
    NP_ = len(P)
    w = [[0] * NP_ for _ in range(NP_)]
    pinv = [0] * NP_
    for qd in range(1, NP_):
        pr = 1
        for k in range(qd):
            w[k][qd] = pr - P[qd] if pr > P[qd] // 2 else pr
            pr = (pr * (P[k] % P[qd])) % P[qd]
        pinv[qd] = pow(pr, -1, P[qd])

    def trunc_mod(a, p):                                          # C++ % (truncating)
        return -(-a % p) if a < 0 else a % p
    import random
    random.seed(5)
    for _ in range(3000):
        S = random.randrange(-prod // 2 + 2 ** 20, prod // 2 - 2 ** 20)
        d = []
        for p in P:
            r = S % p
            if p == 256:
                r = r - 256 if r >= 128 else r
            else:
                r = r - p if r > p // 2 else r
                if random.random() < 0.1 and -128 <= r + p <= 127:
                    r += p                                        # a non-minimal representative, as the planes kernel may produce
                if random.random() < 0.1 and -128 <= r - p <= 127:
                    r -= p
            d.append(r)
        # the GEMM re-reduces to the symmetric representative before the CRT
        d = [d[0]] + [((x + p // 2) % p) - p // 2 for x, p in zip(d[1:], P[1:])]
        for qd in range(1, NP_):
            p = P[qd]
            u = d[qd]
            for k in range(qd):
                u -= d[k] * w[k][qd]
            assert abs(u) < 2 ** 18
            t = trunc_mod(u * pinv[qd], p)
            if t > p // 2:
                t -= p
            elif t < -(p // 2):
                t += p
            d[qd] = t
        s = 0
        for qd in range(NP_ - 1, -1, -1):
            s = s * P[qd] + d[qd]
        assert s == S


def test_chain_state_roundtrip_resumes_exactly():
    """get_state() / set_state(): a chain resumed from a snapshot repeats the original bit for bit (random inputs are keyed by
    (seed, sweep, neuron); the network draws by (seed, sweep))"""
    from pyglm_amd.models import SparseBernoulliGLM, SparseGaussianGLM
    from tests._oracle_engine import OracleEngine
    for cls, kw in [(SparseBernoulliGLM, dict(S_w=3.0, mu_b=-1.0)), (SparseGaussianGLM, dict(S_w=3.0, a_0=2.0, b_0=1.0))]:
        np.random.seed(4)
        N, B, T = 4, 2, 250
        Y = (np.random.rand(T, N) < 0.2).astype(float) + (0.3 * np.random.randn(T, N) if cls is SparseGaussianGLM else 0.0)
        m = cls(N, B=B, regression_kwargs=kw, seed=7, engine_factory=OracleEngine)
        m.add_data(Y)
        for _ in range(2):
            m.resample_model()
        st = m.get_state()
        for _ in range(2):
            m.resample_model()
        first = (m.adjacency.copy(), m.weights.copy(), m.biases.copy(), m.log_likelihood())
        for r in m.regressions:
            r.W[:] = 5.0                                    # (later edits must not reach the snapshot)
        m.set_state(st)
        assert m.sweeps_done == 2 and not np.any(m.weights == 5.0)
        for _ in range(2):
            m.resample_model()
        np.testing.assert_array_equal(m.adjacency, first[0])
        np.testing.assert_array_equal(m.weights, first[1])
        np.testing.assert_array_equal(m.biases, first[2])
        assert m.log_likelihood() == first[3]


def test_initial_state_is_the_references_draw_at_every_size(golden):
    """regression.py:86-92 draws a, W, b from the prior with NumPy's global generator; fixture G13 holds what the REFERENCE drew under
    np.random.seed(1234) for N = 7 and N = 300.  Same seed -> the same initial chain, bit for
    bit, at the benchmark sizes too (round 2 switched to a vectorised draw above N = 256, which was a different stream)."""
    from pyglm_amd.regression import SparseBernoulliRegression
    for tag, (N, B, kw) in {"small": (7, 3, dict(rho=0.6, S_w=2.0, mu_w=0.3, mu_b=-1.0, S_b=0.5)),
                            "large": (300, 2, dict(rho=0.5, S_w=10.0, mu_b=-2.0))}.items():
        np.random.seed(1234)
        r = SparseBernoulliRegression(N, B, **kw)
        np.testing.assert_array_equal(r.a, golden["I_%s_a" % tag])
        np.testing.assert_array_equal(r.W, golden["I_%s_W" % tag])
        np.testing.assert_array_equal(np.asarray(r.b), golden["I_%s_b" % tag])


def test_shard_override_sweeps_its_rows_only():
    """`shard=(n0, n1)` (bench.py --neurons: k neurons of one rank's shard of a config too large for the GPUs at hand) sweeps exactly those
    regressions -- with the same random inputs as a full model, so their new rows equal the full model's -- and leaves every other row of
    (A, W, b) as it was; nothing is exchanged."""
    from pyglm_amd.models import SparseBernoulliGLM
    from tests._oracle_engine import OracleEngine
    np.random.seed(11)
    N, B, T = 6, 2, 250
    Y = (np.random.rand(T, N) < 0.2).astype(float)
    kw = dict(B=B, regression_kwargs=dict(S_w=2.0, mu_b=-1.0), seed=5, engine_factory=OracleEngine)
    np.random.seed(3)
    full = SparseBernoulliGLM(N, **kw)
    np.random.seed(3)
    part = SparseBernoulliGLM(N, shard=(2, 5), **kw)
    assert (part.n0, part.n1) == (2, 5)
    for m in (full, part):
        m.add_data(Y)
    A0, W0, b0 = part.adjacency.copy(), part.weights.copy(), part.biases.copy()
    np.testing.assert_array_equal(A0, full.adjacency)
    full.resample_regressions()
    part.resample_regressions()
    np.testing.assert_array_equal(part.adjacency[2:5], full.adjacency[2:5])
    np.testing.assert_allclose(part.weights[2:5], full.weights[2:5], rtol=0, atol=0)
    for lo, hi in ((0, 2), (5, N)):
        np.testing.assert_array_equal(part.adjacency[lo:hi], A0[lo:hi])
        np.testing.assert_array_equal(part.weights[lo:hi], W0[lo:hi])
        np.testing.assert_array_equal(part.biases[lo:hi], b0[lo:hi])
    assert np.isfinite(part.log_likelihood())


def test_integer_gram_planning_arithmetic():
    """host arithmetic of the integer Gram's plan (no GPU): item-times of a product launch per group size, and the cost model that decides
    gram='auto' below 1024 columns from the rates measured in profiles/archive/r04_small_D_crossover.md"""
    import types
    from pyglm_amd.engine import GibbsEngine
    r = GibbsEngine._i8_rounds
    assert r(8, 136, 13) == 55.25            # BASELINE configs[2]: one neuron per XCD, 12 x 136 whole items + 4 x 136 quarter items on 32 CUs
    assert r(8, 3, 13) == 2.0 and r(16, 3, 13) == 3.0 and r(32, 3, 13) == 5.0 and r(64, 3, 13) == 9.75     # configs[1]: per neuron 0.25 -> 0.152
    assert r(5, 3, 13) == 1.0 and r(1, 5356, 13) == 272.0                                                   # flat lists (ragged groups; configs[4])

    def pays(N, B, T, nloc=None):
        eng = types.SimpleNamespace(D=N * B, N=N, nb=None, nloc=nloc or N, I8_SMALL_T=GibbsEngine.I8_SMALL_T, I8_GROUPS=GibbsEngine.I8_GROUPS,
                                    _i8_rounds=GibbsEngine._i8_rounds)
        return GibbsEngine._i8_pays(eng, T)
    assert pays(128, 5, 50000) and pays(130, 5, 50000) and pays(64, 5, 50000) and pays(180, 5, 50000)
    assert not pays(100, 5, 50000)           # D = 500 pads to 640: a tie with the fp64 kernel (33.6 / 33.7 ms per sweep measured)
    assert pays(128, 5, 16384) and not pays(128, 5, 16383) and not pays(128, 5, 9000)            # short data sets keep the fp64 kernel
    assert pays(128, 5, 50000, nloc=2) and pays(128, 5, 50000, nloc=16)   # the choice follows the whole model, not the shard: 1 GPU and 8 take the same path
    assert not pays(2, 320, 50000)           # a model of two neurons does not fill a launch
    assert not pays(32, 5, 50000)            # D = 160: one padded tile against three small fp64 tiles


def test_chain_state_lives_in_three_model_arrays_and_the_regressions_see_views():
    """round 5: a population model keeps (A, W, b) as three arrays; `regressions[n].a / .W / .b` are views of row n (no 3 N copies per sweep,
    pyglm/models.py:54-64 read-backs unchanged).  The reference's idioms keep working: in-place edits through a regression reach the model,
    assignment copies values into the row, a regression swapped in from outside is adopted, get_state / set_state restore the link."""
    from pyglm_amd.models import SparseBernoulliGLM
    from pyglm_amd.regression import SparseBernoulliRegression
    from tests._oracle_engine import OracleEngine
    np.random.seed(6)
    N, B, T = 5, 2, 200
    m = SparseBernoulliGLM(N, B=B, regression_kwargs=dict(S_w=2.0, mu_b=-1.0), seed=3, engine_factory=OracleEngine)
    A, W, b = m._adopt_state()
    r2 = m.regressions[2]
    assert r2.a.base is A and r2.W.base is W and r2.b.shape == (1,) and r2.a.dtype == bool and r2.W.shape == (N, B)
    r2.a[4] = not r2.a[4]                              # in place through the view
    assert m.adjacency[2, 4] == r2.a[4]
    r2.W = np.full((N, B), 7.0)                        # assignment copies the values into the model's row
    r2.b = np.array([0.25])
    assert np.all(m.weights[2] == 7.0) and m.biases[2] == 0.25 and r2.W.base is W
    w = m.weights
    w[:] = 0.0                                         # the read-backs are copies, as np.array([...]) is in the reference
    assert np.all(m.weights[2] == 7.0)
    # a stand-alone regression owns its arrays until a model adopts it
    other = SparseBernoulliRegression(N, B, S_w=2.0, mu_b=-1.0)
    oa, oW = other.a.copy(), other.W.copy()
    assert other._store is None
    m.regressions[1] = other
    np.testing.assert_array_equal(m.adjacency[1], oa)
    np.testing.assert_array_equal(m.weights[1], oW)
    assert other._store is not None and other.a.base is m._adopt_state()[0]
    # sweeps write through the arrays; a saved state restores both the values and the link
    m.add_data((np.random.rand(T, N) < 0.2).astype(float))
    st = m.get_state()
    before = (m.adjacency, m.weights, m.biases)
    m.resample_model()
    assert not np.array_equal(m.weights, before[1])
    m.set_state(st)
    for got, want in zip((m.adjacency, m.weights, m.biases), before):
        np.testing.assert_array_equal(got, want)
    assert all(r.a.base is m._adopt_state()[0] for r in m.regressions)
    m.regressions[0].a[:] = True
    assert m.adjacency[0].all()


def test_copies_of_an_adopted_regression_are_detached():
    """copy.copy / copy.deepcopy / pickle of a regression that a model has adopted give a regression that owns its (a, W, b) (the
    row's values): assigning to the copy does not overwrite the model's row, and a pickled regression does not carry the (N, N, B) store."""
    import copy
    import pickle
    from pyglm_amd.models import SparseBernoulliGLM
    from tests._oracle_engine import OracleEngine
    np.random.seed(8)
    N, B = 40, 3
    m = SparseBernoulliGLM(N, B=B, regression_kwargs=dict(S_w=2.0, mu_b=-1.0), seed=3, engine_factory=OracleEngine)
    r = m.regressions[5]
    W5 = m.weights[5].copy()
    for c in (copy.copy(r), copy.deepcopy(r), pickle.loads(pickle.dumps(r))):
        assert c._store is None and c.W.shape == (N, B) and c.b.shape == (1,)
        np.testing.assert_array_equal(c.W, W5)
        np.testing.assert_array_equal(c.a, m.adjacency[5])
        c.W = np.full((N, B), 9.0)
        c.a[:] = True
        np.testing.assert_array_equal(m.weights[5], W5)
        assert c.rho.shape == (N,) and c.S_w.shape == (N, B, B)
    assert len(pickle.dumps(r)) < 0.2 * len(pickle.dumps(m._adopt_state()[1]))      # one row + hyper-parameters, not the model's arrays
    assert r._store is not None and r.W.base is m._adopt_state()[1]                  # the original stays a view


def test_counter_files_are_tied_to_the_kernel_sources():
    """committed hardware-counter summaries record the hash of the kernel sources they were taken on (pyglm_amd._lib.source_hash); bench.py
    quotes them only on a match.  The hash is stable, 16 hex digits, and moves with the sources."""
    import json
    import os
    import shutil
    import tempfile
    from pyglm_amd import _lib
    h = _lib.source_hash()
    assert len(h) == 16 and int(h, 16) >= 0 and h == _lib.source_hash()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pmc = json.load(open(os.path.join(root, "profiles", "gram_pmc.json")))
    assert "source_hash" in pmc["int8"], "profiles/gram_pmc.json: re-take with tools/profile_final.sh (records the source hash)"
    # a copy of the package with one kernel source touched hashes differently
    tmp = tempfile.mkdtemp()
    try:
        shutil.copytree(os.path.join(root, "pyglm_amd"), os.path.join(tmp, "pyglm_amd"), ignore=shutil.ignore_patterns("lib", "_build", "__pycache__"))
        shutil.copytree(os.path.join(root, "include"), os.path.join(tmp, "include"))
        with open(os.path.join(tmp, "pyglm_amd", "csrc", "pgl_chol.hip"), "a") as f:
            f.write("// touched\n")
        here, _lib._HERE = _lib._HERE, os.path.join(tmp, "pyglm_amd")
        try:
            assert _lib.source_hash() != h
        finally:
            _lib._HERE = here
    finally:
        shutil.rmtree(tmp)
