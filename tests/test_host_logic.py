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
    assert FixedMeanSparseNetwork(N, B, mu=0.5, sigma=2.0, rho=0.2).sigma_W[0, 1, 1, 1] == 2.0
    with pytest.raises(AssertionError):
        net.resample((A.astype(float), W))
    # posterior parameters of the NIW update equal the oracle's restatement
    data = np.random.randn(40, B) + 1.0
    ours = _NIW(np.zeros(B), np.eye(B), 1.0, 4.0)
    ref = orc.NIWGaussian(np.zeros(B), np.eye(B), 1.0, 4.0, np.random.default_rng(0))
    mu_n, sig_n, k_n, nu_n = ref.posterior(data)
    draws = []
    for _ in range(3000):
        ours.resample(data)
        draws.append(ours.mu)
    np.testing.assert_allclose(np.mean(draws, 0), mu_n, atol=0.02)


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
    np.testing.assert_array_equal(m.regressions[4].S_w, m.network.sigma_W[4])
    np.testing.assert_array_equal(m.regressions[4].mu_w, m.network.mu_W[4])
    versions, (rho, prior, _, Jb, hb, _) = m._hyper_cache
    assert prior.Jw_u.shape[0] <= 3 and prior.label.shape == (N, N)      # a handful of distinct blocks, never expanded
    Jw, hw, c0 = prior.dense()
    regs = m.regressions
    want = prior_terms(np.array([r.S_w for r in regs]), np.array([r.mu_w for r in regs]), np.array([r.S_b[0, 0] for r in regs]),
                       np.array([r.mu_b[0] for r in regs]))
    for got, w in zip((Jw, hw, Jb, hb, c0), want):
        np.testing.assert_allclose(got, w, rtol=1e-13, atol=1e-15)
    np.testing.assert_array_equal(rho, np.array([r.rho for r in regs]))
    assert versions == tuple(r._hyp_version for r in regs)
    regs[2].S_w = 5.0
    assert versions != tuple(r._hyp_version for r in regs)      # stale cache is detected


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
