"""
oracle/pyglm_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain NumPy/SciPy (fp64, CPU) restatement of the reference algorithm on the Gibbs hot path of
slinderman/pyglm, written function-by-function with the reference file:line each one follows.
It exists so that tests can compare the HIP path with "what the reference computes" on the GPU
box, where /root/reference does not exist.  It is pinned against golden vectors captured from
the reference's own NumPy code (tests/golden/make_fixtures.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (pyglm_amd/) never does; it fails loudly when its HIP library is missing.

Random numbers are never drawn here implicitly: every stochastic function takes the uniforms /
normals / permutation it should consume as arguments, in the order the reference consumes them
from `numpy.random` (permutation(N) -> N uniforms -> randn(sum(a)*B+1)).

Two third-party pieces the reference calls are restated from their published definitions because
their sources are absent from /root/reference (setup.py:13, un-pinned) -- PARITY UNPINNED there:
  * pybasicbayes.util.stats.sample_discrete_from_log  -> `discrete_from_log`
  * pybasicbayes.util.stats.sample_gaussian(J=, h=)   -> `gaussian_info_draw`
  * pybasicbayes.distributions.Gaussian (NIW)         -> `NIWGaussian`
  * pypolyagamma.pgdrawvpar                           -> oracle/pg_oracle.c (`pg_draw`)
"""
import ctypes
import os

import numpy as np
import scipy.linalg as sla

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------- small utils
# reference: pyglm/utils/utils.py:3-27


def logistic(x):
    return 1.0 / (1.0 + np.exp(-x))


def expand_scalar(x, shp):
    """utils.py:7-12 -- a Python/NumPy scalar is broadcast, an array must already have `shp`."""
    if np.isscalar(x):
        return x * np.ones(shp)
    x = np.asarray(x, dtype=float)
    assert x.shape == tuple(shp)
    return x


def expand_cov(c, shp):
    """utils.py:15-27 -- scalar c -> c*I tiled over the leading dims."""
    shp = tuple(shp)
    assert len(shp) >= 2 and shp[-2] == shp[-1]
    if np.isscalar(c):
        d = shp[-1]
        return np.tile(c * np.eye(d), shp[:-2] + (1, 1))
    c = np.asarray(c, dtype=float)
    assert c.shape == shp
    return c


# --------------------------------------------------------------------------- basis / design matrix
# reference: pyglm/utils/basis.py


def cosine_basis(B, L=100, orth=False, norm=True, n_eye=0, a=1.0 / 120, b=0.5):
    """basis.py:61-106 raised-cosine bumps on a log-warped time axis, columns normalised to sum L."""
    n_cos = B - n_eye
    assert n_cos >= 0 and n_eye >= 0
    out = np.zeros((L, B))
    out[:n_eye, :n_eye] = np.eye(n_eye)
    u = np.log(a * np.arange(L) + b)
    centres = u[np.floor(np.linspace(n_eye, L / 2.0, n_cos)).astype(int)]
    width = centres / 2 if len(centres) == 1 else (centres[-1] - centres[0]) / (n_cos - 1)
    for i in range(n_cos):
        arg = np.clip((u - centres[i]) * np.pi / width / 2.0, -np.pi, np.pi)
        out[:, n_eye + i] = (np.cos(arg) + 1) / 2.0
    if orth:
        out = sla.orth(out)
    elif norm:
        if np.any(out < 0):
            raise Exception("We can only normalize nonnegative impulse responses!")
        out = out / np.tile(np.sum(out, axis=0), [L, 1]) / (1.0 / L)
    return out


def convolve_with_basis(S, basis, method="direct"):
    """basis.py:5-34.  F[t,n,b] = sum_{l>=0} basis[l,b] * S[t-1-l, n]  (a zero row is prepended to
    the basis so the filter is strictly causal, :18), clipped at 0 when both are non-negative (:30-32).
    `direct` evaluates the lagged sum exactly; `fft` uses scipy.signal.fftconvolve as the reference does
    (they agree to ~1e-16 absolute)."""
    S = np.asarray(S, dtype=float)
    T, N = S.shape
    R, B = basis.shape
    F = np.zeros((T, N, B))
    if method == "fft":
        import scipy.signal as sig
        bz = np.vstack((np.zeros((1, B)), basis))
        for b in range(B):
            F[:, :, b] = sig.fftconvolve(S, bz[:, b].reshape(R + 1, 1), "full")[:T, :]
    else:
        for l in range(min(R, T - 1)):
            F[l + 1:, :, :] += S[: T - 1 - l, :, None] * basis[l][None, None, :]
    if np.amin(basis) >= 0 and np.amin(S) >= 0:
        np.clip(F, 0, np.inf, out=F)
    return F


# --------------------------------------------------------------------------- pybasicbayes restatements


def discrete_from_log(lps, u):
    """pybasicbayes.util.stats.sample_discrete_from_log (call site regression.py:315), published form:
    cum = cumsum(exp(lps - max)); r = u * cum[-1]; return #{k : r > cum[k]}."""
    lps = np.asarray(lps, dtype=float)
    with np.errstate(invalid="ignore"):
        cum = np.cumsum(np.exp(lps - lps.max()))
        r = u * cum[-1]
        return int(np.sum(r > cum))       # NaN anywhere -> every comparison False -> 0


def gaussian_info_draw(J, h, z):
    """pybasicbayes.util.stats.sample_gaussian(J=J, h=h) (call site regression.py:334), published form:
    L = chol(J) (lower); x = solve(L^T, z) + J^{-1} h."""
    L = np.linalg.cholesky(J)
    mean = sla.cho_solve((L, True), h)
    return sla.solve_triangular(L, z, lower=True, trans="T") + mean


# --------------------------------------------------------------------------- PG draw (ctypes -> pg_oracle.c)
_pg_lib = None


def _load_pg():
    global _pg_lib
    if _pg_lib is None:
        path = os.path.join(_HERE, "_build", "libpg_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle PG library not built: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.oracle_pg_draw.restype = ctypes.c_int
        lib.oracle_pg_draw.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        lib.oracle_philox_stream.restype = None
        lib.oracle_philox_stream.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                                             ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t]
        lib.oracle_philox4x32_10.restype = None
        lib.oracle_philox4x32_10.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_num_threads.restype = ctypes.c_int
        _pg_lib = lib
    return _pg_lib


def stream_id(neuron, sweep):
    """stream = (global neuron index, sweep) packed lo/hi -- see pg_oracle.c header."""
    return (int(sweep) << 32) | (int(neuron) & 0xFFFFFFFF)


def pg_draw(b, z, seed, stream, elem0=0):
    """omega[i] ~ PG(b[i], z[i]) on the shared Philox stream; stands where regression.py:504-507 calls
    ppg.pgdrawvpar(self.ppgs, b, psi, omega)."""
    lib = _load_pg()
    z = np.ascontiguousarray(z, dtype=np.float64).ravel()
    out = np.empty_like(z)
    if b is None:
        bp = None
    else:
        b = np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.float64), z.shape)).ravel()
        bp = b.ctypes.data
    rc = lib.oracle_pg_draw(bp, z.ctypes.data, out.ctypes.data, z.size, seed, stream, elem0)
    if rc != 0:
        raise ValueError("oracle_pg_draw: b must be non-negative")
    return out


def philox_words(seed, purpose, j, elem0, stream, n):
    lib = _load_pg()
    out = np.empty((n, 4), dtype=np.uint32)
    lib.oracle_philox_stream(seed, purpose, j, elem0, stream, out.ctypes.data, n)
    return out


def philox_block(ctr, key):
    lib = _load_pg()
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.empty(4, dtype=np.uint32)
    lib.oracle_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return o


# --------------------------------------------------------------------------- one neuron's regression


class Regression:
    """State + conditionals of ONE postsynaptic neuron: reference `_SparseScalarRegressionBase`
    (regression.py:40-378) + `_SparsePGRegressionBase` (:459-511).  `obs` selects the observation model
    through the a/b/c hooks (:479-489):  'bernoulli' (:514-526)  a=y, b=1, c=1;
    'negbin' (named only in the docstring :463-466)  a=y, b=y+xi, c=C(y+xi-1, y);
    'gaussian' = `SparseGaussianRegression` (:380-446): omega = 1/eta, kappa = y/eta, eta ~ InvGamma(a_0, b_0)."""

    def __init__(self, N, B, rho=0.5, mu_w=0.0, S_w=1.0, mu_b=0.0, S_b=1.0, obs="bernoulli", xi=1.0, a_0=2.0, b_0=2.0, eta=1.0):
        self.N, self.B = N, B
        self.obs, self.xi = obs, xi
        self.a_0, self.b_0, self.eta = a_0, b_0, eta
        self.set_hypers(rho, mu_w, S_w, mu_b, S_b)
        self.a = np.zeros(N, dtype=bool)
        self.W = np.zeros((N, B))
        self.b = np.zeros(1)

    # hyper-parameter broadcasting: regression.py:95-136
    def set_hypers(self, rho=None, mu_w=None, S_w=None, mu_b=None, S_b=None):
        N, B = self.N, self.B
        if rho is not None:
            self.rho = expand_scalar(rho, (N,))
        if mu_w is not None:
            self.mu_w = expand_scalar(mu_w, (N, B))
        if S_w is not None:
            self.S_w = expand_cov(S_w, (N, B, B))
        if mu_b is not None:
            self.mu_b = expand_scalar(mu_b, (1,))
        if S_b is not None:
            self.S_b = expand_cov(S_b, (1, 1))

    def init_from_prior(self, u_a, z_w, z_b):
        """regression.py:86-92 with the draws injected: a = u < rho; W[n] = a[n]*N(mu_w[n], S_w[n]); b ~ N(mu_b, S_b).
        (z_w (N,B), z_b (1,) standard normals mapped through the Cholesky factor -- distributionally the
        reference's npr.multivariate_normal, not its exact SVD mapping.)"""
        self.a = np.asarray(u_a) < self.rho
        for n in range(self.N):
            self.W[n] = self.a[n] * (self.mu_w[n] + np.linalg.cholesky(self.S_w[n]) @ z_w[n])
        self.b = self.mu_b + np.linalg.cholesky(self.S_b) @ np.atleast_1d(z_b)

    # ---- observation-model hooks (regression.py:479-489, 514-522)
    def a_func(self, y):
        return y

    def b_func(self, y):
        if self.obs == "bernoulli":
            return np.ones_like(y, dtype=float)
        return y + self.xi

    def log_c_func(self, y):
        if self.obs == "bernoulli":
            return 0.0
        from scipy.special import gammaln
        return gammaln(y + self.xi) - gammaln(y + 1) - gammaln(self.xi)

    def kappa(self, y):
        """regression.py:510-511; Gaussian :425-426"""
        if self.obs == "gaussian":
            return y / self.eta
        return self.a_func(y) - self.b_func(y) / 2.0

    def omega_gaussian(self, T):
        """regression.py:421-423"""
        return 1. / self.eta * np.ones(T)

    def resample_eta(self, datas, g):
        """regression.py:433-445 with sample_invgamma(alpha, beta) = 1/gamma(alpha, scale=1/beta) and the standard Gamma(alpha, 1)
        variate g injected.  (As in the reference, beta accumulates the FULL sum of squared residuals.)"""
        alpha, beta = self.a_0, self.b_0
        for X, y in datas:
            alpha += self.flat(X).shape[0] / 2.0
            beta += np.sum((np.asarray(y, dtype=float).reshape(-1) - self.mean(X)) ** 2)
        self.eta = 1.0 / (g * (1.0 / beta))
        return alpha, beta

    # ---- deterministic pieces
    def flat(self, X):
        """regression.py:173-180"""
        X = np.asarray(X)
        if X.ndim == 3:
            X = X.reshape(-1, self.N * self.B)
        assert X.ndim == 2 and X.shape[1] == self.N * self.B
        return X

    def activation(self, X):
        """regression.py:195-201  psi = X . vec(a*W) + b"""
        w = (self.a[:, None] * self.W).reshape(self.N * self.B)
        return self.flat(X).dot(w) + self.b[0]

    def mean(self, X):
        """regression.py:524-526; Gaussian :430-431"""
        if self.obs == "gaussian":
            return self.activation(X)
        return logistic(self.activation(X))

    def log_likelihood(self, X, y):
        """regression.py:491-494 (per-bin vector); Gaussian :399-403"""
        psi = self.activation(X)
        if self.obs == "gaussian":
            return -0.5 * np.log(2 * np.pi * self.eta) - 0.5 * (y - psi) ** 2 / self.eta
        return self.log_c_func(y) + self.a_func(y) * psi - self.b_func(y) * np.log1p(np.exp(psi))

    def natural_params(self):
        """regression.py:138-151"""
        J_w = np.array([np.linalg.inv(S) for S in self.S_w])
        h_w = np.einsum("nij,nj->ni", J_w, self.mu_w)
        J_b = np.linalg.inv(self.S_b)
        h_b = J_b.dot(self.mu_b)
        return J_w, h_w, J_b, h_b

    def deterministic_sparsity(self):
        """regression.py:153-155"""
        return bool(np.all((self.rho < 1e-6) | (self.rho > 1 - 1e-6)))

    def prior_stats(self):
        """regression.py:210-223  dense (D+1)x(D+1) block-diagonal prior precision + potential"""
        J_w, h_w, J_b, h_b = self.natural_params()
        J = sla.block_diag(*J_w, J_b)
        h = np.concatenate((h_w.ravel(), h_b.ravel()))
        return J, h

    def lkhd_stats(self, datas, omegas):
        """regression.py:225-262 with omega injected per dataset:
        J = [X'OX, X'O1; 1'OX, sum(o)],  h = [X'k; sum(k)], summed over datasets."""
        D = self.N * self.B
        J = np.zeros((D + 1, D + 1))
        h = np.zeros(D + 1)
        for (X, y), om in zip(datas, omegas):
            X = self.flat(X)
            y = np.asarray(y, dtype=float).reshape(-1)
            kap = self.kappa(y)
            XO = X * om[:, None]
            J[:D, :D] += XO.T.dot(X)
            xs = XO.sum(0)
            J[:D, -1] += xs
            J[-1, :D] += xs
            J[-1, -1] += om.sum()
            h[:D] += kap.dot(X)
            h[-1] += kap.sum()
        return J, h

    def active_mask(self, a=None):
        a = self.a if a is None else a
        return np.concatenate((np.repeat(a, self.B), [True])).astype(bool)

    def marginal_likelihood(self, J_prior, h_prior, J_post, h_post, a=None):
        """regression.py:343-378  log-normaliser ratio on the active sub-block, via two Choleskys."""
        m = self.active_mask(a)
        J0, h0 = J_prior[np.ix_(m, m)], h_prior[m]
        Jp, hp = J_post[np.ix_(m, m)], h_post[m]
        L0 = np.linalg.cholesky(J0)
        Lp = np.linalg.cholesky(Jp)
        ml = -np.sum(np.log(np.diag(Lp))) + np.sum(np.log(np.diag(L0)))
        ml += 0.5 * hp.dot(sla.cho_solve((Lp, True), hp))
        ml -= 0.5 * h0.dot(sla.cho_solve((L0, True), h0))
        return ml

    def collapsed_resample_a(self, J_prior, h_prior, J_post, h_post, perm, u, trace=None):
        """regression.py:282-320  sequential collapsed flips in the order `perm`, uniform u[k] for step k."""
        rho = self.rho
        ml_prev = self.marginal_likelihood(J_prior, h_prior, J_post, h_post)
        with np.errstate(divide="ignore"):
            lr, l1r = np.log(rho), np.log(1 - rho)
        for k, n in enumerate(perm):
            lps = np.zeros(2)
            v_prev = int(self.a[n])
            # literal form of :298/:307 -- with rho[n] exactly 0 or 1 the product 0*log(0) is NaN, the NaN
            # reaches sample_discrete_from_log and the draw comes out 0 (kept: "identical results").
            with np.errstate(invalid="ignore"):
                lps[v_prev] += ml_prev + (v_prev * lr[n] + (1 - v_prev) * l1r[n])
            v_new = 1 - v_prev
            self.a[n] = bool(v_new)
            ml_new = self.marginal_likelihood(J_prior, h_prior, J_post, h_post)
            with np.errstate(invalid="ignore"):
                lps[v_new] += ml_new + (v_new * lr[n] + (1 - v_new) * l1r[n])
            v = discrete_from_log(lps, u[k])
            self.a[n] = bool(v)
            if trace is not None:
                trace.append((int(n), float(lps[1] - lps[0]), int(v)))
            if v != v_prev:
                ml_prev = ml_new

    def resample_W(self, J_post, h_post, z):
        """regression.py:323-340  [W_active; b] ~ N(Jp^-1 hp, Jp^-1) using the first sum(a)*B+1 entries of z."""
        m = self.active_mask()
        k = int(m.sum())
        w = gaussian_info_draw(J_post[np.ix_(m, m)], h_post[m], np.asarray(z)[:k])
        self.W = np.zeros((self.N, self.B))
        self.W[self.a, :] = w[:-1].reshape(-1, self.B)
        self.b = w[-1].reshape(1)

    def resample(self, datas, omegas, perm, u, z, trace=None):
        """regression.py:265-280 with every random input injected."""
        J_prior, h_prior = self.prior_stats()
        J_l, h_l = self.lkhd_stats(datas, omegas)
        J_post, h_post = J_prior + J_l, h_prior + h_l
        if self.deterministic_sparsity():
            self.a = np.round(self.rho).astype(bool)
        else:
            self.collapsed_resample_a(J_prior, h_prior, J_post, h_post, perm, u, trace)
        self.resample_W(J_post, h_post, z)
        return J_post, h_post


# --------------------------------------------------------------------------- network prior (host side)


def invwishart_draw(S, nu, rng):
    """pybasicbayes.util.stats.sample_invwishart, published Bartlett form (small-nu branch)."""
    n = S.shape[0]
    chol = np.linalg.cholesky(S)
    if (nu <= 81 + n) and (nu == np.round(nu)):
        x = rng.standard_normal((int(nu), n))
    else:
        x = np.diag(np.sqrt(np.atleast_1d(rng.chisquare(nu - np.arange(n)))))
        x[np.triu_indices_from(x, 1)] = rng.standard_normal(n * (n - 1) // 2)
    R = np.linalg.qr(x, "r")
    T = sla.solve_triangular(R.T, chol.T, lower=True).T
    return T.dot(T.T)


class NIWGaussian:
    """pybasicbayes.distributions.Gaussian with NIW prior (used at networks.py:89-94, 141-149): published
    conjugate update + (Sigma ~ IW, mu ~ N(mu_n, Sigma/kappa_n)) draw."""

    def __init__(self, mu_0, sigma_0, kappa_0, nu_0, rng):
        self.mu_0, self.sigma_0, self.kappa_0, self.nu_0 = np.asarray(mu_0, float), np.asarray(sigma_0, float), kappa_0, nu_0
        self.rng = rng
        self.resample(np.zeros((0, len(self.mu_0))))

    def posterior(self, data):
        data = np.asarray(data, dtype=float).reshape(-1, len(self.mu_0))
        n = data.shape[0]
        if n == 0:
            return self.mu_0, self.sigma_0, self.kappa_0, self.nu_0
        xbar = data.mean(0)
        c = data - xbar
        scatter = c.T.dot(c)
        k0, m0 = self.kappa_0, self.mu_0
        mu_n = k0 / (k0 + n) * m0 + n / (k0 + n) * xbar
        sigma_n = self.sigma_0 + scatter + k0 * n / (k0 + n) * np.outer(xbar - m0, xbar - m0)
        return mu_n, sigma_n, k0 + n, self.nu_0 + n

    def resample(self, data):
        mu_n, sigma_n, kappa_n, nu_n = self.posterior(data)
        self.sigma = invwishart_draw(sigma_n, nu_n, self.rng)
        self.mu = self.rng.multivariate_normal(mu_n, self.sigma / kappa_n)


# --------------------------------------------------------------------------- population model


class GLM:
    """reference NonlinearAutoregressiveModel / HierarchicalNonlinearAutoregressiveModel (models.py:8-236),
    regression part only; the sweep takes its random inputs per neuron from `draws`."""

    def __init__(self, N, B, basis=None, **reg_kwargs):
        self.N = N
        self.basis = np.eye(B) if basis is None else basis
        self.B = self.basis.shape[1]
        self.regressions = [Regression(N, self.B, **reg_kwargs) for _ in range(N)]
        self.data_list = []

    @property
    def weights(self):      # models.py:54-56
        return np.array([r.W for r in self.regressions])

    @property
    def adjacency(self):    # models.py:58-60
        return np.array([r.a for r in self.regressions])

    @property
    def biases(self):       # models.py:62-64
        return np.array([r.b for r in self.regressions]).ravel()

    def add_data(self, Y, X=None):   # models.py:66-80
        assert isinstance(Y, np.ndarray) and Y.ndim == 2 and Y.shape[1] == self.N
        if X is None:
            X = convolve_with_basis(Y, self.basis)
        else:
            assert X.shape == (Y.shape[0], self.N, self.B)
        self.data_list.append((X, Y))

    def log_likelihood(self):        # models.py:82-96
        ll = 0.0
        for X, Y in self.data_list:
            for n, r in enumerate(self.regressions):
                ll += r.log_likelihood(X, Y[:, n]).sum()
        return ll

    def means(self):                 # models.py:153-163
        return [np.column_stack([r.mean(X) for r in self.regressions]) for X, _ in self.data_list]

    def omegas(self, n, seed, sweep):
        """regression.py:496-508 for neuron n on every dataset (element index continues across datasets)."""
        r = self.regressions[n]
        if r.obs == "gaussian":
            return [r.omega_gaussian(X.shape[0]) for X, _ in self.data_list]
        out, off = [], 0
        for X, Y in self.data_list:
            psi = r.activation(X)
            out.append(pg_draw(r.b_func(Y[:, n].astype(float)), psi, seed, stream_id(n, sweep), off))
            off += X.shape[0]
        return out

    def resample_regressions(self, seed, sweep, perms, us, zs, gs=None):   # models.py:169-171
        for n, r in enumerate(self.regressions):
            datas = [(X, Y[:, n]) for X, Y in self.data_list]
            r.resample(datas, self.omegas(n, seed, sweep), perms[n], us[n], zs[n])
            if r.obs == "gaussian":            # SparseGaussianRegression.resample, regression.py:427-429
                r.resample_eta(datas, gs[n])
