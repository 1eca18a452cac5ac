"""Per-neuron regression objects with the reference's interface (pyglm/regression.py), host state only.

A regression owns its NumPy state (a, W, b) and hyper-parameters exactly like the reference's objects, so user code
that pokes `model.regressions[n].a[n] = True` or sets `reg.S_w = ...` keeps working.  All arithmetic of the Gibbs
path runs on the GPU: a population model batches its regressions through `pyglm_amd.engine.GibbsEngine`; a
stand-alone regression (examples/bernoulli_regression.py style) builds a one-neuron engine on demand.
"""
import ctypes

import numpy as np
import numpy.random as npr

from .utils.utils import logistic, expand_scalar, expand_cov, fingerprint


class _BlockRows(object):
    """N copies of one block, with row n replaced by another: what a network prior pushes into a regression (networks.py:96-130)"""

    def __init__(self, off, own, n):
        self.off, self.own, self.n = off, own, n

    def expand(self, N):
        out = np.empty((N,) + self.off.shape)
        out[:] = self.off
        if self.own is not None:
            out[self.n] = self.own
        return out


class _SparseScalarRegressionBase(object):
    """spike-and-slab regression y_t ~ sum_n a_n w_n.x_{t,n} + b  (reference regression.py:40-92)."""
    _obs = None

    def __init__(self, N, B, rho=0.5, mu_w=0.0, S_w=1.0, mu_b=0.0, S_b=1.0):
        self.N, self.B = N, B
        self.rho, self.mu_w, self.mu_b, self.S_w, self.S_b = rho, mu_w, mu_b, S_w, S_b
        # initial state: a draw from the prior, the reference's own NumPy stream (:86-92) at every N: `npr.rand(N)`, then one
        # `npr.multivariate_normal(mu_w[n], S_w[n])` per presynaptic neuron -- which is standard_normal(B) pushed through the SVD factor of
        # the covariance (numpy/random/mtrand: x . (sqrt(s)[:, None] v) + mean).  The normals of all N calls are one standard_normal((N, B))
        # (the legacy generator fills an array by the same successive draws), the factor is formed once per DISTINCT covariance instead of
        # a million times at N = 1024, and the per-row product is the same 1-D np.dot: same seed -> the same initial chain as the
        # reference, bit for bit (fixture G13, tests/test_host_logic.py).
        self.a = npr.rand(N) < self._rho
        self.W = np.zeros((N, B))
        Z = npr.standard_normal((N, B))
        factors = {}
        S_all, mu_all = self._get_rows("_S_w"), self._get_rows("_mu_w")
        for n in np.nonzero(self.a)[0]:
            key = S_all[n].tobytes()
            M = factors.get(key)
            if M is None:
                _, sv, v = np.linalg.svd(S_all[n])
                M = factors[key] = np.sqrt(sv)[:, None] * v
            x = np.dot(Z[n], M)
            x += mu_all[n]
            self.W[n] = x
        self.b = npr.multivariate_normal(self._mu_b, self._S_b)
        self._engine_cache = None
        self._lik_engine_cache = None

    # ---- chain state (a, W, b).  A stand-alone regression owns its three arrays, as in the reference.  A population model ADOPTS the state of
    # its regressions into three arrays of its own -- A (N, N) bool, W (N, N, B), b (N, 1) -- so that a sweep writes back with three array
    # assignments instead of 3 N copies in a Python loop (40 ms per sweep and rank at N = 1024, and it does not shrink with the number of
    # GPUs); `reg.a`, `reg.W`, `reg.b` are then VIEWS of row n of those arrays: `reg.a[m] = True` edits the model's state in place as it did
    # before, and `reg.W = x` copies x's values into the row.
    _store = None                   # (A, W, b, row) once adopted by a population model

    def _state_get(self, k, own):
        st = self._store
        return self.__dict__[own] if st is None else st[k][st[3]]

    def _state_set(self, k, own, value):
        st = self._store
        if st is None:
            self.__dict__[own] = value
        else:
            st[k][st[3]] = value

    a = property(lambda self: self._state_get(0, "_a"), lambda self, v: self._state_set(0, "_a", v))
    W = property(lambda self: self._state_get(1, "_W"), lambda self, v: self._state_set(1, "_W", v))
    b = property(lambda self: self._state_get(2, "_b"), lambda self, v: self._state_set(2, "_b", v))

    def _adopt(self, A, W, b, row):
        """move this regression's (a, W, b) into row `row` of a population model's state arrays"""
        if self._store is not None and self._store[0] is A and self._store[3] == row:
            return
        a0, W0, b0 = self.a, self.W, self.b
        A[row], W[row], b[row] = a0, W0, np.asarray(b0).reshape(1)
        self._store = (A, W, b, row)
        self.__dict__.pop("_a", None), self.__dict__.pop("_W", None), self.__dict__.pop("_b", None)

    def __getstate__(self):
        """copy.copy / copy.deepcopy / pickle of a regression DETACH it from its population model: the copy owns its (a, W, b) again (the values
        of its row) instead of aliasing -- or carrying along -- the model's (N, N, B) state arrays; device-side caches are not state.  A
        model that deep-copies its regressions (get_state / set_state) adopts the copies into fresh arrays."""
        d = dict(self.__dict__)
        st = d.pop("_store", None)
        if st is not None:
            d["_a"], d["_W"], d["_b"] = st[0][st[3]].copy(), st[1][st[3]].copy(), st[2][st[3]].copy()
        d["_engine_cache"] = d["_lik_engine_cache"] = None
        return d

    # hyper-parameter setters broadcast scalars (:95-136).  The population model caches the natural-parameter terms of its regressions;
    # a version counter says whether they are still current.  It is bumped by every assignment AND by every read through the public
    # properties: a getter hands out the live array, and the reference's users edit those in place (`reg.rho[m] = 0.9`) -- after such a
    # read the terms are recomputed from the arrays, as the reference does every sweep (:266).  Internal code reads `_hyper()`.
    _hyp_version = 0
    _handed_out = frozenset()       # hyper-parameter arrays somebody outside may still hold: their cached terms are never trusted

    def _set(self, name, value, given=None):
        setattr(self, name, value)
        self._hyp_version = self._hyp_version + 1
        # an array the caller passed in and that was stored as it is (no copy) stays editable in the caller's hands
        if isinstance(given, np.ndarray) and isinstance(value, np.ndarray) and np.may_share_memory(value, given):
            self._handed_out = self._handed_out | {name}
        else:
            self._handed_out = self._handed_out - {name}

    def _get_rows(self, name):
        v = getattr(self, name)
        if isinstance(v, _BlockRows):          # pushed by a network prior as (shared block, own block): expanded on first read
            v = v.expand(self.N)
            setattr(self, name, v)
        return v

    def _touch(self, name, rows=False):
        # a getter hands out the LIVE array: whoever keeps it can edit it at any later time (`rho = reg.rho; ...; rho[m] = 0.9` after
        # further sweeps), which no version counter sees.  From here on the terms of this regression are recomputed from the arrays every
        # sweep, as the reference does (:266), until the array is replaced by an assignment or a network push.
        self._hyp_version = self._hyp_version + 1
        self._handed_out = self._handed_out | {name}
        return self._get_rows(name) if rows else getattr(self, name)

    def _hyper(self):
        """(rho, S_w, mu_w, S_b, mu_b) without marking the cached terms stale"""
        return self._rho, self._get_rows("_S_w"), self._get_rows("_mu_w"), self._S_b, self._mu_b

    rho = property(lambda self: self._touch("_rho"), lambda self, v: self._set("_rho", expand_scalar(v, (self.N,)), v))
    mu_w = property(lambda self: self._touch("_mu_w", True), lambda self, v: self._set("_mu_w", expand_scalar(v, (self.N, self.B)), v))
    mu_b = property(lambda self: self._touch("_mu_b"), lambda self, v: self._set("_mu_b", expand_scalar(v, (1,)), v))
    S_w = property(lambda self: self._touch("_S_w", True), lambda self, v: self._set("_S_w", expand_cov(v, (self.N, self.B, self.B)), v))

    def _push_block_prior(self, mu_off, S_off, mu_self, S_self, n, rho_row):
        """the hyper-parameter push of models.py:233-236 for a prior with one shared weight block and (optionally) one for the
        self-connection n: same values as assigning the expanded (N, B) / (N, B, B) arrays, which are only built if somebody reads them"""
        self._mu_w = _BlockRows(mu_off, mu_self, n)
        self._S_w = _BlockRows(S_off, S_self, n)
        self._handed_out = self._handed_out - {"_mu_w", "_S_w"}
        self._set("_rho", np.array(expand_scalar(rho_row, (self.N,))))          # (a copy: the network keeps its own array)

    @property
    def S_b(self):
        return self._touch("_S_b")

    @S_b.setter
    def S_b(self, value):
        assert np.isscalar(value)
        self._set("_S_b", expand_cov(value, (1, 1)))

    @property
    def natural_params(self):
        """(:138-151)"""
        rho, S_w, mu_w, S_b, mu_b = self._hyper()
        J_w = np.linalg.inv(S_w)
        h_w = np.einsum("nij,nj->ni", J_w, mu_w)
        J_b = np.linalg.inv(S_b)
        return J_w, h_w, J_b, J_b.dot(mu_b)

    @property
    def deterministic_sparsity(self):
        return bool(np.all((self._rho < 1e-6) | (self._rho > 1 - 1e-6)))

    def _flatten_X(self, X):
        X = np.asarray(X)
        if X.ndim == 3:
            X = X.reshape(-1, self.N * self.B)
        elif X.ndim != 2:
            raise Exception
        assert X.shape[1] == self.N * self.B
        return X

    def extract_data(self, data):
        assert isinstance(data, tuple) and len(data) == 2
        X, y = data
        T = X.shape[0]
        assert y.shape == (T, 1) or y.shape == (T,)
        return self._flatten_X(X), y

    # ---- GPU-backed stand-alone operations
    def _engine(self, datas, likelihood_only=False):
        """one-neuron engine holding `datas` = [(X, y), ...] on the GPU.  Cached by CONTENT (utils.fingerprint of every array): a second
        call with equal data reuses the device copy; different data of the same shape, or data edited in place, is uploaded again.
        Activation / mean / log-likelihood calls use a likelihood-only engine keyed by X alone (no sweep buffers, y irrelevant)."""
        from .engine import GibbsEngine
        flat = [self.extract_data((X, y)) for X, y in datas]
        key = (bool(likelihood_only),) + tuple((fingerprint(X), None if likelihood_only else fingerprint(y)) for X, y in flat)
        slot = "_lik_engine_cache" if likelihood_only else "_engine_cache"
        cache = getattr(self, slot, None)
        if cache is None or cache[0] != key:
            setattr(self, slot, None)              # release the previous engine's device memory before allocating the next
            eng = GibbsEngine(self.N, self.B, 0, 1, obs=self._obs, xi=getattr(self, "xi", 1.0), batch=1, likelihood_only=likelihood_only)
            for X, y in flat:
                Y = np.zeros((X.shape[0], self.N))
                if not likelihood_only:
                    Y[:, 0] = np.asarray(y, dtype=float).ravel()
                eng.add_data(Y, X=np.asarray(X, dtype=float).reshape(-1, self.N, self.B))
            cache = (key, eng)
            setattr(self, slot, cache)
        return cache[1]

    def activation(self, X):
        """psi = X.vec(a*W) + b  (:195-201)"""
        X = self._flatten_X(X)
        eng = self._engine([(X, np.zeros(X.shape[0]))], likelihood_only=True)
        return eng.psi(self.a[None], self.W[None], self.b)[:, 0]

    def _before_sweep(self, eng):
        pass

    def _after_sweep(self, eng, seed, sweep):
        pass

    def resample(self, datas, seed=None, sweep=0):
        """One Gibbs update of (a, W, b) given datasets [(X, y), ...]  (:265-280)."""
        from .engine import make_draws, prior_terms
        eng = self._engine(datas)
        seed = int(npr.randint(2 ** 31)) if seed is None else seed
        rho_, S_w_, mu_w_, S_b_, mu_b_ = self._hyper()
        Jw, hw, Jb, hb, c0 = prior_terms(S_w_[None], mu_w_[None], S_b_.reshape(1), mu_b_.reshape(1))
        perm, u, z = make_draws(seed, sweep, [0], self.N, self.N * self.B)
        self._before_sweep(eng)
        a, W, b, _ = eng.sweep(self.a[None], self.W[None], self.b, rho_[None], Jw, hw, Jb, hb, c0, perm, u, z, seed, sweep)
        self.a, self.W, self.b = a[0], W[0], b.reshape(1)
        self._after_sweep(eng, seed, sweep)


class SparseGaussianRegression(_SparseScalarRegressionBase):
    """y_t ~ N(psi_t, eta), eta ~ InvGamma(a_0, b_0)  (reference regression.py:380-446).  omega = 1/eta is constant in t, so on the GPU
    the likelihood precision is (1/eta) X'X with X'X formed once per dataset (engine.add_data) instead of one Gram per neuron."""
    _obs = "gaussian"

    def __init__(self, N, B, a_0=2.0, b_0=2.0, eta=None, **kwargs):
        super(SparseGaussianRegression, self).__init__(N, B, **kwargs)
        assert np.isscalar(a_0) and a_0 > 0
        assert np.isscalar(b_0) and a_0 > 0
        self.a_0, self.b_0 = a_0, b_0
        if eta is not None:
            assert np.isscalar(eta) and eta > 0
            self.eta = eta
        else:
            self.eta = 1.0 / npr.gamma(self.a_0, 1.0 / self.b_0)      # sample_invgamma(a_0, b_0)  (:396-398)

    def mean(self, X):
        return self.activation(X)

    def log_likelihood(self, x):
        """per-bin log N(y_t | mean_t, eta)  (:399-403)"""
        X, y = self.extract_data(x)
        return -0.5 * np.log(2 * np.pi * self.eta) - 0.5 * (y - self.mean(X)) ** 2 / self.eta

    def rvs(self, size=[], X=None, psi=None):
        if psi is None:
            if X is None:
                assert isinstance(size, int)
                X = npr.randn(size, self.N * self.B)
            psi = self.mean(self._flatten_X(X))
        return psi + np.sqrt(self.eta) * npr.randn(*psi.shape)

    def omega(self, X, y):
        return 1.0 / self.eta * np.ones(X.shape[0])

    def kappa(self, X, y):
        return y / self.eta

    def eta_posterior(self, T_total, sse):
        """(alpha, beta) of the conditional of eta (:433-445; the reference adds the full sum of squares, not half of it)"""
        return self.a_0 + T_total / 2.0, self.b_0 + sse

    def _before_sweep(self, eng):
        eng.set_noise([self.eta])

    def _after_sweep(self, eng, seed, sweep):
        from .engine import make_gamma_draws
        T_total = sum(ds.T for ds in eng.datasets)
        alpha, beta = self.eta_posterior(T_total, float(eng.sse(self.a[None], self.W[None], self.b)[0]))
        self.eta = 1.0 / (make_gamma_draws(seed, sweep, [0], alpha)[0] * (1.0 / beta))


class GaussianRegression(SparseGaussianRegression):
    """dense weights: rho = 1 (:448-456)"""

    def __init__(self, N, B, **kwargs):
        kwargs["rho"] = np.ones(N)
        super(GaussianRegression, self).__init__(N, B, **kwargs)


class _SparsePGRegressionBase(_SparseScalarRegressionBase):
    """count observations through Polya-gamma augmentation (reference regression.py:459-511); the observation model is
    given by the hooks a_func/b_func/c_func (:479-489)."""

    def log_likelihood(self, x):
        """per-bin log p(y_t | psi_t) = log c + a psi - b log(1+e^psi)  (:491-494).  The per-bin vector is formed on the
        host from the GPU activation (the population model uses the fused device reduction instead)."""
        X, y = self.extract_data(x)
        psi = self.activation(X)
        return np.log(self.c_func(y)) + self.a_func(y) * psi - self.b_func(y) * np.log1p(np.exp(psi))

    def kappa(self, X, y):
        return self.a_func(y) - self.b_func(y) / 2.0

    def omega(self, X, y, seed=None, sweep=0):
        """omega_t ~ PG(b(y_t), psi_t)  (:496-508): the activation on the GPU, then one draw per time bin from the device sampler
        (pgl_pg_draw, which stands where the reference calls pypolyagamma's pgdrawvpar).  Returns a NumPy vector shaped like y.
        The reference draws from per-thread samplers seeded once from NumPy's global stream (:476); here every draw has its own
        counter-based stream keyed by (seed, sweep, neuron, time bin): seed=None takes a fresh seed from NumPy's global stream (so
        np.random.seed(k) makes the call reproducible, and successive calls differ, as in the reference), an explicit (seed, sweep)
        replays the draws a model sweep of that key makes for this neuron (the regression's row in its model, 0 when stand-alone)."""
        import torch
        from ._lib import call, ptr
        X = self._flatten_X(X)
        y = np.asarray(y)
        assert y.shape == (X.shape[0], 1) or y.shape == (X.shape[0],)
        psi = self.activation(X)
        bshape = np.ascontiguousarray(np.broadcast_to(np.asarray(self.b_func(y.reshape(-1)), dtype=np.float64), psi.shape))
        seed = int(npr.randint(2 ** 31)) if seed is None else int(seed)
        neuron = 0 if self._store is None else int(self._store[3])
        dev = self._lik_engine_cache[1].dev
        with torch.cuda.device(dev):
            zd, bd = torch.from_numpy(np.array(psi, dtype=np.float64)).to(dev), torch.from_numpy(np.array(bshape)).to(dev)
            out = torch.empty_like(zd)
            call("pgl_pg_draw", ptr(bd), ptr(zd), ptr(out), psi.size, seed, (int(sweep) << 32) | neuron, 0,
                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            return out.cpu().numpy().reshape(y.shape)


class SparseBernoulliRegression(_SparsePGRegressionBase):
    """a=y, b=1, c=1 (:514-522)"""
    _obs = "bernoulli"

    def a_func(self, data):
        return data

    def b_func(self, data):
        return np.ones_like(data, dtype=float)

    def c_func(self, data):
        return 1.0

    def mean(self, X):
        return logistic(self.activation(X))

    def rvs(self, X=None, size=[], psi=None):
        if psi is None:
            if X is None:
                assert isinstance(size, int)
                X = npr.randn(size, self.N * self.B)
            p = self.mean(self._flatten_X(X))
        else:
            p = logistic(psi)
        return npr.rand(*p.shape) < p


class BernoulliRegression(SparseBernoulliRegression):
    """dense weights: rho = 1 (:544-552)"""

    def __init__(self, N, B, **kwargs):
        kwargs["rho"] = np.ones(N)
        super(BernoulliRegression, self).__init__(N, B, **kwargs)


class SparseNegativeBinomialRegression(_SparsePGRegressionBase):
    """named in the reference docstring (:463-466) but not implemented there; defined through the hooks (:479-489):
    y ~ NB(xi, sigma(psi)):  a = y, b = y + xi, c = C(y+xi-1, y)."""
    _obs = "negbin"

    def __init__(self, N, B, xi=1.0, **kwargs):
        assert xi > 0
        self.xi = float(xi)
        super(SparseNegativeBinomialRegression, self).__init__(N, B, **kwargs)

    def a_func(self, data):
        return data

    def b_func(self, data):
        return data + self.xi

    def c_func(self, data):
        from scipy.special import gammaln
        return np.exp(gammaln(data + self.xi) - gammaln(data + 1) - gammaln(self.xi))

    def mean(self, X):
        psi = self.activation(X)
        return self.xi * np.exp(psi)

    def rvs(self, X=None, size=[], psi=None):
        if psi is None:
            psi = self.activation(self._flatten_X(X))
        p = logistic(psi)
        return npr.negative_binomial(self.xi, 1 - p).astype(float)


class NegativeBinomialRegression(SparseNegativeBinomialRegression):
    def __init__(self, N, B, **kwargs):
        kwargs["rho"] = np.ones(N)
        super(NegativeBinomialRegression, self).__init__(N, B, **kwargs)
