"""TEST-ONLY stand-in for pyglm_amd.engine.GibbsEngine backed by the oracle (CPU).  It lets the host-side logic of the
population model -- sharding, all_gather of state, all_reduce of the log-likelihood, identical network draws on every
rank -- be exercised with the gloo backend on machines without a GPU.  Never imported by the product."""
import numpy as np

from oracle import pyglm_oracle as orc


class OracleEngine(object):
    def __init__(self, N, B, n0=0, n1=None, obs="bernoulli", xi=1.0, **kw):
        self.N, self.B, self.D = N, B, N * B
        self.n0, self.n1 = n0, N if n1 is None else n1
        self.nloc = self.n1 - self.n0
        self.obs, self.xi = obs, xi
        self.datasets = []
        self.nb = self.nloc
        self.profile = False
        self.eta = np.ones(self.nloc)

    def set_noise(self, eta):
        self.eta = np.asarray(eta, float).reshape(self.nloc).copy()

    def sse(self, a, W, b):
        out = np.zeros(self.nloc)
        for i in range(self.nloc):
            r = self._reg(a, W, b, i)
            out[i] = sum(np.sum((Y[:, self.n0 + i] - r.mean(X)) ** 2) for X, Y in self.datasets)
        return out

    def add_data(self, Y, X=None, basis=None):
        if X is None:
            X = orc.convolve_with_basis(Y, basis)
        self.datasets.append((np.asarray(X).reshape(Y.shape[0], self.N, self.B), np.asarray(Y, float)))

    def design_matrix(self, i=0):
        return self.datasets[i][0]

    def _reg(self, a, W, b, i, **hyp):
        r = orc.Regression(self.N, self.B, obs=self.obs, xi=self.xi, eta=self.eta[i], **hyp)
        r.a, r.W, r.b = np.asarray(a[i]).astype(bool).copy(), np.asarray(W[i], float).copy(), np.atleast_1d(np.asarray(b, float)[i]).copy()
        return r

    def log_likelihood(self, a, W, b):
        out = np.zeros(self.nloc)
        for i in range(self.nloc):
            r = self._reg(a, W, b, i)
            out[i] = sum(r.log_likelihood(X, Y[:, self.n0 + i]).sum() for X, Y in self.datasets)
        return out

    def psi(self, a, W, b, i=0):
        X = self.datasets[i][0]
        return np.column_stack([self._reg(a, W, b, k).activation(X) for k in range(self.nloc)])

    def sweep(self, a, W, b, rho, Jw, hw, Jb, hb, c0, perm, u, z, seed, sweep, omega_override=None, host_overlap=None):
        a_new, W_new, b_new, ll = [], [], [], self.log_likelihood(a, W, b)
        if hasattr(Jw, "dense"):          # pyglm_amd.engine.BlockPrior (tables + labels)
            Jw, hw, c0 = Jw.dense()
        if host_overlap is not None:
            host_overlap()
        for i in range(self.nloc):
            S_w = np.linalg.inv(Jw[i])
            mu_w = np.einsum("mij,mj->mi", S_w, hw[i])
            r = self._reg(a, W, b, i, rho=rho[i], S_w=S_w, mu_w=mu_w, S_b=1.0 / Jb[i], mu_b=hb[i] / Jb[i])
            n = self.n0 + i
            datas, oms, off = [], [], 0
            for X, Y in self.datasets:
                datas.append((X, Y[:, n]))
                if self.obs == "gaussian":
                    oms.append(r.omega_gaussian(X.shape[0]))
                    continue
                oms.append(orc.pg_draw(r.b_func(Y[:, n]), r.activation(X), seed, orc.stream_id(n, sweep), off))
                off += X.shape[0]
            r.resample(datas, oms, perm[i], u[i], z[i])
            a_new.append(r.a)
            W_new.append(r.W)
            b_new.append(r.b[0])
        return np.array(a_new), np.array(W_new), np.array(b_new), ll

    def collect_timings(self):
        return {}
