"""
Generates tests/golden/*.npz by IMPORTING THE REFERENCE (/root/reference, read-only) in this container.

Run here only (the reference never travels to the GPU box):   python tests/golden/make_fixtures.py

The reference needs two third-party packages that are not installed (pybasicbayes, pypolyagamma) and two
NumPy aliases removed in NumPy 2 (np.int, np.float).  This script provides minimal stand-in modules for
the IMPORTS ONLY (base classes and the four helper functions, written from their published definitions),
and patches every source of randomness so that what the reference consumed is recorded in the fixture:
  * pypolyagamma.pgdrawvpar     -> fills `omega` from an array this script chose (recorded)
  * sample_discrete_from_log    -> published algorithm, uniform taken from a recorded list
  * sample_gaussian(J=, h=)     -> published algorithm, normal vector taken from a recorded list
  * npr.permutation             -> recorded
Every array saved is an INPUT or an OUTPUT of a reference function; no reference source text is stored.
"""
import os
import sys
import types

import numpy as np
import scipy.linalg as sla

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

np.int = int      # removed aliases used at utils/basis.py:84, regression.py:519
np.float = float


# ----------------------------------------------------------------------------- stand-in modules
class Tape:
    """Random inputs handed to the reference, in consumption order."""

    def __init__(self):
        self.omega = None      # next omega vector(s) for pgdrawvpar
        self.uniforms = []     # consumed by sample_discrete_from_log
        self.normals = []      # consumed by sample_gaussian
        self.used_u = []
        self.used_z = []
        self.gammas = []       # consumed by sample_invgamma
        self.used_g = []


TAPE = Tape()


def _install_shims():
    pbb = types.ModuleType("pybasicbayes")
    absm = types.ModuleType("pybasicbayes.abstractions")
    util = types.ModuleType("pybasicbayes.util")
    stats = types.ModuleType("pybasicbayes.util.stats")
    text = types.ModuleType("pybasicbayes.util.text")
    dist = types.ModuleType("pybasicbayes.distributions")

    class GibbsSampling(object):
        pass

    class ModelGibbsSampling(object):
        pass

    absm.GibbsSampling = GibbsSampling
    absm.ModelGibbsSampling = ModelGibbsSampling

    def sample_discrete_from_log(p_log):
        p_log = np.asarray(p_log, dtype=float)
        u = TAPE.uniforms.pop(0)
        TAPE.used_u.append(u)
        cum = np.exp(p_log - p_log.max()).cumsum()
        return int(np.sum(u * cum[-1] > cum))

    def sample_gaussian(mu=None, Sigma=None, J=None, h=None):
        assert J is not None and h is not None
        z = TAPE.normals.pop(0)
        assert z.shape == h.shape, (z.shape, h.shape)
        TAPE.used_z.append(z)
        L = np.linalg.cholesky(J)
        return sla.solve_triangular(L, z, lower=True, trans="T") + sla.lapack.dpotrs(L, h, lower=True)[0]

    def sample_invgamma(alpha, beta):
        # published definition: 1 / Gamma(shape alpha, scale 1/beta); the standard Gamma(alpha, 1) variate is taken from the tape
        # when one is queued (recorded in the fixture), else from NumPy's global generator
        if TAPE.gammas:
            g = TAPE.gammas.pop(0)
            TAPE.used_g.append((alpha, beta, g))
            return 1.0 / (g * (1.0 / beta))
        return 1.0 / np.random.gamma(alpha, 1.0 / beta)

    stats.sample_discrete_from_log = sample_discrete_from_log
    stats.sample_gaussian = sample_gaussian
    stats.sample_invgamma = sample_invgamma
    text.progprint_xrange = lambda n, **kw: range(n)

    class Gaussian(object):
        """Fixed-parameter stand-in: the network prior's NIW draw is host-side and outside the fixtures."""

        def __init__(self, mu_0=None, sigma_0=None, kappa_0=None, nu_0=None, **kw):
            self.mu_0, self.sigma_0, self.kappa_0, self.nu_0 = mu_0, sigma_0, kappa_0, nu_0
            self.mu, self.sigma = np.array(mu_0, float), np.array(sigma_0, float)

        def resample(self, data=[]):
            self.last_data = np.array(data)

    dist.Gaussian = Gaussian

    ppg = types.ModuleType("pypolyagamma")
    ppg.get_omp_num_threads = lambda: 1
    ppg.PyPolyaGamma = lambda seed: ("ppg", seed)

    def pgdrawvpar(ppgs, n, z, out):
        om = TAPE.omega.pop(0) if isinstance(TAPE.omega, list) else TAPE.omega
        assert om.shape == out.shape
        out[:] = om

    ppg.pgdrawvpar = pgdrawvpar

    for name, m in [("pybasicbayes", pbb), ("pybasicbayes.abstractions", absm), ("pybasicbayes.util", util),
                    ("pybasicbayes.util.stats", stats), ("pybasicbayes.util.text", text),
                    ("pybasicbayes.distributions", dist), ("pypolyagamma", ppg)]:
        sys.modules[name] = m
    sys.path.insert(0, REF)


def pg_moment_matched(psi, rng):
    """A positive stand-in for omega with PG(1,psi)-like scale (mean b/(2z) tanh(z/2)); the fixture records it,
    so its law is irrelevant -- every consumer is a deterministic function of omega."""
    z = np.abs(psi) + 1e-9
    mean = np.tanh(z / 2) / (2 * z)
    return mean * rng.gamma(4.0, 0.25, size=psi.shape)


def main():
    _install_shims()
    import numpy.random as npr
    from pyglm.regression import SparseBernoulliRegression
    from pyglm.models import SparseBernoulliGLM, NonlinearAutoregressiveModel
    from pyglm.utils.basis import cosine_basis, convolve_with_basis
    import pyglm.regression as refreg

    rng = np.random.default_rng(20240601)
    out = {}

    # ---- G1 cosine_basis (utils/basis.py:61-106)
    for (B, L) in [(1, 100), (3, 10), (5, 100)]:
        out["G1_cosine_B%d_L%d" % (B, L)] = cosine_basis(B, L=L)

    # ---- G2 convolve_with_basis (utils/basis.py:5-34)
    T, N, B, L = 300, 3, 3, 10
    basis = cosine_basis(B, L=L) / L
    S = (rng.random((T, N)) < 0.2).astype(float)
    out["G2_S"], out["G2_basis"], out["G2_F"] = S, basis, convolve_with_basis(S, basis)
    Sg = rng.standard_normal((50, 2))
    bg = rng.standard_normal((7, 2))
    out["G2b_S"], out["G2b_basis"], out["G2b_F"] = Sg, bg, convolve_with_basis(Sg, bg)   # signed: no clipping

    # ---- G3 generate(): lag identity + conv == online X (test/test_generate.py:22-24, 50-55)
    np.random.seed(7)
    N, B, L = 2, 3, 10
    basis = cosine_basis(B, L=L) / L
    regs = [SparseBernoulliRegression(N, B, mu_b=-2, S_b=0.1) for _ in range(N)]
    model = NonlinearAutoregressiveModel(N, regs, basis=basis)
    Xg, Yg = model.generate(T=400, keep=False)
    out["G3_basis"], out["G3_X"], out["G3_Y"] = basis, Xg, Yg
    out["G3_W"] = model.weights
    out["G3_A"] = model.adjacency
    out["G3_b"] = model.biases
    model.add_data(Yg)
    out["G3_Xconv"] = model.data_list[0][0]
    out["G3_means"] = model.means[0]
    regs = [SparseBernoulliRegression(N, B, mu_b=-2, S_b=0.1) for _ in range(N)]
    model = NonlinearAutoregressiveModel(N, regs, B=B)
    Xi, Yi = model.generate(T=200, keep=False)
    out["G3i_X"], out["G3i_Y"] = Xi, Yi

    # ---- G4..G10: one regression, several shapes / hyper-parameter settings
    cases = [
        dict(tag="c0", N=4, B=1, T=400, rho=0.5, S_w=10.0, mu_w=0.0, mu_b=-2.0, S_b=1.0),
        dict(tag="c1", N=5, B=3, T=500, rho=0.3, S_w=2.0, mu_w=0.1, mu_b=-1.0, S_b=0.5),
        dict(tag="c2", N=6, B=2, T=300, rho="mixed", S_w="full", mu_w="rand", mu_b=0.3, S_b=2.0),
        dict(tag="c3", N=3, B=2, T=250, rho=1.0, S_w=1.0, mu_w=0.0, mu_b=0.0, S_b=1.0),      # deterministic sparsity
    ]
    for c in cases:
        tag, N, B, T = c["tag"], c["N"], c["B"], c["T"]
        np.random.seed(11)
        kw = dict(mu_b=c["mu_b"], S_b=c["S_b"])
        if c["rho"] == "mixed":
            rho = np.array([0.5, 0.9, 0.1, 1.0, 0.0, 0.4])[:N]
        else:
            rho = c["rho"]
        if isinstance(c["S_w"], str):
            A = rng.standard_normal((N, B, B))
            S_w = np.einsum("nij,nkj->nik", A, A) + 0.5 * np.eye(B)
        else:
            S_w = c["S_w"]
        mu_w = rng.standard_normal((N, B)) * 0.3 if isinstance(c["mu_w"], str) else c["mu_w"]
        reg = SparseBernoulliRegression(N, B, rho=rho if np.isscalar(rho) else rho.copy(),
                                        mu_w=mu_w if np.isscalar(mu_w) else mu_w.copy(),
                                        S_w=S_w if np.isscalar(S_w) else S_w.copy(), **kw)
        X = np.abs(rng.standard_normal((T, N, B))) * 0.3
        y = (rng.random(T) < 0.3).astype(float)
        out[tag + "_rho"], out[tag + "_mu_w"], out[tag + "_S_w"] = reg.rho.copy(), reg.mu_w.copy(), reg.S_w.copy()
        out[tag + "_mu_b"], out[tag + "_S_b"] = reg.mu_b.copy(), reg.S_b.copy()
        out[tag + "_X"], out[tag + "_y"] = X, y
        out[tag + "_a0"], out[tag + "_W0"], out[tag + "_b0"] = reg.a.copy(), reg.W.copy(), reg.b.copy()
        # G10 activation / kappa / mean, G9 per-bin log-likelihood
        psi = reg.activation(X)
        out[tag + "_psi"], out[tag + "_kappa"], out[tag + "_mean"] = psi, reg.kappa(X, y), reg.mean(X)
        out[tag + "_ll"] = reg.log_likelihood((X, y))
        # G5 prior stats
        Jp, hp = reg._prior_sufficient_statistics()
        out[tag + "_J_prior"], out[tag + "_h_prior"] = Jp, hp
        # G4 likelihood stats with omega injected (two datasets to pin the accumulation at :237-260)
        X2 = np.abs(rng.standard_normal((T // 2, N, B))) * 0.3
        y2 = (rng.random(T // 2) < 0.3).astype(float)
        om1 = pg_moment_matched(psi, rng)
        om2 = pg_moment_matched(reg.activation(X2), rng)
        out[tag + "_X2"], out[tag + "_y2"], out[tag + "_om1"], out[tag + "_om2"] = X2, y2, om1, om2
        TAPE.omega = [om1.copy(), om2.copy()]
        Jl, hl = reg._lkhd_sufficient_statistics([(X, y), (X2, y2)])
        out[tag + "_J_lkhd"], out[tag + "_h_lkhd"] = Jl, hl
        J_post, h_post = Jp + Jl, hp + hl
        # G6 marginal likelihood for several masks
        masks = (rng.random((6, N)) < 0.5)
        masks[0] = False
        masks[1] = True
        mls = []
        a_keep = reg.a.copy()
        for m in masks:
            reg.a = m.copy()
            mls.append(reg._marginal_likelihood(Jp, hp, J_post, h_post))
        reg.a = a_keep
        out[tag + "_ml_masks"], out[tag + "_ml"] = masks, np.array(mls)
        # G7 + G8: the full resample() with every random input recorded
        perm = rng.permutation(N)
        u = rng.random(N)
        z = rng.standard_normal(N * B + 1)
        orig_perm = npr.permutation
        refreg.npr.permutation = lambda n: perm.copy()
        TAPE.omega = [om1.copy(), om2.copy()]
        TAPE.uniforms = list(u)
        TAPE.used_u, TAPE.used_z = [], []

        class _Z(list):
            def pop(self, i=0):
                raise RuntimeError
        # sample_gaussian needs a vector of the (data-dependent) active size: serve it lazily
        class LazyNormals(object):
            def pop(self, i=0):
                k = int(reg.a.sum()) * B + 1
                return z[:k].copy()
        TAPE.normals = LazyNormals()
        reg.resample([(X, y), (X2, y2)])
        refreg.npr.permutation = orig_perm
        out[tag + "_perm"], out[tag + "_u"], out[tag + "_z"] = perm, u, z
        out[tag + "_n_u_used"] = np.array(len(TAPE.used_u))
        out[tag + "_a1"], out[tag + "_W1"], out[tag + "_b1"] = reg.a.copy(), reg.W.copy(), reg.b.copy()
        out[tag + "_ll1"] = reg.log_likelihood((X, y)).sum()

    # ---- G9/G11 model level: SparseBernoulliGLM, one sweep of resample_model with injected randomness
    np.random.seed(3)
    N, B, L, T = 4, 2, 20, 600
    basis = cosine_basis(B, L=L) / L
    glm = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0))
    Y = (rng.random((T, N)) < 0.15).astype(float)
    glm.add_data(Y)
    out["M_basis"], out["M_Y"], out["M_X"] = basis, Y, glm.data_list[0][0]
    out["M_A0"], out["M_W0"], out["M_b0"] = glm.adjacency.copy(), glm.weights.copy(), glm.biases.copy()
    out["M_ll0"] = np.array(glm.log_likelihood())
    out["M_ll0_rawY"] = np.array(glm.log_likelihood([Y]))           # models.py:88-91 raw-Y branch
    perms = np.array([rng.permutation(N) for _ in range(N)])
    us = rng.random((N, N))
    zs = rng.standard_normal((N, N * B + 1))
    oms = np.array([pg_moment_matched(glm.regressions[n].activation(glm.data_list[0][0]), rng) for n in range(N)])
    out["M_perms"], out["M_us"], out["M_zs"], out["M_omegas"] = perms, us, zs, oms
    state = dict(n=0)
    orig_perm = npr.permutation

    def next_perm(nn):
        p = perms[state["n"]].copy()
        return p
    refreg.npr.permutation = next_perm

    class ModelNormals(object):
        def pop(self, i=0):
            r = glm.regressions[state["n"]]
            return zs[state["n"], : int(r.a.sum()) * B + 1].copy()
    TAPE.normals = ModelNormals()
    # drive the reference's own loop (models.py:169-171) one neuron at a time so the tape index is known
    for n, reg in enumerate(glm.regressions):
        state["n"] = n
        TAPE.omega = [oms[n].copy()]
        TAPE.uniforms = list(us[n])
        reg.resample([(X_, Y_[:, n]) for (X_, Y_) in glm.data_list])
    refreg.npr.permutation = orig_perm
    out["M_A1"], out["M_W1"], out["M_b1"] = glm.adjacency.copy(), glm.weights.copy(), glm.biases.copy()
    out["M_ll1"] = np.array(glm.log_likelihood())
    out["M_means1"] = glm.means[0]
    # G11: hyper-parameter push shapes/values (models.py:228-236) with the fixed-parameter Gaussian stand-in
    glm.resample_network()
    out["M_push_S_w"] = np.array([r.S_w for r in glm.regressions])
    out["M_push_mu_w"] = np.array([r.mu_w for r in glm.regressions])
    out["M_push_rho"] = np.array([r.rho for r in glm.regressions])
    out["M_net_offdiag_data"] = glm.network._gaussian.last_data
    out["M_net_diag_data"] = glm.network._self_gaussian.last_data

    # G12: what the reference's network constructors do with their keyword arguments (networks.py:83-94 the NIW hyper-parameters of the
    # two Gaussians, :178-183 / :271-285 the mixin order that decides which keywords arrive at all).  The Gaussian stand-in records the
    # hyper-parameters it was constructed with; captured for B on both sides of the nu_0 >= B boundary.
    from pyglm.networks import NIWSparseNetwork, NIWDenseNetwork, FixedMeanSparseNetwork
    for name, cls in (("sparse", NIWSparseNetwork), ("dense", NIWDenseNetwork)):
        for B_ in (1, 2, 3, 5):
            net = cls(3, B_, nu_0=7.0, kappa_0=3.0, mu_0=0.5, sigma_0=2.0, rho=0.2, rho_self=0.9)
            g_, s_ = net._gaussian, net._self_gaussian
            out["N_%s_B%d_niw" % (name, B_)] = np.array([g_.nu_0, g_.kappa_0, g_.mu_0[0], g_.sigma_0[0, 0], s_.nu_0, s_.kappa_0, s_.mu_0[0], s_.sigma_0[0, 0]])
            out["N_%s_B%d_rho" % (name, B_)] = np.array(net.rho)
    net = FixedMeanSparseNetwork(3, 2, mu=0.7, sigma=4.0, rho=0.3)
    out["N_fixed_mu"], out["N_fixed_sigma"], out["N_fixed_rho"] = np.array(net.mu_W), np.array(net.sigma_W), np.array(net.rho)

    # G13: the initial state a regression draws from its prior under a NumPy seed (regression.py:86-92: rand(N), then one
    # multivariate_normal per presynaptic neuron, then the bias), at a size above and below where a vectorised draw would be tempting
    from pyglm.regression import SparseBernoulliRegression
    for tag, (N_, B_, kw_) in {"small": (7, 3, dict(rho=0.6, S_w=2.0, mu_w=0.3, mu_b=-1.0, S_b=0.5)),
                               "large": (300, 2, dict(rho=0.5, S_w=10.0, mu_b=-2.0))}.items():
        np.random.seed(1234)
        r_ = SparseBernoulliRegression(N_, B_, **kw_)
        out["I_%s_a" % tag], out["I_%s_W" % tag], out["I_%s_b" % tag] = np.array(r_.a), np.array(r_.W), np.array(r_.b)

    path = os.path.join(OUT, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1024))


def main_gaussian():
    """SparseGaussianRegression / GaussianRegression (regression.py:380-456) and SparseGaussianGLM (models.py:274-276):
    written to a second file so that reference_vectors.npz stays byte-stable."""
    import numpy.random as npr
    from pyglm.regression import SparseGaussianRegression, GaussianRegression
    from pyglm.models import SparseGaussianGLM
    from pyglm.utils.basis import cosine_basis
    import pyglm.regression as refreg

    rng = np.random.default_rng(20240917)
    out = {}
    cases = [
        dict(tag="g0", cls=SparseGaussianRegression, N=4, B=2, T=300, kw=dict(rho=0.5, S_w=4.0, mu_b=0.5, S_b=2.0, a_0=2.0, b_0=2.0)),
        dict(tag="g1", cls=SparseGaussianRegression, N=6, B=3, T=400, kw=dict(rho=0.3, S_w=1.5, mu_w=0.2, a_0=3.0, b_0=1.0, eta=0.37)),
        dict(tag="g2", cls=GaussianRegression, N=3, B=2, T=250, kw=dict(S_w=2.0, a_0=2.5, b_0=0.5)),     # dense: rho = 1
    ]
    for c in cases:
        tag, N, B, T = c["tag"], c["N"], c["B"], c["T"]
        np.random.seed(23)
        reg = c["cls"](N, B, **c["kw"])
        X = rng.standard_normal((T, N, B)) * 0.5
        wtrue = rng.standard_normal(N * B) * (rng.random(N * B) < 0.6)
        y = X.reshape(T, -1).dot(wtrue) + 0.3 + 0.7 * rng.standard_normal(T)
        X2 = rng.standard_normal((T // 3, N, B)) * 0.5
        y2 = X2.reshape(T // 3, -1).dot(wtrue) + 0.3 + 0.7 * rng.standard_normal(T // 3)
        out[tag + "_rho"], out[tag + "_mu_w"], out[tag + "_S_w"] = reg.rho.copy(), reg.mu_w.copy(), reg.S_w.copy()
        out[tag + "_mu_b"], out[tag + "_S_b"] = reg.mu_b.copy(), reg.S_b.copy()
        out[tag + "_a_0"], out[tag + "_b_0"], out[tag + "_eta0"] = np.array(reg.a_0), np.array(reg.b_0), np.array(reg.eta)
        out[tag + "_X"], out[tag + "_y"], out[tag + "_X2"], out[tag + "_y2"] = X, y, X2, y2
        out[tag + "_a0"], out[tag + "_W0"], out[tag + "_b0"] = reg.a.copy(), reg.W.copy(), reg.b.copy()
        out[tag + "_psi"], out[tag + "_mean"] = reg.activation(X), reg.mean(X)
        out[tag + "_omega"], out[tag + "_kappa"] = reg.omega(X, y), reg.kappa(X, y)
        out[tag + "_ll"] = reg.log_likelihood((X, y))
        Jl, hl = reg._lkhd_sufficient_statistics([(X, y), (X2, y2)])
        out[tag + "_J_lkhd"], out[tag + "_h_lkhd"] = Jl, hl
        perm = rng.permutation(N)
        u = rng.random(N)
        z = rng.standard_normal(N * B + 1)
        g = rng.standard_gamma(reg.a_0 + (T + T // 3) / 2.0)
        orig_perm = npr.permutation
        refreg.npr.permutation = lambda n: perm.copy()
        TAPE.uniforms = list(u)
        TAPE.used_u, TAPE.used_z, TAPE.used_g = [], [], []
        TAPE.gammas = [g]

        class LazyNormals(object):
            def pop(self, i=0):
                return z[: int(reg.a.sum()) * B + 1].copy()
        TAPE.normals = LazyNormals()
        reg.resample([(X, y), (X2, y2)])
        refreg.npr.permutation = orig_perm
        assert len(TAPE.used_g) == 1
        out[tag + "_perm"], out[tag + "_u"], out[tag + "_z"], out[tag + "_g"] = perm, u, z, np.array(g)
        out[tag + "_alpha1"], out[tag + "_beta1"] = np.array(TAPE.used_g[0][0]), np.array(TAPE.used_g[0][1])
        out[tag + "_a1"], out[tag + "_W1"], out[tag + "_b1"], out[tag + "_eta1"] = reg.a.copy(), reg.W.copy(), reg.b.copy(), np.array(reg.eta)
        out[tag + "_ll1"] = reg.log_likelihood((X, y)).sum()

    # model level: SparseGaussianGLM, two sweeps of the regressions with injected randomness
    np.random.seed(5)
    N, B, L, T = 4, 2, 15, 500
    basis = cosine_basis(B, L=L) / L
    glm = SparseGaussianGLM(N, basis=basis, regression_kwargs=dict(S_w=5.0, a_0=2.0, b_0=1.0))
    Y = rng.standard_normal((T, N))
    for t in range(1, T):
        Y[t] += 0.5 * Y[t - 1, ::-1]
    glm.add_data(Y)
    out["MG_basis"], out["MG_Y"], out["MG_X"] = basis, Y, glm.data_list[0][0]
    out["MG_A0"], out["MG_W0"], out["MG_b0"] = glm.adjacency.copy(), glm.weights.copy(), glm.biases.copy()
    out["MG_eta0"] = np.array([r.eta for r in glm.regressions])
    out["MG_ll0"] = np.array(glm.log_likelihood())
    out["MG_means0"] = glm.means[0]
    nsw = 2
    perms = np.array([[rng.permutation(N) for _ in range(N)] for _ in range(nsw)])
    us = rng.random((nsw, N, N))
    zs = rng.standard_normal((nsw, N, N * B + 1))
    gs = rng.standard_gamma(2.0 + T / 2.0, size=(nsw, N))
    out["MG_perms"], out["MG_us"], out["MG_zs"], out["MG_gs"] = perms, us, zs, gs
    state = dict(n=0, s=0)
    orig_perm = npr.permutation
    refreg.npr.permutation = lambda nn: perms[state["s"], state["n"]].copy()

    class ModelNormals(object):
        def pop(self, i=0):
            r = glm.regressions[state["n"]]
            return zs[state["s"], state["n"], : int(r.a.sum()) * B + 1].copy()
    TAPE.normals = ModelNormals()
    for sw in range(nsw):
        for n, reg in enumerate(glm.regressions):
            state["n"], state["s"] = n, sw
            TAPE.uniforms = list(us[sw, n])
            TAPE.gammas = [gs[sw, n]]
            reg.resample([(X_, Y_[:, n]) for (X_, Y_) in glm.data_list])
        out["MG_A%d" % (sw + 1)], out["MG_W%d" % (sw + 1)] = glm.adjacency.copy(), glm.weights.copy()
        out["MG_b%d" % (sw + 1)] = glm.biases.copy()
        out["MG_eta%d" % (sw + 1)] = np.array([r.eta for r in glm.regressions])
        out["MG_ll%d" % (sw + 1)] = np.array(glm.log_likelihood())
    refreg.npr.permutation = orig_perm

    path = os.path.join(OUT, "reference_vectors_gaussian.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
    main_gaussian()
