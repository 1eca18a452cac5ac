"""Pins the oracle's Gaussian observation model (reference regression.py:380-456, models.py:270-276) against vectors captured
from the reference (tests/golden/make_fixtures.py::main_gaussian). CPU only."""
import numpy as np
import pytest

from oracle import pyglm_oracle as orc

RT = dict(rtol=1e-12, atol=1e-12)


def make_reg(g, tag):
    N, B = g[tag + "_mu_w"].shape
    r = orc.Regression(N, B, rho=g[tag + "_rho"], mu_w=g[tag + "_mu_w"], S_w=g[tag + "_S_w"], mu_b=g[tag + "_mu_b"], S_b=g[tag + "_S_b"],
                       obs="gaussian", a_0=float(g[tag + "_a_0"]), b_0=float(g[tag + "_b_0"]), eta=float(g[tag + "_eta0"]))
    r.a, r.W, r.b = g[tag + "_a0"].copy(), g[tag + "_W0"].copy(), g[tag + "_b0"].copy()
    return r


@pytest.mark.parametrize("tag", ["g0", "g1", "g2"])
def test_gaussian_statistics(golden_gauss, tag):
    g = golden_gauss
    r = make_reg(g, tag)
    X, y = g[tag + "_X"], g[tag + "_y"]
    np.testing.assert_allclose(r.activation(X), g[tag + "_psi"], **RT)
    np.testing.assert_allclose(r.mean(X), g[tag + "_mean"], **RT)
    np.testing.assert_allclose(r.omega_gaussian(X.shape[0]), g[tag + "_omega"], **RT)
    np.testing.assert_allclose(r.kappa(y), g[tag + "_kappa"], **RT)
    np.testing.assert_allclose(r.log_likelihood(X, y), g[tag + "_ll"], **RT)
    datas = [(X, y), (g[tag + "_X2"], g[tag + "_y2"])]
    Jl, hl = r.lkhd_stats(datas, [r.omega_gaussian(X.shape[0]), r.omega_gaussian(g[tag + "_X2"].shape[0])])
    np.testing.assert_allclose(Jl, g[tag + "_J_lkhd"], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(hl, g[tag + "_h_lkhd"], rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("tag", ["g0", "g1", "g2"])
def test_gaussian_full_resample(golden_gauss, tag):
    g = golden_gauss
    r = make_reg(g, tag)
    datas = [(g[tag + "_X"], g[tag + "_y"]), (g[tag + "_X2"], g[tag + "_y2"])]
    oms = [r.omega_gaussian(X.shape[0]) for X, _ in datas]
    r.resample(datas, oms, g[tag + "_perm"], g[tag + "_u"], g[tag + "_z"])
    alpha, beta = r.resample_eta(datas, float(g[tag + "_g"]))
    np.testing.assert_array_equal(r.a, g[tag + "_a1"])
    np.testing.assert_allclose(r.W, g[tag + "_W1"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(r.b, g[tag + "_b1"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(alpha, g[tag + "_alpha1"], rtol=1e-14)
    np.testing.assert_allclose(beta, g[tag + "_beta1"], rtol=1e-10)
    np.testing.assert_allclose(r.eta, g[tag + "_eta1"], rtol=1e-10)
    np.testing.assert_allclose(r.log_likelihood(*datas[0]).sum(), g[tag + "_ll1"], rtol=1e-10)
    if tag == "g2":
        assert r.a.all()          # GaussianRegression: dense (regression.py:448-456)


def test_gaussian_model_sweeps(golden_gauss):
    g = golden_gauss
    N, _, B = g["MG_W0"].shape
    m = orc.GLM(N, B, basis=g["MG_basis"], S_w=5.0, obs="gaussian", a_0=2.0, b_0=1.0)
    for n, r in enumerate(m.regressions):
        r.a, r.W, r.b, r.eta = g["MG_A0"][n].copy(), g["MG_W0"][n].copy(), g["MG_b0"][n:n + 1].copy(), float(g["MG_eta0"][n])
    m.add_data(g["MG_Y"])
    np.testing.assert_allclose(m.data_list[0][0], g["MG_X"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(m.log_likelihood(), g["MG_ll0"], rtol=1e-11)
    np.testing.assert_allclose(m.means()[0], g["MG_means0"], rtol=1e-10, atol=1e-12)
    for sw in range(2):
        m.resample_regressions(0, sw, g["MG_perms"][sw], g["MG_us"][sw], g["MG_zs"][sw], gs=g["MG_gs"][sw])
        k = str(sw + 1)
        np.testing.assert_array_equal(m.adjacency, g["MG_A" + k])
        np.testing.assert_allclose(m.weights, g["MG_W" + k], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(m.biases, g["MG_b" + k], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose([r.eta for r in m.regressions], g["MG_eta" + k], rtol=1e-9)
        np.testing.assert_allclose(m.log_likelihood(), g["MG_ll" + k], rtol=1e-9)
