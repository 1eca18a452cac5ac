"""Pins oracle/pyglm_oracle.py against vectors captured from the reference's own NumPy code
(tests/golden/make_fixtures.py; SURVEY.md section 8(c) fixtures G1-G11). CPU only."""
import numpy as np
import pytest

from oracle import pyglm_oracle as orc

RT = dict(rtol=1e-12, atol=1e-12)


def make_reg(g, tag):
    N, B = g[tag + "_mu_w"].shape
    r = orc.Regression(N, B, rho=g[tag + "_rho"], mu_w=g[tag + "_mu_w"], S_w=g[tag + "_S_w"],
                       mu_b=g[tag + "_mu_b"], S_b=g[tag + "_S_b"])
    r.a, r.W, r.b = g[tag + "_a0"].copy(), g[tag + "_W0"].copy(), g[tag + "_b0"].copy()
    return r


@pytest.mark.parametrize("B,L", [(1, 100), (3, 10), (5, 100)])
def test_G1_cosine_basis(golden, B, L):
    np.testing.assert_allclose(orc.cosine_basis(B, L=L), golden["G1_cosine_B%d_L%d" % (B, L)], **RT)


@pytest.mark.parametrize("method", ["direct", "fft"])
def test_G2_convolve(golden, method):
    F = orc.convolve_with_basis(golden["G2_S"], golden["G2_basis"], method=method)
    np.testing.assert_allclose(F, golden["G2_F"], rtol=1e-12, atol=1e-14)
    Fb = orc.convolve_with_basis(golden["G2b_S"], golden["G2b_basis"], method=method)
    np.testing.assert_allclose(Fb, golden["G2b_F"], rtol=1e-12, atol=1e-13)


def test_G3_generate_lag_convention(golden):
    # test/test_generate.py:22-24: generate()'s online X equals the basis convolution of Y
    X = orc.convolve_with_basis(golden["G3_Y"], golden["G3_basis"])
    np.testing.assert_allclose(X, golden["G3_X"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(X, golden["G3_Xconv"], rtol=1e-10, atol=1e-12)
    # test/test_generate.py:50-55: identity basis column b is lag b+1
    Xi, Yi = golden["G3i_X"], golden["G3i_Y"]
    Xo = orc.convolve_with_basis(Yi, np.eye(Xi.shape[2]))
    np.testing.assert_allclose(Xo, Xi, atol=1e-12)
    for n in range(Yi.shape[1]):
        for b in range(Xi.shape[2]):
            np.testing.assert_allclose(Yi[:-(b + 1), n], Xo[(b + 1):, n, b], atol=1e-12)
    # means (models.py:153-163)
    N, B = golden["G3_W"].shape[1:]
    m = orc.GLM(N, B, basis=golden["G3_basis"])
    for n, r in enumerate(m.regressions):
        r.a, r.W, r.b = golden["G3_A"][n], golden["G3_W"][n], golden["G3_b"][n:n + 1]
    m.add_data(golden["G3_Y"])
    np.testing.assert_allclose(m.means()[0], golden["G3_means"], **RT)


@pytest.mark.parametrize("tag", ["c0", "c1", "c2", "c3"])
def test_G4_G5_G10_statistics(golden, tag):
    g = golden
    r = make_reg(g, tag)
    X, y = g[tag + "_X"], g[tag + "_y"]
    np.testing.assert_allclose(r.activation(X), g[tag + "_psi"], **RT)
    np.testing.assert_allclose(r.kappa(y), g[tag + "_kappa"], **RT)
    np.testing.assert_allclose(r.mean(X), g[tag + "_mean"], **RT)
    np.testing.assert_allclose(r.log_likelihood(X, y), g[tag + "_ll"], **RT)
    J0, h0 = r.prior_stats()
    np.testing.assert_allclose(J0, g[tag + "_J_prior"], **RT)
    np.testing.assert_allclose(h0, g[tag + "_h_prior"], **RT)
    datas = [(X, y), (g[tag + "_X2"], g[tag + "_y2"])]
    Jl, hl = r.lkhd_stats(datas, [g[tag + "_om1"], g[tag + "_om2"]])
    np.testing.assert_allclose(Jl, g[tag + "_J_lkhd"], rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(hl, g[tag + "_h_lkhd"], rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize("tag", ["c0", "c1", "c2", "c3"])
def test_G6_marginal_likelihood(golden, tag):
    g = golden
    r = make_reg(g, tag)
    Jp, hp = g[tag + "_J_prior"], g[tag + "_h_prior"]
    Jq, hq = Jp + g[tag + "_J_lkhd"], hp + g[tag + "_h_lkhd"]
    for m, ml in zip(g[tag + "_ml_masks"], g[tag + "_ml"]):
        np.testing.assert_allclose(r.marginal_likelihood(Jp, hp, Jq, hq, a=m), ml, rtol=1e-11, atol=1e-10)


@pytest.mark.parametrize("tag", ["c0", "c1", "c2", "c3"])
def test_G7_G8_full_resample(golden, tag):
    g = golden
    r = make_reg(g, tag)
    datas = [(g[tag + "_X"], g[tag + "_y"]), (g[tag + "_X2"], g[tag + "_y2"])]
    r.resample(datas, [g[tag + "_om1"], g[tag + "_om2"]], g[tag + "_perm"], g[tag + "_u"], g[tag + "_z"])
    np.testing.assert_array_equal(r.a, g[tag + "_a1"])
    np.testing.assert_allclose(r.W, g[tag + "_W1"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(r.b, g[tag + "_b1"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(r.log_likelihood(*datas[0]).sum(), g[tag + "_ll1"], rtol=1e-10)
    # the deterministic-sparsity branch consumes no uniforms (regression.py:274-275)
    assert int(g[tag + "_n_u_used"]) == (0 if tag == "c3" else len(g[tag + "_perm"]))


def test_G9_G11_model_sweep(golden):
    g = golden
    N, _, B = g["M_W0"].shape
    m = orc.GLM(N, B, basis=g["M_basis"], S_w=10.0, mu_b=-2.0)
    for n, r in enumerate(m.regressions):
        r.a, r.W, r.b = g["M_A0"][n].copy(), g["M_W0"][n].copy(), g["M_b0"][n:n + 1].copy()
    m.add_data(g["M_Y"])
    np.testing.assert_allclose(m.data_list[0][0], g["M_X"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(m.log_likelihood(), g["M_ll0"], rtol=1e-11)
    np.testing.assert_allclose(g["M_ll0_rawY"], g["M_ll0"], rtol=1e-11)
    for n, r in enumerate(m.regressions):
        datas = [(X, Y[:, n]) for X, Y in m.data_list]
        r.resample(datas, [g["M_omegas"][n]], g["M_perms"][n], g["M_us"][n], g["M_zs"][n])
    np.testing.assert_array_equal(m.adjacency, g["M_A1"])
    np.testing.assert_allclose(m.weights, g["M_W1"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m.biases, g["M_b1"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m.log_likelihood(), g["M_ll1"], rtol=1e-10)
    np.testing.assert_allclose(m.means()[0], g["M_means1"], rtol=1e-10)
    # G11 push shapes (models.py:233-236): rows = postsynaptic
    assert g["M_push_S_w"].shape == (N, N, B, B) and g["M_push_mu_w"].shape == (N, N, B) and g["M_push_rho"].shape == (N, N)
