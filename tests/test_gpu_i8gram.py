"""The integer-MFMA Gram (pgl_i8_*: opt-in alternative to pgl_weighted_gram) against NumPy integer arithmetic, against the fp64 kernel
and against an extended-precision reference."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODULI = [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197]


def _beta(T):
    """bits of the scaled operands: T 2^(2 beta) < prod(p) / 2 (pgl_i8gram.hip: pgl_i8_beta)"""
    return min(50, int(np.floor((sum(np.log2(p) for p in MODULI) - 1.0 - np.log2(T)) * 0.5 - 1e-9)))


def _scale_exp(m, T):
    return np.where(m > 0, _beta(T) - np.frexp(np.maximum(m, 1e-300))[1], 0).astype(np.int64)


def _planes(P, n, Dq, Kp):
    """blocked plane layout [row / 16][K tile][row % 16][64 B] -> (n, Dq, Kp)"""
    return P.reshape(n, Dq // 16, Kp // 64, 16, 64).transpose(0, 1, 3, 2, 4).reshape(n, Dq, Kp)


def _setup(T, D, G, seed=0):
    import torch
    from pyglm_amd._lib import call, ptr, load
    rng = np.random.default_rng(seed)
    X = rng.random((T, D)) * (rng.random((T, D)) < 0.3) * 0.2
    X[:, 1] *= 1e-6                                           # a column on a very different scale
    X[:, 2] = 0.0                                             # an empty column
    Om = 0.25 * rng.gamma(4.0, 0.25, size=(T, G))
    dev = torch.device("cuda:0")
    Xd = torch.from_numpy(X).to(dev)
    Od = torch.from_numpy(Om).to(dev)
    xmax = torch.zeros(D, dtype=torch.float64, device=dev)
    wmax = torch.zeros(G, dtype=torch.float64, device=dev)
    call("pgl_i8_colmax", ptr(Xd), D, T, D, ptr(xmax), None)
    call("pgl_i8_colmax", ptr(Od), G, T, G, ptr(wmax), None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(xmax.cpu().numpy(), np.abs(X).max(0))
    np.testing.assert_array_equal(wmax.cpu().numpy(), Om.max(0))
    return X, Om, Xd, Od, xmax, wmax


def test_residue_planes_match_numpy():
    import torch
    from pyglm_amd._lib import call, ptr, load
    T, D, G = 300, 37, 2
    X, Om, Xd, Od, xmax, wmax = _setup(T, D, G)
    lib = load()
    Dq, Kp = 256, 320
    assert lib.pgl_i8_plane_bytes(D, T) == 15 * Dq * Kp and lib.pgl_i8_residue_bytes(D) == 15 * Dq * Dq
    PA = torch.full((15 * Dq * Kp,), 77, dtype=torch.int8, device="cuda:0")
    PB = torch.full((G * 15 * Dq * Kp,), 77, dtype=torch.int8, device="cuda:0")
    call("pgl_i8_planes", ptr(Xd), D, None, 0, ptr(xmax), None, ptr(PA), T, D, 1, None)
    call("pgl_i8_planes", ptr(Xd), D, ptr(Od), G, ptr(xmax), ptr(wmax), ptr(PB), T, D, G, None)
    torch.cuda.synchronize()
    PA = _planes(PA.cpu().numpy(), 15, Dq, Kp).astype(np.int64)
    PB = _planes(PB.cpu().numpy(), G * 15, Dq, Kp).reshape(G, 15, Dq, Kp).astype(np.int64)
    assert _beta(T) == 50 and _beta(112000) == 50 and _beta(113000) == 49 and _beta(200000) == 49
    eA = _scale_exp(np.abs(X).max(0), T)
    IA = np.rint(np.ldexp(X, eA[None, :].astype(np.int32))).astype(np.int64)          # (T, D), |.| < 2^50
    assert np.abs(IA).max() < 2 ** 50
    for q, p in enumerate(MODULI):
        got = PA[q, :D, :T]
        assert not ((got - IA.T) % p).any() and got.min() >= -128 and got.max() <= 127   # a signed-byte representative of the residue
        assert not PA[q, D:].any() and not PA[q, :, T:].any()                          # padding rows / time bins are zero
    for g in range(G):
        V = Om[:, g:g + 1] * X
        fB = _scale_exp(Om[:, g].max() * np.abs(X).max(0), T)
        IB = np.rint(np.ldexp(V, fB[None, :].astype(np.int32))).astype(np.int64)
        for q in (0, 7, 14):
            p = MODULI[q]
            got = PB[g, q, :D, :T]
            assert not ((got - IB.T) % p).any() and got.min() >= -128 and got.max() <= 127


@pytest.mark.parametrize("T,D,G", [(5000, 300, 3), (20000, 520, 2), (140000, 40, 1), (300, 1700, 1)])
def test_integer_gram_matches_fp64_kernel_and_reference(T, D, G):
    import torch
    from pyglm_amd._lib import call, ptr, load
    X, Om, Xd, Od, xmax, wmax = _setup(T, D, G, seed=T)
    lib = load()
    dev = "cuda:0"
    PA = torch.empty(lib.pgl_i8_plane_bytes(D, T), dtype=torch.int8, device=dev)
    PB = torch.empty(G * lib.pgl_i8_plane_bytes(D, T), dtype=torch.int8, device=dev)
    R = torch.empty(G * lib.pgl_i8_residue_bytes(D), dtype=torch.int8, device=dev)
    ldj = (D + 2 + 15) // 16 * 16
    J = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    call("pgl_i8_planes", ptr(Xd), D, None, 0, ptr(xmax), None, ptr(PA), T, D, 1, None)
    call("pgl_i8_planes", ptr(Xd), D, ptr(Od), G, ptr(xmax), ptr(wmax), ptr(PB), T, D, G, None)
    call("pgl_i8_gram", ptr(PA), ptr(PB), ptr(R), T, D, G, None)
    call("pgl_i8_crt", ptr(R), ptr(xmax), ptr(wmax), ptr(J), ldj, ldj * ldj, T, D, G, 0, None)
    torch.cuda.synchronize()
    Ji = J.cpu().numpy()[:, :D, :D]
    # (a) the exact integer answer: S = A'B on the scaled integers (Python ints), J = S 2^-(eA + fB)
    eA = _scale_exp(np.abs(X).max(0), T)
    IA = np.rint(np.ldexp(X, eA[None, :].astype(np.int32))).astype(np.int64)
    for g in (0, G - 1):
        fB = _scale_exp(Om[:, g].max() * np.abs(X).max(0), T)
        IB = np.rint(np.ldexp(Om[:, g:g + 1] * X, fB[None, :].astype(np.int32))).astype(np.int64)
        cols = [0, 1, 2, 5, D // 2, D - 1]
        S = IA.astype(object).T.dot(IB[:, cols].astype(object))                       # exact big-integer product, (D, len(cols))
        for k, j in enumerate(cols):
            rows = np.arange(j, D)
            want = np.array([float(S[i, k]) for i in rows]) * np.ldexp(1.0, -(eA[rows] + fB[j]).astype(np.int32))
            np.testing.assert_allclose(Ji[g, rows, j], want, rtol=2e-15, atol=0)     # CRT + 14-step Horner: a few ulp of the exact value
    # (b) against an extended-precision reference and the fp64 kernel: error relative to |a_i||b_j|
    Xl = X.astype(np.longdouble)
    Tp = (T + 15) // 16 * 16
    Xp = torch.zeros(Tp, (D + 1 + 15) // 16 * 16, dtype=torch.float64, device=dev)
    Xp[:T, :D] = Xd
    Wp = torch.zeros(Tp, G + (G & 1), dtype=torch.float64, device=dev)
    Wp[:T, :G] = Od
    Jn = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    call("pgl_weighted_gram", ptr(Xp), Xp.shape[1], Xp.shape[1], ptr(Wp), Wp.shape[1], Tp, D, G, ptr(Jn), ldj, ldj * ldj, 0, None)
    torch.cuda.synchronize()
    Jn = Jn.cpu().numpy()[:, :D, :D]
    low = np.tril(np.ones((D, D), dtype=bool))
    for g in range(G):
        ref = np.asarray((Xl * Om[:, g].astype(np.longdouble)[:, None]).T @ Xl, dtype=np.longdouble)
        na = np.sqrt((X * X).sum(0))
        nb = np.sqrt(((Om[:, g:g + 1] * X) ** 2).sum(0))
        den = np.maximum(np.outer(na, nb), 1e-300)
        e_int = float(np.max((np.abs(Ji[g] - ref) / den)[low]))
        e_f64 = float(np.max((np.abs(Jn[g] - ref) / den)[low]))
        assert e_int < 5e-15 and e_int < 20 * max(e_f64, 2e-16), (e_int, e_f64)
    # accumulate flag (second data set)
    call("pgl_i8_crt", ptr(R), ptr(xmax), ptr(wmax), ptr(J), ldj, ldj * ldj, T, D, G, 1, None)
    torch.cuda.synchronize()
    np.testing.assert_allclose(np.tril(J.cpu().numpy()[0, :D, :D]), 2 * np.tril(Ji[0]), rtol=1e-15)


@pytest.mark.parametrize("N,B,T,batch", [(60, 3, 1500, 16), (110, 4, 2500, None)])
def test_sweep_with_integer_gram_equals_fp64_sweep_and_oracle(N, B, T, batch):
    """a full engine sweep with gram='int8' against the default fp64 Gram (same decisions, weights to 1e-9) and against the oracle"""
    from oracle import pyglm_oracle as orc
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    from tests.test_gpu_parity import _random_problem
    basis, X, Y, rng = _random_problem(N, B, T, seed=N + 1)
    kw = dict(rho=0.5, S_w=4.0, mu_w=0.0, mu_b=-1.5, S_b=2.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * a[:, :, None]
    b = rng.standard_normal(N) - 1.5
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    hyp = prior_terms(np.array([r.S_w for r in regs]), np.array([r.mu_w for r in regs]), np.array([r.S_b[0, 0] for r in regs]),
                      np.array([r.mu_b[0] for r in regs]))
    rho = np.array([r.rho for r in regs])
    perm, u, z = make_draws(41, 3, range(N), N, N * B)
    outs, Js = [], []
    for gram in ("int8", "fp64"):
        eng = GibbsEngine(N, B, batch=batch, gram=gram)
        eng.add_data(Y[: T // 2], X=X[: T // 2])
        eng.add_data(Y[T // 2:], X=X[T // 2:])                  # two data sets: the second Gram accumulates
        outs.append(eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=41, sweep=3)[:3])
        Js.append(eng.posterior(0))
    np.testing.assert_allclose(Js[0][0], Js[1][0], rtol=0, atol=1e-13 * np.abs(Js[1][0]).max())
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-9, atol=1e-11)
    omegas = [eng.datasets[k].OK[: eng.datasets[k].T, :N].cpu().numpy() for k in range(2)]
    for n in (0, N // 2, N - 1):
        r = orc.Regression(N, B, **kw)
        r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
        r.resample([(X[: T // 2], Y[: T // 2, n]), (X[T // 2:], Y[T // 2:, n])], [omegas[0][:, n], omegas[1][:, n]], perm[n], u[n], z[n])
        np.testing.assert_array_equal(outs[0][0][n], r.a)
        np.testing.assert_allclose(outs[0][1][n], r.W, rtol=1e-7, atol=1e-9)


def test_auto_takes_the_integer_gram_where_it_pays():
    """gram='auto' (the default): int8 planes for large D and long T, the fp64 kernel for small shapes and for the Gaussian model"""
    from pyglm_amd.engine import GibbsEngine
    rng = np.random.default_rng(0)
    for N, B, T, obs, want in [(210, 5, 2100, "bernoulli", True), (210, 5, 1500, "bernoulli", False), (60, 3, 4000, "bernoulli", False),
                               (210, 5, 2100, "gaussian", False)]:
        eng = GibbsEngine(N, B, n1=4, obs=obs, batch=4)
        ds = eng.add_data((rng.random((T, N)) < 0.1).astype(float), X=rng.random((T, N, B)) * 0.1)
        assert eng.gram == "auto" and ds.int8 == want, (N, B, T, obs)
        assert (eng._i8_scratch is not None) == want


def test_non_finite_weights_give_nan_not_garbage():
    """a NaN / inf in a neuron's omega (a diverged chain) must surface as NaN in that neuron's Gram, as it does on the fp64 kernel"""
    import torch
    from pyglm_amd._lib import call, ptr, load
    T, D, G = 700, 40, 3
    X, Om, Xd, Od, xmax, wmax = _setup(T, D, G, seed=5)
    Od[123, 1] = float("nan")
    Od[55, 2] = float("inf")
    wmax.zero_()
    call("pgl_i8_colmax", ptr(Od), G, T, G, ptr(wmax), None)
    lib = load()
    dev = "cuda:0"
    PA = torch.empty(lib.pgl_i8_plane_bytes(D, T), dtype=torch.int8, device=dev)
    PB = torch.empty(G * lib.pgl_i8_plane_bytes(D, T), dtype=torch.int8, device=dev)
    R = torch.empty(G * lib.pgl_i8_residue_bytes(D), dtype=torch.int8, device=dev)
    ldj = (D + 2 + 15) // 16 * 16
    J = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    call("pgl_i8_planes", ptr(Xd), D, None, 0, ptr(xmax), None, ptr(PA), T, D, 1, None)
    call("pgl_i8_planes", ptr(Xd), D, ptr(Od), G, ptr(xmax), ptr(wmax), ptr(PB), T, D, G, None)
    call("pgl_i8_gram", ptr(PA), ptr(PB), ptr(R), T, D, G, None)
    call("pgl_i8_crt", ptr(R), ptr(xmax), ptr(wmax), ptr(J), ldj, ldj * ldj, T, D, G, 0, None)
    torch.cuda.synchronize()
    Jh = J.cpu().numpy()
    low = np.tril(np.ones((D, D), dtype=bool))
    assert np.isfinite(Jh[0, :D, :D][low]).all()
    assert np.isnan(Jh[1, :D, :D][low]).all() and np.isnan(Jh[2, :D, :D][low]).all()
    ref = (X * Om[:, 0:1]).T @ X
    np.testing.assert_allclose(Jh[0, :D, :D][low], ref[low], rtol=1e-11, atol=1e-13 * np.abs(ref).max())
