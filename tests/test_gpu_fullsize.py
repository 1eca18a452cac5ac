"""GPU checks at BASELINE.json's full sizes: the NumPy oracle itself on the box's host cores where it finishes in a minute or two
(configs[1] whole: 128 neurons, every flip proposal; two neurons of configs[2] and of configs[3] at their own size), and
size-independent properties where it cannot (configs[2]-shaped per-kernel identities, a maximum-size dense Cholesky, D = 16384,
configs[4]-shaped sweeps, D = 32768, T = 200000, against long-double blocks)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(N, B, T, L=100, seed=0):
    from pyglm_amd.utils.basis import cosine_basis
    rng = np.random.default_rng(seed)
    basis = cosine_basis(B, L=L) / L
    Y = (rng.random((T, N)) < 0.08).astype(np.float64)
    return basis, Y, rng


def test_gram_identities_at_cfg3_shape():
    """D=5120, T=100000 (configs[2]); for a handful of neurons:
    linearity  J(w1 + 2 w2) = J(w1) + 2 J(w2);  trace identity  tr J = sum_t w_t |x_t|^2;  J(1)[i][j] = (X'X)[i][j] on a torch
    fp64 reference for a band of columns;  symmetry of diagonal tiles; the border contraction against torch."""
    import torch
    from pyglm_amd.engine import GibbsEngine
    from pyglm_amd._lib import call, ptr
    N, B, T = 1024, 5, 100000
    basis, Y, rng = _problem(N, B, T)
    eng = GibbsEngine(N, B, 0, 4, batch=4)
    ds = eng.add_data(Y, basis=basis)
    D, Dp, ldj = eng.D, eng.Dp, eng.ldj
    W = torch.zeros(ds.Tp, 4, dtype=torch.float64, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    W[:T, 0] = torch.rand(T, generator=g, device="cuda", dtype=torch.float64) * 0.25
    W[:T, 1] = torch.rand(T, generator=g, device="cuda", dtype=torch.float64) * 0.25
    W[:T, 2] = W[:T, 0] + 2 * W[:T, 1]
    W[:T, 3] = 1.0
    J = eng.Jslots[0]
    call("pgl_weighted_gram", ptr(ds.X), Dp, Dp, ptr(W), 4, ds.Tp, D, 4, ptr(J), ldj, ldj * ldj, 0, None)
    torch.cuda.synchronize()
    L = [torch.tril(J[k, :D, :D]) for k in range(4)]
    scale = L[2].abs().max().item()
    assert (L[2] - (L[0] + 2 * L[1])).abs().max().item() <= 1e-12 * scale                      # linearity in omega
    X = ds.X[:T, :D]
    for k in range(4):
        tr = torch.diagonal(L[k]).sum().item()
        want = (W[:T, k, None] * X * X).sum().item()
        assert abs(tr - want) <= 1e-11 * abs(want)                                               # trace identity
    cols = slice(2000, 2256)
    ref = X.t() @ X[:, cols]                                                                     # (D, 256) fp64 reference
    got = J[3, :D, cols]
    mask = torch.arange(D, device="cuda")[:, None] >= torch.arange(2000, 2256, device="cuda")[None, :]
    assert ((got - ref)[mask]).abs().max().item() <= 1e-11 * ref.abs().max().item()
    blk = J[3, 1280:1408, 1280:1408]                                                             # a diagonal tile is written in full
    assert (blk - blk.t()).abs().max().item() <= 1e-12 * blk.abs().max().item()
    # the same four Grams through the integer matrix cores (what gram='auto' runs at this shape): residue planes, int8 GEMM mod p, CRT
    assert ds.int8 and eng._i8_scratch is not None and eng._i8_scratch[2] >= 4
    J8 = torch.zeros(4, ldj, ldj, dtype=torch.float64, device="cuda")
    eng._i8_group(ds, ptr(W), 4, 4, ptr(J8), 0)
    torch.cuda.synchronize()
    L8 = [torch.tril(J8[k, :D, :D]) for k in range(4)]
    for k in range(4):
        assert (L8[k] - L[k]).abs().max().item() <= 1e-12 * L[k].abs().max().item()            # against the fp64 kernel
        tr = torch.diagonal(L8[k]).sum().item()
        want = (W[:T, k, None] * X * X).sum().item()
        assert abs(tr - want) <= 1e-11 * abs(want)
    assert (L8[2] - (L8[0] + 2 * L8[1])).abs().max().item() <= 1e-12 * scale
    assert ((J8[3, :D, cols] - ref)[mask]).abs().max().item() <= 1e-11 * ref.abs().max().item()
    del J8, L8
    # border sums  [Omega|Kappa]' [X, 1]
    OK = torch.zeros(ds.Tp, 2 * eng.ldn, dtype=torch.float64, device="cuda")
    OK[:T, :4] = W[:T]
    OK[:T, eng.ldn:eng.ldn + 4] = Y_dev = torch.from_numpy(Y[:, :4]).cuda() - 0.5
    call("pgl_contract_tn", ptr(OK), 2 * eng.ldn, 2 * eng.ldn, ptr(ds.X), Dp, Dp, ptr(eng.border), Dp, 2 * eng.ldn, D + 1, ds.Tp, 1.0, 0.0, None)
    torch.cuda.synchronize()
    want = OK[:T].t() @ ds.X[:T, :D + 1]
    assert (eng.border[:, :D + 1] - want).abs().max().item() <= 1e-11 * want.abs().max().item()


def test_cfg2_full_sweeps_invariants():
    """configs[1] at full size: two complete resample_model() sweeps; checks that need no CPU replay:
    the weight draw with z = 0 is the posterior mean (J_SS mu = h_S), with z = e_k it moves by L^-T e_k (U'U = J_SS);
    the flips respect rho in {0,1}; log-likelihood equals the closed form from a torch fp64 activation; determinism."""
    import torch
    from pyglm_amd.models import SparseBernoulliGLM
    from pyglm_amd.engine import make_draws, prior_terms
    N, B, T = 128, 5, 50000
    basis, Y, rng = _problem(N, B, T)
    np.random.seed(0)
    model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=3)
    model.add_data(Y)
    eng = model.engine
    for _ in range(2):
        model.resample_model()
    A, Wt, b = model.adjacency, model.weights, model.biases
    assert np.all(Wt[~A] == 0) and np.all(np.isfinite(Wt)) and np.all(np.isfinite(b))
    # closed-form log-likelihood from an independent fp64 activation
    X = eng.datasets[0].X[:T, :eng.D]
    psi = X @ torch.from_numpy((A[:, :, None] * Wt).reshape(N, -1).T).cuda() + torch.from_numpy(b).cuda()
    Yd = torch.from_numpy(Y).cuda()
    want = (Yd * psi - torch.log1p(torch.exp(psi))).sum().item()
    assert abs(model.log_likelihood() - want) <= 1e-10 * abs(want)
    # posterior-mean identity on the engine's last assembled batch: replay one sweep with z = 0 and rho in {0,1}
    regs = model.regressions
    a0, W0, b0 = model._local_state()
    rho = np.where(a0, 1.0, 0.0)                       # deterministic sparsity: a stays as it is, no flips
    S_w, mu_w = np.array([r.S_w for r in regs]), np.array([r.mu_w for r in regs])
    hyp = prior_terms(S_w, mu_w, np.array([r.S_b[0, 0] for r in regs]), np.array([r.mu_b[0] for r in regs]))
    perm, u, z = make_draws(1, 0, range(N), N, N * B)
    a1, W1, b1, _ = eng.sweep(a0, W0, b0, rho, *hyp, perm, u, np.zeros_like(z), seed=3, sweep=7)
    np.testing.assert_array_equal(a1, a0)
    a2, W2, b2, _ = eng.sweep(a0, W0, b0, rho, *hyp, perm, u, np.zeros_like(z), seed=3, sweep=7)
    np.testing.assert_array_equal(W1, W2)              # same seed/sweep -> bitwise reproducible
    for n in (0, 17, N - 1):                           # J_SS mu = h_S on the assembled posterior
        Jp, hp = eng.posterior(n)
        m = np.concatenate((np.repeat(a0[n], B), [True]))
        mu = np.concatenate((W1[n][a0[n]].ravel(), [b1[n]]))
        resid = Jp[np.ix_(m, m)] @ mu - hp[m]
        assert np.abs(resid).max() <= 1e-8 * np.abs(hp[m]).max()


@pytest.mark.parametrize("N", [2048, 4096])
def test_dense_maximum_size_cholesky_path(N):
    """a dense-prior regression (rho = 1: regression.py:153-155, 274-275 -- no flips, every block on) at D = 16 384 and at BASELINE.json
    configs[4]'s own D = 32 768 (N = 4096, B = 8; T kept short, the weight draw does not depend on it): the blocked Cholesky with the
    forward solve riding along as column na, the panel-wise backward solve and the rank-256 MFMA updates on a 32 769-dimensional
    system -- the largest the code will ever factor; checked through J mu = h and (x - mu)' J (x - mu) = z'z."""
    import gc
    import torch
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    gc.collect()
    torch.cuda.empty_cache()
    B, T = 8, 4000
    basis, Y, rng = _problem(N, B, T, L=50, seed=2)
    eng = GibbsEngine(N, B, 0, 2, batch=2)
    eng.add_data(Y, basis=basis)
    a = np.ones((2, N), dtype=bool)
    W = np.zeros((2, N, B))
    b = np.full(2, -2.0)
    hyp = prior_terms(np.tile(np.eye(B) * 0.5, (2, N, 1, 1)), np.zeros((2, N, B)), np.ones(2), np.full(2, -2.0))
    perm, u, z = make_draws(5, 0, range(2), N, N * B)
    a1, W1, b1, _ = eng.sweep(a, W, b, np.ones((2, N)), *hyp, perm, u, np.zeros_like(z), seed=5, sweep=0)
    assert a1.all()
    D = N * B
    M = torch.tril(eng.Jbuf[1, :D + 2, :D + 2])
    Jfull = M[:D + 1, :D + 1] + torch.tril(M[:D + 1, :D + 1], -1).t()
    h = M[D + 1, :D + 1]
    mu = torch.from_numpy(np.concatenate((W1[1].ravel(), [b1[1]]))).cuda()
    resid = (Jfull @ mu - h).abs().max().item()
    assert resid <= 1e-7 * h.abs().max().item()
    # and the draw itself: x - mu = L^-T z  <=>  (x - mu)' J (x - mu) = z'z
    a2, W2, b2, _ = eng.sweep(a, W, b, np.ones((2, N)), *hyp, perm, u, z, seed=5, sweep=0)
    d = torch.from_numpy(np.concatenate((W2[1].ravel(), [b2[1]]))).cuda() - mu
    q = (d @ (Jfull @ d)).item()
    assert abs(q - float(z[1] @ z[1])) <= 1e-7 * float(z[1] @ z[1])


def test_cfg5_shape_sweep_with_flips():
    """BASELINE.json configs[4] shape (N=4096, B=8, T=200000: D=32768, 157 GB resident) for two local neurons with a ~50 % dense
    chain: blocks of the likelihood Gram against extended-precision NumPy (regression.py:251-252 on 64 x 32 blocks, the GPU's omega), the final sweep tableau
    against its definition (M_SS = -J_SS^-1 on a probe vector) and the weight draw against an independent torch fp64 Cholesky solve
    ((x - mu)' J_SS (x - mu) = z'z on a ~16000-dimensional active system)."""
    import gc
    import torch
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    gc.collect()
    torch.cuda.empty_cache()
    N, B, T, nloc = 4096, 8, 200000, 2
    basis, Y, rng = _problem(N, B, T)
    eng = GibbsEngine(N, B, 0, nloc, batch=nloc, visit_order=False)     # J's row order + full-tableau updates: the final tableau is sweep(A, S)
    eng.add_data(Y, basis=basis)
    a = rng.random((nloc, N)) < 0.5
    W = rng.standard_normal((nloc, N, B)) * 0.05 * a[:, :, None]
    b = np.full(nloc, -2.0)
    # a tight slab (S_w = 1e-3 I) makes the flip log-odds ~ logit(rho): the chain stays about half dense
    hyp = prior_terms(np.tile(np.eye(B) * 1e-3, (nloc, N, 1, 1)), np.zeros((nloc, N, B)), np.ones(nloc), np.full(nloc, -2.0))
    perm, u, z = make_draws(5, 0, range(nloc), N, N * B)
    a1, W1, b1, _ = eng.sweep(a, W, b, np.full((nloc, N), 0.5), *hyp, perm, u, z, seed=5, sweep=0)
    D = N * B
    assert np.all(W1[~a1] == 0) and 0.3 * N < a1[0].sum() < 0.7 * N and not np.array_equal(a1, a)
    # the likelihood Gram at this configuration's own size (time-sliced integer path) against an EXTENDED-PRECISION NumPy product on blocks of
    # J: regression.py:251-252 restricted to 64 rows x 32 columns (the whole J is 8.6 GB per neuron and X 52 GB: blocks are what the host can
    # take; a float64 dgemm over 200 000 bins is itself off by 1e-14 |x_i||omega x_j|, hence long double), with the GPU's own omega
    ds = eng.datasets[0]
    assert ds.int8 and eng._i8_scratch[6] > 0                       # in time slices
    for r0, c0 in ((20000, 5000), (32768 - 64, 32768 - 64 - 32), (64, 0)):
        rows, cols = slice(r0, r0 + 64), slice(c0, c0 + 32)
        Xr, Xc = ds.X[:T, rows].cpu().numpy(), ds.X[:T, cols].cpu().numpy()
        for i in range(nloc):
            om = ds.OK[:T, i].cpu().numpy()
            want = np.asarray((Xr * om[:, None]).astype(np.longdouble).T.dot(Xc.astype(np.longdouble)), dtype=np.float64)
            got = eng.Jbuf[i, rows, cols].cpu().numpy()                # (off-diagonal blocks below the diagonal: the block-diagonal prior adds nothing)
            den = np.outer(np.sqrt((Xr * Xr).sum(0)), np.sqrt(((om[:, None] * Xc) ** 2).sum(0)))
            err = np.abs(got - want) / np.maximum(den, 1e-300)
            # measured 1.0e-15 .. 1.4e-15 at the worst entry, 2.9e-16 .. 3.3e-16 rms.  (With X rounded to nearest it was 1.0e-14 / 3.6e-15: this
            # configuration's first basis functions are narrow, their columns take few distinct values, and repeated values rounded the same
            # way -- the reason the planes of X are dithered, pgl_i8gram.hip.)
            print("cfg5 block (%d, %d) neuron %d: max error %.2e, rms %.2e of |x_i||omega x_j|" % (r0, c0, i, err.max(), np.sqrt(np.mean(err ** 2))))
            assert err.max() < 3e-15 and np.sqrt(np.mean(err ** 2)) < 7e-16, (r0, c0, i, err.max())
    for i in range(nloc):
        M = torch.tril(eng.Jbuf[i, :D + 2, :D + 2])
        m = torch.from_numpy(np.concatenate((np.repeat(a1[i], B), [True]))).cuda()
        idx = torch.nonzero(m)[:, 0]
        k = idx.numel()
        Js = M[:D + 1, :D + 1][idx][:, idx]
        Js = Js + torch.tril(Js, -1).t()
        h = M[D + 1, :D + 1][idx]
        del M
        mu = torch.cholesky_solve(h[:, None], torch.linalg.cholesky(Js))[:, 0]
        d = torch.from_numpy(np.concatenate((W1[i][a1[i]].ravel(), [b1[i]]))).cuda() - mu
        zz = float(z[i, :k] @ z[i, :k])
        assert abs((d @ (Js @ d)).item() / zz - 1) < 1e-8
        low = torch.tril(eng.Mtab[i, :D + 1, :D + 1])
        Ms = low[idx][:, idx]
        del low
        Ms = Ms + torch.tril(Ms, -1).t()
        v = torch.from_numpy(np.random.default_rng(3).standard_normal(k)).cuda()
        assert (Js @ (Ms @ v) + v).abs().max().item() < 1e-8 * v.abs().max().item()
        del Js, Ms
    del eng
    gc.collect()
    torch.cuda.empty_cache()


def _fullsize_sweep_checks(obs, N, B, T, nloc, rho, S_w, xi=1.0, seed=11):
    """one sweep of an nloc-neuron shard at a BASELINE.json configuration's own size, from the same state with the likelihood Gram on the
    integer matrix cores and on the fp64 kernel: identical decisions, weights equal to 1e-8; on the fp64 engine (plain tableau order) the
    weight draw against an independent torch fp64 Cholesky solve ((x - mu)' J_SS (x - mu) = z'z), J_SS mu = h_S with z = 0, and the
    final sweep tableau against its definition (M_SS = -J_SS^-1 on a probe vector)."""
    import gc
    import torch
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    gc.collect()
    torch.cuda.empty_cache()
    basis, Y, rng = _problem(N, B, T)
    if obs == "negbin":
        Y = np.random.default_rng(1).negative_binomial(2, 0.85, size=(T, N)).astype(np.float64)
    D = N * B
    a = rng.random((nloc, N)) < 0.8                     # ~4100 initial active rows at cfg3: the chunked initial sweep of the tableau
    if rho == 1.0:
        a[:] = True
    W = rng.standard_normal((nloc, N, B)) * 0.05 * a[:, :, None]
    b = np.full(nloc, -2.0)
    hyp = prior_terms(np.tile(np.eye(B) * S_w, (nloc, N, 1, 1)), np.zeros((nloc, N, B)), np.ones(nloc), np.full(nloc, -2.0))
    perm, u, z = make_draws(seed, 0, range(nloc), N, D)
    rho_a = np.full((nloc, N), rho)
    outs = {}
    for gram in ("int8", "fp64"):
        eng = GibbsEngine(N, B, 0, nloc, batch=nloc, obs=obs, xi=xi, gram=gram, visit_order=(gram == "int8"))
        ds = eng.add_data(Y, basis=basis)
        assert ds.int8 == (gram == "int8")
        outs[gram] = eng.sweep(a, W, b, rho_a, *hyp, perm, u, z, seed=seed, sweep=0)
        if gram == "int8":
            del eng, ds
            gc.collect()
            torch.cuda.empty_cache()
    a8, W8, b8, ll8 = outs["int8"]
    a1, W1, b1, ll1 = outs["fp64"]
    np.testing.assert_array_equal(a8, a1)
    np.testing.assert_allclose(W8, W1, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(b8, b1, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ll8, ll1, rtol=1e-12)
    assert np.all(W1[~a1] == 0)
    if rho < 1.0:
        assert not np.array_equal(a1, a) and 0.05 * N < a1[0].sum() < 0.98 * N
    for i in (0, nloc - 1):
        M = torch.tril(eng.Jbuf[i, :D + 2, :D + 2])
        idx = torch.nonzero(torch.from_numpy(np.concatenate((np.repeat(a1[i], B), [True]))).cuda())[:, 0]
        k = idx.numel()
        Js = M[:D + 1, :D + 1][idx][:, idx]
        Js = Js + torch.tril(Js, -1).t()
        h = M[D + 1, :D + 1][idx]
        del M
        mu = torch.cholesky_solve(h[:, None], torch.linalg.cholesky(Js))[:, 0]
        d = torch.from_numpy(np.concatenate((W1[i][a1[i]].ravel(), [b1[i]]))).cuda() - mu
        zz = float(z[i, :k] @ z[i, :k])
        assert abs((d @ (Js @ d)).item() / zz - 1) < 1e-8                                   # the draw: x - mu = L^-T z
        if rho < 1.0:
            low = torch.tril(eng.Mtab[i, :D + 1, :D + 1])
            Ms = low[idx][:, idx]
            del low
            Ms = Ms + torch.tril(Ms, -1).t()
            v = torch.from_numpy(np.random.default_rng(3).standard_normal(k)).cuda()
            assert (Js @ (Ms @ v) + v).abs().max().item() < 1e-8 * v.abs().max().item()     # tableau: M_SS = -J_SS^-1
            del Ms
        del Js
    # z = 0: the draw is the posterior mean, J_SS mu = h_S
    a2, W2, b2, _ = eng.sweep(a, W, b, rho_a, *hyp, perm, u, np.zeros_like(z), seed=seed, sweep=0)
    np.testing.assert_array_equal(a2, a1)
    Jp, hp = eng.posterior(0)
    m = np.concatenate((np.repeat(a2[0], B), [True]))
    mu = np.concatenate((W2[0][a2[0]].ravel(), [b2[0]]))
    assert np.abs(Jp[np.ix_(m, m)] @ mu - hp[m]).max() <= 1e-8 * np.abs(hp[m]).max()
    del eng
    gc.collect()
    torch.cuda.empty_cache()


def test_cfg3_full_size_sweep_int8_equals_fp64():
    """BASELINE.json configs[2] -- the metric's own configuration, N = 1024, B = 5, T = 100 000 -- on an 8-neuron shard"""
    _fullsize_sweep_checks("bernoulli", 1024, 5, 100000, 8, rho=0.5, S_w=1e-3)


def test_cfg4_full_size_sweep_int8_equals_fp64():
    """BASELINE.json configs[3]: NegativeBinomialGLM N = 512, B = 5, T = 100 000 (PG shape b = y + xi; dense prior: rho = 1, no flips)
    on an 8-neuron shard"""
    _fullsize_sweep_checks("negbin", 512, 5, 100000, 8, rho=1.0, S_w=1.0, xi=2.0)


def _oracle_at_full_size(obs, N, B, T, rho, S_w, xi=1.0, nprop=32, seed=21, nloc=2, gram="auto"):
    """Two neurons of a BASELINE.json configuration at its OWN size, one sweep through the default path (gram='auto': the integer Gram),
    against the oracle -- the NumPy restatement of regression.py:225-262, 282-320, 323-340 on the box's host cores -- fed the GPU's own
    omega: (i) the assembled posterior (J, h) against prior_stats + lkhd_stats, error relative to |x_i| |omega x_j|; (ii) the first `nprop`
    collapsed-flip proposals: log-odds to 1e-8, decisions bit-equal (each costs the oracle two dense Choleskys of the ~3000-dim active
    block, so not all 1024); (iii) the weight draw against gaussian_info_draw on the final active set."""
    import gc
    import torch
    from oracle import pyglm_oracle as orc
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    gc.collect()
    torch.cuda.empty_cache()
    basis, Y, rng = _problem(N, B, T)
    if obs == "negbin":
        Y = np.random.default_rng(1).negative_binomial(2, 0.85, size=(T, N)).astype(np.float64)
    D = N * B
    a = rng.random((nloc, N)) < (1.0 if rho == 1.0 else 0.6)
    W = rng.standard_normal((nloc, N, B)) * 0.05 * a[:, :, None]
    b = np.full(nloc, -2.0)
    hyp = prior_terms(np.tile(np.eye(B) * S_w, (nloc, N, 1, 1)), np.zeros((nloc, N, B)), np.ones(nloc), np.full(nloc, -2.0))
    perm, u, z = make_draws(seed, 0, range(nloc), N, D)
    eng = GibbsEngine(N, B, 0, nloc, batch=nloc, obs=obs, xi=xi, gram=gram)
    ds = eng.add_data(Y, basis=basis)
    assert eng.gram == gram and ds.int8                        # the path the benchmark runs
    eng.keep_logodds = True
    a1, W1, b1, ll1 = eng.sweep(a, W, b, np.full((nloc, N), rho), *hyp, perm, u, z, seed=seed, sweep=0)
    lo = eng.logodds.cpu().numpy()
    om = ds.OK[:T, :nloc].cpu().numpy()
    Xd = ds.X[:T, :D]
    na = torch.sqrt((Xd * Xd).sum(0)).cpu().numpy()
    X = Xd.cpu().numpy()
    for i in range(nloc):
        r = orc.Regression(N, B, rho=rho, mu_w=0.0, S_w=S_w, mu_b=-2.0, S_b=1.0, obs=obs, xi=xi)
        r.a, r.W, r.b = a[i].copy(), W[i].copy(), b[i:i + 1].copy()
        y = Y[:, i]
        np.testing.assert_allclose(ll1[i], r.log_likelihood(X, y).sum(), rtol=1e-11)
        # (i) posterior system
        Jp0, hp0 = r.prior_stats()
        Jl, hl = r.lkhd_stats([(X, y)], [om[:, i]])
        Jq, hq = Jp0 + Jl, hp0 + hl
        Jg, hg = eng.posterior(i)
        nb_ = torch.sqrt(((ds.OK[:T, i] ** 2)[:, None] * Xd * Xd).sum(0)).cpu().numpy()
        err = np.abs(Jg[:D, :D] - Jq[:D, :D]) / np.maximum(np.outer(na, nb_), 1e-300)
        if nloc <= 4 or i % 16 == 0:
            print("%s neuron %d: |J_gpu - J_oracle| / (|x_i| |omega x_j|): max %.2e, rms %.2e" % (obs, i, err.max(), np.sqrt(np.mean(err ** 2))))
        # measured 2.7e-15 .. 3.6e-15 at the worst entry and 5.3e-16 .. 5.8e-16 rms (the oracle's own dgemm included)
        assert err.max() < 8e-15 and np.sqrt(np.mean(err ** 2)) < 1.5e-15, (err.max(), np.sqrt(np.mean(err ** 2)))
        np.testing.assert_allclose(Jg[D, :], Jq[D, :], rtol=1e-12, atol=0)         # bias row: X' omega, sum omega (+ prior)
        np.testing.assert_allclose(hg, hq, rtol=1e-11, atol=1e-9 * np.abs(hq).max())
        del Jl, Jg, err
        # (ii) the first proposals of the collapsed flips
        if rho < 1.0:
            trace = []
            npi = nprop[i] if isinstance(nprop, (tuple, list)) else nprop
            r.collapsed_resample_a(Jp0, hp0, Jq, hq, perm[i][:npi], u[i][:npi], trace)
            assert [t[0] for t in trace] == perm[i][:npi].tolist()
            # (two Choleskys of a ~3000-dim block per proposal on either side: the 1e-15 |x_i||omega x_j| between the two J's shows as up to 6e-9)
            np.testing.assert_allclose(lo[i][:npi], [t[1] for t in trace], rtol=1e-8, atol=2e-8)
            assert [int(a1[i][t[0]]) for t in trace] == [t[2] for t in trace]      # decisions: bit-equal
            assert any(t[2] != int(a[i][t[0]]) for t in trace)                     # (some of them flip)
        # (iii) the weight draw on the final active set, from the oracle's own posterior system
        r.a = a1[i].copy()
        r.resample_W(Jq, hq, z[i])
        np.testing.assert_allclose(W1[i], r.W, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(b1[i], r.b[0], rtol=1e-7, atol=1e-9)
        assert np.all(W1[i][~a1[i]] == 0)
        del Jp0, Jq
    del eng, ds, X, Xd
    gc.collect()
    torch.cuda.empty_cache()


def test_cfg3_two_neurons_against_the_oracle_at_full_size():
    """BASELINE.json configs[2], the metric's own configuration (SparseBernoulliGLM N = 1024, B = 5, T = 100 000).  Neuron 0 is followed for 136
    proposals: a proposal window is 64 blocks, so 64..127 read a tableau strip that the update after window 0 has brought up to date and 128..135
    the trailing tableau after the ONE rank-(k0 + k1) pass that applies windows 0 and 1 together (pgl_k_flip_apply_pair) -- the window machinery
    at the metric's size against the oracle's plain Choleskys, ~0.2 s of host time per proposal"""
    _oracle_at_full_size("bernoulli", 1024, 5, 100000, rho=0.5, S_w=10.0, nprop=(136, 32))


def test_cfg2_whole_model_every_proposal_against_the_oracle_at_full_size():
    """BASELINE.json configs[1] (SparseBernoulliGLM N = 128, B = 5, T = 50 000), the WHOLE model through the default path (gram='auto' takes the
    integer Gram for the 128 neurons of this configuration, 64 per product launch: what bench.py --config cfg2 times): the posterior system of all
    128 neurons, ALL 128 flip proposals of each (16 384 decisions, bit-equal; log-odds to 1e-8) and every weight draw against the oracle -- whose
    two Choleskys per proposal are 385-dim here, so a whole sweep of a neuron costs it half a second"""
    _oracle_at_full_size("bernoulli", 128, 5, 50000, rho=0.5, S_w=10.0, nprop=128, nloc=128, seed=23)


def test_cfg4_two_neurons_against_the_oracle_at_full_size():
    """BASELINE.json configs[3] (NegativeBinomialGLM N = 512, B = 5, T = 100 000, xi = 2, dense prior: rho = 1 -- no flips, a 2561-dim draw)"""
    _oracle_at_full_size("negbin", 512, 5, 100000, rho=1.0, S_w=1.0, xi=2.0)


def test_cfg3_whole_model_in_batches_equals_the_oracle_checked_shards():
    """The whole model of the metric's configuration -- 1024 neurons as 4 batches of 256, groups of 8 per product launch: what bench.py times --
    tied to the shard path the oracle checks above: one sweep of the full engine, then the same sweep on two 2-neuron shards (one inside the
    last batch, one straddling the first batch boundary), which must reproduce the full run's rows: every random input is keyed by the global
    neuron, the integer Gram is exact, and every sum over time is cut into slices whose number follows from T and D alone (pgl_sweep.hip), so
    EVERYTHING agrees bit for bit -- decisions, log-odds, weights, log-likelihoods: what 8 GPUs would compute is what one computes."""
    import gc
    import torch
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    gc.collect()
    torch.cuda.empty_cache()
    N, B, T = 1024, 5, 100000
    D = N * B
    basis, Y, rng = _problem(N, B, T)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.05 * a[:, :, None]
    b = np.full(N, -2.0)
    hyp = prior_terms(np.tile(np.eye(B) * 10.0, (N, N, 1, 1)), np.zeros((N, N, B)), np.ones(N), np.full(N, -2.0))
    rho = np.full((N, N), 0.4)
    perm, u, z = make_draws(31, 2, range(N), N, D)
    eng = GibbsEngine(N, B)
    ds = eng.add_data(Y, basis=basis)
    assert ds.int8 and eng.nb == 256 and eng._i8_scratch[2] == 8
    eng.keep_logodds = True
    a1, W1, b1, ll1 = eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=31, sweep=2)
    lo1 = eng.logodds.cpu().numpy()
    assert np.all(W1[~a1] == 0) and not np.array_equal(a1, a)
    del eng, ds
    gc.collect()
    torch.cuda.empty_cache()
    for n0 in (1000, 255):
        sl = slice(n0, n0 + 2)
        sh = GibbsEngine(N, B, n0, n0 + 2, batch=2)
        sh.add_data(Y, basis=basis)
        sh.keep_logodds = True
        a2, W2, b2, ll2 = sh.sweep(a[sl], W[sl], b[sl], rho[sl], *[h[sl] for h in hyp], perm[sl], u[sl], z[sl], seed=31, sweep=2)
        np.testing.assert_array_equal(a2, a1[sl])
        np.testing.assert_array_equal(ll2, ll1[sl])
        np.testing.assert_array_equal(sh.logodds.cpu().numpy(), lo1[sl])
        np.testing.assert_array_equal(W2, W1[sl])
        np.testing.assert_array_equal(b2, b1[sl])
        del sh
        gc.collect()
        torch.cuda.empty_cache()
