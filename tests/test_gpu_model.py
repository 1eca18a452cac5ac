"""GPU tests of the drop-in API: the reference's own tests (test/test_generate.py) re-expressed on pyglm_amd, the flows of
examples/synthetic.py and examples/bernoulli_regression.py, a sweep-by-sweep replay of the population model against the
oracle, and the negative-binomial observation model."""
import numpy as np
import pytest

from oracle import pyglm_oracle as orc
from tests._pg_agree import assert_pg_agree

pytestmark = pytest.mark.gpu


def test_generate_lags_reference_test_basis():
    # reference test/test_generate.py:42-55
    from pyglm_amd.regression import SparseBernoulliRegression
    from pyglm_amd.models import NonlinearAutoregressiveModel
    np.random.seed(0)
    N, B = 2, 3
    regs = [SparseBernoulliRegression(N, B, mu_b=-2, S_b=0.1) for _ in range(N)]
    model = NonlinearAutoregressiveModel(N, regs, B=B)
    X, Y = model.generate(T=1000, keep=False)
    for n in range(N):
        for b in range(B):
            assert np.allclose(Y[:-(b + 1), n], X[(b + 1):, n, b])


def test_generate_means_reference_test_means():
    # reference test/test_generate.py:10-39
    from pyglm_amd.regression import SparseBernoulliRegression
    from pyglm_amd.models import NonlinearAutoregressiveModel
    from pyglm_amd.utils.basis import cosine_basis
    np.random.seed(1)
    N, B, L = 2, 3, 10
    basis = cosine_basis(B, L=L) / L
    regs = [SparseBernoulliRegression(N, B, mu_b=-2, S_b=0.1) for _ in range(N)]
    model = NonlinearAutoregressiveModel(N, regs, basis=basis)
    X, Y = model.generate(T=1000, keep=False)
    model.add_data(Y)
    Xtest = np.asarray(model.data_list[0][0])
    assert np.allclose(X, Xtest)
    means = model.means
    model2 = NonlinearAutoregressiveModel(N, regs, basis=basis)
    model2.add_data(Y, X=X)
    assert np.allclose(means[0], model2.means[0])
    # and against the oracle's closed form
    want = np.column_stack([orc.logistic(X.reshape(1000, -1) @ (r.a[:, None] * r.W).ravel() + r.b[0]) for r in regs])
    np.testing.assert_allclose(means[0], want, rtol=1e-10)


def test_model_sweeps_replay_against_oracle():
    """three full resample_model() sweeps (regressions + network prior push); every sweep is replayed on the CPU with the
    oracle from the same pre-sweep state, hyper-parameters, random inputs and the GPU's own omega."""
    from pyglm_amd.models import SparseBernoulliGLM
    from pyglm_amd.engine import make_draws
    from pyglm_amd.utils.basis import cosine_basis
    np.random.seed(3)
    N, B, T = 6, 2, 1500
    basis = cosine_basis(B, L=20) / 20
    true = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=1)
    for n in range(N):
        true.regressions[n].a[n] = True
        true.regressions[n].W[n, :] = -2.0
    _, Y = true.generate(T=T, keep=False)
    model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=17)
    model.add_data(Y)
    X = np.asarray(model.data_list[0][0])
    for sweep in range(3):
        pre = [(r.a.copy(), r.W.copy(), r.b.copy(), r.rho.copy(), r.mu_w.copy(), r.S_w.copy(), r.mu_b.copy(), r.S_b.copy()) for r in model.regressions]
        ll_pre = model.log_likelihood()
        model.resample_model()
        om = model.engine.datasets[0].OK[:T, :N].cpu().numpy()
        perm, u, z = make_draws(17, sweep, range(N), N, N * B)
        ll_want = 0.0
        for n, (a, W, b, rho, mu_w, S_w, mu_b, S_b) in enumerate(pre):
            r = orc.Regression(N, B, rho=rho, mu_w=mu_w, S_w=S_w, mu_b=mu_b, S_b=S_b)
            r.a, r.W, r.b = a, W, b
            ll_want += r.log_likelihood(X, Y[:, n]).sum()
            r.resample([(X, Y[:, n])], [om[:, n]], perm[n], u[n], z[n])
            np.testing.assert_array_equal(model.regressions[n].a, r.a)
            np.testing.assert_allclose(model.regressions[n].W, r.W, rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(model.regressions[n].b, r.b, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(ll_pre, ll_want, rtol=1e-10)
        # network push (models.py:233-236): rows = postsynaptic
        assert model.regressions[2].S_w.shape == (N, B, B)
        np.testing.assert_array_equal(model.regressions[2].mu_w[2], model.network._self_gaussian.mu)
        np.testing.assert_array_equal(model.regressions[2].mu_w[3], model.network._gaussian.mu)
    assert model.weights.shape == (N, N, B) and model.adjacency.shape == (N, N) and model.biases.shape == (N,)


def test_synthetic_example_flow_recovers_self_inhibition():
    """examples/synthetic.py:17-84 at its own size (N=4, B=1, L=100, T=10000 = BASELINE.json configs[0])"""
    from pyglm_amd.models import SparseBernoulliGLM
    from pyglm_amd.utils.basis import cosine_basis
    np.random.seed(0)
    T, N, B, L = 10000, 4, 1, 100
    basis = cosine_basis(B=B, L=L) / L
    true_model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.), seed=3)
    for n in range(N):
        true_model.regressions[n].a[:] = False
        true_model.regressions[n].W[:] = 0
        true_model.regressions[n].a[n] = True
        true_model.regressions[n].W[n, :] = -2.0
        true_model.regressions[n].b[:] = -1.0
    _, Y = true_model.generate(T=T, keep=True)
    ll_true = true_model.log_likelihood()
    test_model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.), seed=4)
    test_model.add_data(Y)
    lps, As, Ws = [], [], []
    for itr in range(40):
        test_model.resample_model()
        lps.append(test_model.log_likelihood())
        As.append(test_model.adjacency.copy())
        Ws.append(test_model.weights.copy())
    A_mean = np.mean(As[20:], axis=0)
    W_mean = np.mean(Ws[20:], axis=0)
    assert np.all(np.diag(A_mean) > 0.9), A_mean
    assert np.all(np.diag(W_mean[:, :, 0]) < -1.0), W_mean[:, :, 0]
    assert np.mean(lps[20:]) > ll_true - 30          # the fitted chain explains the data as well as the truth does
    assert np.mean(lps[20:]) > lps[0] - 1.0


def test_standalone_regression_flow():
    """examples/bernoulli_regression.py:12-39: start from the complement adjacency, the chain finds the true one"""
    from pyglm_amd.regression import SparseBernoulliRegression
    np.random.seed(2)
    N, B, T = 2, 1, 1000
    true_reg = SparseBernoulliRegression(N, B)
    true_reg.a[:] = [True, False]
    true_reg.W[:] = [[2.5], [0.0]]
    X = np.random.randn(T, N * B)
    y = true_reg.rvs(X=X)
    test_reg = SparseBernoulliRegression(N, B)
    test_reg.a = np.bitwise_not(true_reg.a)
    As = []
    for i in range(60):
        test_reg.resample([(X, y)], seed=5, sweep=i)
        As.append(test_reg.a.copy())
    A_mean = np.mean(As[20:], axis=0)
    assert A_mean[0] > 0.9 and A_mean[1] < 0.5, A_mean
    ll = test_reg.log_likelihood((X, y))
    assert ll.shape == (T,) and np.all(np.isfinite(ll))


@pytest.mark.parametrize("xi,tol", [(3.0, 1e-12), (2.5, 1e-12), (0.7, 1e-8)])
def test_negative_binomial_sweep_vs_oracle(xi, tol):
    """negative-binomial observations, PG shape b = y + xi (regression.py:479-489): integer xi (Devroye draws only), real-valued xi >= 1
    (Devroye draws + one exact draw of the alternate sampler for the fractional part) and xi < 1 (bins with y = 0 take the sum-of-gammas
    series; 1e-8: the series' remainder moments are computed differently on the two sides)"""
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    rng = np.random.default_rng(4)
    N, B, T = 10, 2, 900
    basis = orc.cosine_basis(B, L=15) / 15
    Y = rng.negative_binomial(xi, 0.8, size=(T, N)).astype(float)
    X = orc.convolve_with_basis(Y, basis)
    kw = dict(rho=0.5, S_w=2.0, mu_w=0.0, mu_b=-1.0, S_b=1.0)
    a = rng.random((N, N)) < 0.3
    W = rng.standard_normal((N, N, B)) * 0.2 * a[:, :, None]
    b = np.full(N, -1.5)
    eng = GibbsEngine(N, B, obs="negbin", xi=xi)
    eng.add_data(Y, X=X)
    regs = [orc.Regression(N, B, obs="negbin", xi=xi, **kw) for _ in range(N)]
    hyp = prior_terms(np.array([r.S_w for r in regs]), np.array([r.mu_w for r in regs]), np.ones(N), np.full(N, -1.0))
    perm, u, z = make_draws(8, 0, range(N), N, N * B)
    a1, W1, b1, ll = eng.sweep(a, W, b, np.full((N, N), 0.5), *hyp, perm, u, z, seed=8, sweep=0)
    om = eng.datasets[0].OK[:T, :N].cpu().numpy()
    for n, r in enumerate(regs):
        r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
        np.testing.assert_allclose(ll[n], r.log_likelihood(X, Y[:, n]).sum(), rtol=1e-10)
        want = orc.pg_draw(Y[:, n] + xi, r.activation(X), 8, orc.stream_id(n, 0))      # PG(y + xi, psi), regression.py:479-489
        assert_pg_agree(om[:, n], want, tol=tol)          # (measured: 0 of 8.1e5 such draws differ, tests/_pg_agree.py)
        r.resample([(X, Y[:, n])], [om[:, n]], perm[n], u[n], z[n])
        np.testing.assert_array_equal(a1[n], r.a)
        np.testing.assert_allclose(W1[n], r.W, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(b1[n], r.b[0], rtol=1e-7, atol=1e-9)


def test_random_shapes_fuzz_vs_oracle():
    """ten seeded random (N, B, T, rho, batch, #datasets) draws, each one full sweep replayed by the oracle"""
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    master = np.random.default_rng(2024)
    for trial in range(10):
        N = int(master.integers(1, 48))
        B = int(master.integers(1, 9))
        nds = int(master.integers(1, 3))
        Ts = [int(master.integers(3, 700)) for _ in range(nds)]
        rho = float(master.choice([0.2, 0.5, 0.8, 1.0]))
        batch = int(master.integers(1, N + 1))
        rng = np.random.default_rng(trial)
        Xs = [np.abs(rng.standard_normal((T, N, B))) * 0.4 for T in Ts]
        Ys = [(rng.random((T, N)) < 0.3).astype(float) for T in Ts]
        A = rng.standard_normal((N, N, B, B))
        S_w = np.einsum("nmij,nmkj->nmik", A, A) * 0.3 + np.eye(B)
        mu_w = rng.standard_normal((N, N, B)) * 0.2
        rho_a = np.full((N, N), rho)
        if rho < 1.0:
            rho_a[rng.random((N, N)) < 0.1] = 1.0          # a few forced entries: exercises the NaN quirk of the reference
        a = rng.random((N, N)) < 0.5
        W = rng.standard_normal((N, N, B)) * 0.5 * a[:, :, None]
        b = rng.standard_normal(N) * 0.3
        S_b, mu_b = rng.random(N) + 0.5, rng.standard_normal(N)
        eng = GibbsEngine(N, B, batch=batch)
        for X, Y in zip(Xs, Ys):
            eng.add_data(Y, X=X)
        hyp = prior_terms(S_w, mu_w, S_b, mu_b)
        perm, u, z = make_draws(trial, 0, range(N), N, N * B)
        a1, W1, b1, _ = eng.sweep(a, W, b, rho_a, *hyp, perm, u, z, seed=trial, sweep=0)
        oms = [ds.OK[:ds.T, :N].cpu().numpy() for ds in eng.datasets]
        for n in range(N):
            r = orc.Regression(N, B, rho=rho_a[n], S_w=S_w[n], mu_w=mu_w[n], S_b=float(S_b[n]), mu_b=float(mu_b[n]))
            r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
            r.resample([(X, Y[:, n]) for X, Y in zip(Xs, Ys)], [om[:, n] for om in oms], perm[n], u[n], z[n])
            msg = "trial %d (N=%d B=%d T=%s rho=%s batch=%d) neuron %d" % (trial, N, B, Ts, rho, batch, n)
            np.testing.assert_array_equal(a1[n], r.a, err_msg=msg)
            np.testing.assert_allclose(W1[n], r.W, rtol=1e-6, atol=1e-8, err_msg=msg)
            np.testing.assert_allclose(b1[n], r.b[0], rtol=1e-6, atol=1e-8, err_msg=msg)


def test_long_chain_matches_oracle_chain_statistically():
    """300 sweeps of the README-size model on the GPU and 300 sweeps of the same model driven by the oracle (CPU) from
    independent random streams: posterior summaries agree within Monte-Carlo error -- the GPU sampler targets the same law."""
    from pyglm_amd.models import SparseBernoulliGLM
    from pyglm_amd.utils.basis import cosine_basis
    from tests._oracle_engine import OracleEngine
    np.random.seed(11)
    T, N, B, L = 4000, 4, 1, 50
    basis = cosine_basis(B=B, L=L) / L
    true = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.), seed=1, engine_factory=OracleEngine)
    for n in range(N):
        true.regressions[n].a[:] = False
        true.regressions[n].W[:] = 0
        true.regressions[n].a[n] = True
        true.regressions[n].W[n, :] = -2.0
        true.regressions[n].b[:] = -1.0
    _, Y = true.generate(T=T, keep=False)
    summaries = []
    for factory, seed in [(None, 21), (OracleEngine, 22)]:
        np.random.seed(5)
        m = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.), seed=seed, engine_factory=factory)
        m.add_data(Y)
        As, Ws, bs = [], [], []
        for it in range(300):
            m.resample_model()
            if it >= 100:
                As.append(m.adjacency.copy())
                Ws.append(m.weights[:, :, 0].copy())
                bs.append(m.biases.copy())
        summaries.append((np.mean(As, 0), np.mean(Ws, 0), np.mean(bs, 0)))
    (A1, W1, b1), (A2, W2, b2) = summaries
    print("GPU chain   A diag", np.diag(A1).round(2), "W diag", np.diag(W1).round(2), "b", b1.round(2))
    print("oracle chain A diag", np.diag(A2).round(2), "W diag", np.diag(W2).round(2), "b", b2.round(2))
    assert np.all(np.diag(A1) > 0.6) and np.all(np.diag(A2) > 0.6)          # the self-inhibition is found by both
    np.testing.assert_allclose(np.diag(A1), np.diag(A2), atol=0.2)
    np.testing.assert_allclose(np.diag(W1), np.diag(W2), atol=0.4)
    np.testing.assert_allclose(b1, b2, atol=0.25)
    assert np.abs(A1 - A2).max() < 0.35


def _two_rank_worker(rank, world, port, out_path, backend="gloo", force_group=False, T=1200, pad_rows=0):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from pyglm_amd.models import SparseBernoulliGLM
    from pyglm_amd.utils.basis import cosine_basis
    dev = "cuda:0"
    if backend == "nccl":                      # one GPU per rank, RCCL collectives: the production layout
        dev = "cuda:%d" % rank
        torch.cuda.set_device(rank)
    if world > 1 or force_group:
        kw = dict(device_id=torch.device(dev)) if backend == "nccl" else {}
        dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, **kw)
    np.random.seed(0)
    N, B = 9, 2
    basis = cosine_basis(B, L=10) / 10
    Y = (np.random.rand(T, N) < 0.2).astype(float)
    model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=5.0, mu_b=-1.0), seed=11, device=dev)
    model.add_data(Y)
    if pad_rows:                               # the gathered buffer holds pad_rows zero rows per rank beyond the largest shard (uneven shards' padding, forced)
        from pyglm_amd.models import shard_bounds
        model._gather_min_rows = max(hi - lo for lo, hi in (shard_bounds(N, world, r) for r in range(world))) + pad_rows
    lls = [model.log_likelihood()]
    for _ in range(3):
        model.resample_model()
        lls.append(model.log_likelihood())
    Y2 = (np.random.RandomState(3).rand(300, N) < 0.2).astype(float)
    lls.append(model.log_likelihood([Y2]))        # held-out data: a likelihood-only engine on this rank's own device
    lls.append(model.log_likelihood([Y2]))        # (second call: served from the content-keyed cache)
    assert model._heldout_cache[1].dev == model.engine.dev and torch.cuda.current_device() == model.engine.dev.index
    if rank == 0:
        np.savez(out_path, A=model.adjacency, W=model.weights, b=model.biases, lls=np.array(lls), means=model.means[0])
    else:
        _ = model.means                       # (collective: every rank takes part in the gathers)
    if world > 1 or force_group:
        assert model.collectives == 3 + 6 + 1           # one packed all_gather per sweep, one all_reduce (N per-neuron values) per log_likelihood(), one gather of the means
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(600)
def test_one_rank_over_rccl_equals_no_process_group(tmp_path):
    """the production collectives with the ranks a one-GPU box can give them: backend "nccl" (= RCCL), world_size 1.  The seed broadcast, the
    MIN all-reduce of the integer-Gram decision, the packed all_gather_into_tensor of the shard's rows ON DEVICE buffers and the scalar
    all-reduce of the log-likelihood all execute through RCCL; state, log-likelihoods and means must equal the run without a process group
    bit for bit."""
    import torch.multiprocessing as mp
    one, rccl = str(tmp_path / "one.npz"), str(tmp_path / "rccl.npz")
    mp.spawn(_two_rank_worker, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_two_rank_worker, args=(1, _free_port(), rccl, "nccl", True), nprocs=1, join=True)
    a, b = np.load(one), np.load(rccl)
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.timeout(600)
def test_two_processes_sharing_the_gpu_equal_one(tmp_path):
    """the N > 1 path on the real engine: two processes (gloo for the host collectives, both on cuda:0) shard the neurons 5 + 4;
    state, log-likelihoods and means must equal the single-process run bit for bit (draws are keyed by the global neuron index)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    mp.spawn(_two_rank_worker, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_two_rank_worker, args=(2, port, two), nprocs=2, join=True)
    a, b = np.load(one), np.load(two)
    assert a["lls"][-1] == a["lls"][-2]
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)       # (incl. the log-likelihood totals: per-neuron values summed in neuron order)


@pytest.mark.timeout(600)
def test_two_processes_on_a_long_recording_equal_one(tmp_path):
    """the same with T = 9000: the border sums, the small-model Gram and the column norms are then added in time slices -- whose number
    follows from T and D alone, not from the 5 or 4 neurons a rank holds: still bit for bit"""
    import torch.multiprocessing as mp
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    mp.spawn(_two_rank_worker, args=(1, 0, one, "gloo", False, 9000), nprocs=1, join=True)
    mp.spawn(_two_rank_worker, args=(2, _free_port(), two, "gloo", False, 9000), nprocs=2, join=True)
    a, b = np.load(one), np.load(two)
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)       # (incl. the log-likelihood totals: per-neuron values summed in neuron order)


def _bad_prior_worker(rank, world, port, out_path):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import torch.distributed as dist
    from pyglm_amd.models import SparseBernoulliGLM
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    np.random.seed(0)
    N, B, T = 9, 2, 800
    Y = (np.random.rand(T, N) < 0.2).astype(float)
    model = SparseBernoulliGLM(N, B=B, regression_kwargs=dict(S_w=2.0, mu_b=-1.0), seed=5, device="cuda:0")
    model.add_data(Y)
    model.resample_model()
    before = (model.adjacency, model.weights, model.sweeps_done)
    S = np.array(model.regressions[7].S_w)
    S[:] = -1e-6 * np.eye(B)                       # neuron 7 lives on rank 1 (neurons 5..8)
    model.regressions[7].S_w = S
    model.regressions[7].a[:] = True
    A0 = model.adjacency
    caught = None
    try:
        model.resample_model()
    except np.linalg.LinAlgError as e:
        caught = list(e.neurons)
    unchanged = bool(np.array_equal(model.adjacency, A0) and np.array_equal(model.weights, before[1]) and model.sweeps_done == before[2])
    model.regressions[7].S_w = 2.0
    model.resample_model()                           # every rank is still in step: the next sweep's collectives match up
    np.savez(out_path + ".%d.npz" % rank, caught=np.array(caught if caught is not None else [-1]), unchanged=unchanged, W=model.weights,
             ll=model.log_likelihood())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_non_positive_definite_posterior_raises_on_every_rank(tmp_path):
    """a posterior that is not positive definite (regression.py:369-370: LinAlgError from np.linalg.cholesky) on a neuron of rank 1: the
    status flags travel in the gathered rows, so BOTH ranks raise LinAlgError naming neuron 7 in the same sweep, both keep their pre-sweep
    state, and after the prior is repaired the next sweep runs on both (no rank was left behind at a collective)"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "bad")
    mp.spawn(_bad_prior_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    for r in (r0, r1):
        assert r["caught"].tolist() == [7] and bool(r["unchanged"])
        assert np.all(np.isfinite(r["W"]))
    np.testing.assert_array_equal(r0["W"], r1["W"])
    assert float(r0["ll"]) == float(r1["ll"])


@pytest.mark.timeout(900)
def test_rccl_group_with_device_id_and_padded_rows_equals_one(tmp_path):
    """the production layout -- one process per GPU, torch.distributed backend "nccl" (= RCCL), the group constructed with `device_id` --
    against a run without a process group: bit-equal state, means and log-likelihood totals (per-neuron values summed in neuron order).
    With two GPUs: 2 ranks on 2 GPUs (5 + 4 neurons: rank 1's rows are padded in the gathered buffer).  On a one-GPU box (every box of
    this pool so far) the same code path runs with the one rank RCCL can have there, and the padding is forced (`_gather_min_rows`: three
    zero rows behind the shard's rows in the all_gather_into_tensor buffer), so pack -> pad -> gather -> slice -> unpack all execute on
    device buffers over RCCL."""
    import torch
    import torch.multiprocessing as mp
    one, pad, two = str(tmp_path / "one.npz"), str(tmp_path / "pad.npz"), str(tmp_path / "two.npz")
    mp.spawn(_two_rank_worker, args=(1, 0, one, "nccl"), nprocs=1, join=True)
    mp.spawn(_two_rank_worker, args=(1, _free_port(), pad, "nccl", True, 1200, 3), nprocs=1, join=True)
    a, b = np.load(one), np.load(pad)
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    if torch.cuda.device_count() >= 2:
        mp.spawn(_two_rank_worker, args=(2, _free_port(), two, "nccl"), nprocs=2, join=True)
        b = np.load(two)
        for k in a.files:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("N,B,n0,n1", [(37, 3, 8, 29), (70, 32, 0, 70), (130, 1, 64, 130), (5, 8, 4, 5)])
def test_row_statistics_kernel(N, B, n0, n1):
    """pgl_row_stats (round 6): [count, sum w, sum w w'] over the active off-diagonal weight vectors of every local row -- the network prior's
    sufficient statistics (networks.py:132-149) -- against NumPy; a row's numbers are bit-identical whatever shard it is computed in"""
    import torch
    from pyglm_amd.engine import GibbsEngine
    from pyglm_amd.models import host_row_stats
    rng = np.random.default_rng(N + B)
    a = rng.random((N, N)) < 0.6
    a[min(n0 + 1, N - 1), :] = False                    # an empty row
    W = rng.standard_normal((N, N, B)) * a[:, :, None]

    def stats(lo, hi):
        eng = GibbsEngine(N, B, lo, hi, batch=min(2, hi - lo))
        eng.a_dev.copy_(torch.from_numpy(a[lo:hi].astype(np.int32)))
        eng.W_dev.copy_(torch.from_numpy(W[lo:hi].reshape(hi - lo, -1)))
        return eng.row_stats().cpu().numpy()
    got = stats(n0, n1)
    want = host_row_stats(a[n0:n1], W[n0:n1], n0)
    np.testing.assert_array_equal(got[:, 0], want[:, 0])
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    mid = (n0 + n1) // 2
    if mid > n0:
        np.testing.assert_array_equal(np.concatenate([stats(n0, mid), stats(mid, n1)]), got)


def test_standalone_regression_cache_is_keyed_by_content():
    """reg.mean(X_train) then reg.mean(X_test) with equal shapes must not return the first result again (the engine cache used to be
    keyed by id(), which CPython reuses for temporaries); data edited in place is uploaded again; equal data is served from the cache"""
    from pyglm_amd.regression import SparseBernoulliRegression
    np.random.seed(5)
    N, B, T = 4, 3, 50
    reg = SparseBernoulliRegression(N, B, rho=1.0, S_w=1.0)
    want = lambda X: orc.logistic(X.reshape(T, -1) @ (reg.a[:, None] * reg.W).ravel() + reg.b[0])
    for trial in range(4):
        X1 = np.random.randn(T, N, B)           # 3-D input: _flatten_X makes a temporary view
        X2 = np.random.randn(T, N, B)
        np.testing.assert_allclose(reg.mean(X1), want(X1), rtol=1e-10)
        np.testing.assert_allclose(reg.mean(X2), want(X2), rtol=1e-10)
    eng = reg._lik_engine_cache[1]
    np.testing.assert_allclose(reg.mean(X2.copy()), want(X2), rtol=1e-10)
    assert reg._lik_engine_cache[1] is eng                                  # same content: no new upload
    X2[7] += 1.0                                                             # in-place edit
    np.testing.assert_allclose(reg.mean(X2), want(X2), rtol=1e-10)
    assert reg._lik_engine_cache[1] is not eng
    y = (np.random.rand(T) < 0.3).astype(float)
    ll = reg.log_likelihood((X1, y))
    psi = X1.reshape(T, -1) @ (reg.a[:, None] * reg.W).ravel() + reg.b[0]
    np.testing.assert_allclose(ll, y * psi - np.log1p(np.exp(psi)), rtol=1e-9, atol=1e-12)
    reg.resample([(X1, y)], seed=3)
    eng = reg._engine_cache[1]
    reg.resample([(X1, y)], seed=3, sweep=1)
    assert reg._engine_cache[1] is eng
    reg.resample([(X1, 1.0 - y)], seed=3, sweep=2)                           # different spikes, same X
    assert reg._engine_cache[1] is not eng


def test_one_engine_takes_changing_input_forms_sweep_after_sweep():
    """round 5: a shard's chain state and a sweep's inputs travel as ONE block (engine._io_reserve: one pinned upload, cached pgl_sweep_t).
    The block is re-laid out when the inputs change form -- a dense prior (N x N blocks), then a table prior with labels (what a network push
    produces), an omega override, recorded log-odds, a prefix run -- and every sweep must equal the same sweep on a fresh engine."""
    import torch
    from pyglm_amd.engine import GibbsEngine, BlockPrior, make_draws, prior_terms
    rng = np.random.default_rng(12)
    N, B, T = 9, 2, 600
    Y = (rng.random((T, N)) < 0.15).astype(float)
    X = rng.random((T, N, B)) * 0.2
    a0 = rng.random((N, N)) < 0.5
    W0 = rng.standard_normal((N, N, B)) * 0.1 * a0[:, :, None]
    b0 = np.full(N, -1.5)
    rho = np.full((N, N), 0.4)
    dense = prior_terms(np.tile(np.eye(B) * 2.0, (N, N, 1, 1)), np.zeros((N, N, B)), np.full(N, 2.0), np.full(N, -1.0))
    Ju, hu, _, _, cu = prior_terms(np.array([np.eye(B) * 2.0, np.eye(B) * 0.5])[None], np.zeros((1, 2, B)), np.ones(1), np.zeros(1))
    label = np.zeros((N, N), dtype=np.int32)
    label[np.arange(N), np.arange(N)] = 1
    table = (BlockPrior(Ju[0], hu[0], cu[0], label), None, dense[2], dense[3], None)
    om = rng.random((T, N)) * 0.3 + 0.05

    def run(eng, form, sweep, state, **kw):
        perm, u, z = make_draws(3, sweep, range(N), N, N * B)
        hyp = dense if form == "dense" else table
        return eng.sweep(*state, rho, *hyp, perm, u, z, seed=3, sweep=sweep, **kw)

    plan = [("dense", {}), ("table", {}), ("dense", dict(omega_override=[om])), ("table", dict(nrun=4)), ("dense", {}), ("table", {})]
    eng = GibbsEngine(N, B)
    eng.add_data(Y, X=X)
    state = (a0, W0, b0)
    for s, (form, kw) in enumerate(plan):
        eng.keep_logodds = s == 4
        got = run(eng, form, s, state, **kw)
        fresh = GibbsEngine(N, B)
        fresh.add_data(Y, X=X)
        fresh.keep_logodds = eng.keep_logodds
        want = run(fresh, form, s, state, **kw)
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
        if eng.keep_logodds:
            assert torch.equal(eng.logodds.nan_to_num(), fresh.logodds.nan_to_num())
        # the engine's device state is what it returned (packed_state reads it for the per-sweep all_gather)
        np.testing.assert_array_equal(eng.a_dev.cpu().numpy().astype(bool), got[0])
        np.testing.assert_array_equal(eng.W_dev.cpu().numpy().reshape(N, N, B), got[1])
        state = (got[0], got[1], got[2])
        del fresh
