"""world_size-2 gloo test of the N>1 path (CPU): neuron sharding, all_gather of (a, W, b), all_reduce of the
log-likelihood and identical network-prior draws on every rank.  The GPU engine is replaced by the test-only
oracle-backed engine (tests/_oracle_engine.py); what is under test is the distributed host logic of pyglm_amd.models."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_model(world, rank, port, out_path, kind="bernoulli"):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tests._oracle_engine import OracleEngine
    from pyglm_amd.models import SparseBernoulliGLM, SparseGaussianGLM
    from pyglm_amd.utils.basis import cosine_basis
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    np.random.seed(0)
    N, B, T = 5, 2, 400
    basis = cosine_basis(B, L=10) / 10
    Y = (np.random.rand(T, N) < 0.2).astype(float)
    if kind == "gaussian":
        Y = np.random.randn(T, N) + 0.5 * Y
        model = SparseGaussianGLM(N, basis=basis, regression_kwargs=dict(S_w=5.0, a_0=2.0, b_0=1.0), seed=11, engine_factory=OracleEngine)
    else:
        model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=5.0, mu_b=-1.0), seed=11, engine_factory=OracleEngine)
    model.add_data(Y)
    lls = [model.log_likelihood()]
    for _ in range(3):
        model.resample_model()
        lls.append(model.log_likelihood())
    means = model.means[0]
    # held-out data goes through a second (likelihood-only) engine on the same shard; raw spikes, (X, Y) tuples, and a repeat (cache hit)
    Y2 = (np.random.RandomState(5).rand(150, N) < 0.2).astype(float)
    if kind == "gaussian":
        Y2 = Y2 + np.random.RandomState(6).randn(150, N)
    held = [model.log_likelihood([Y2]), model.log_likelihood([(model._heldout_engine([Y2]).design_matrix(0), Y2)]),
            model.log_likelihood([Y2, Y[:100]]), model.log_likelihood([Y2])]
    if rank == 0:
        np.savez(out_path, held=np.array(held), A=model.adjacency, W=model.weights, b=model.biases, lls=np.array(lls), means=means,
                 rho=np.array([r.rho for r in model.regressions]), S_w=np.array([r.S_w for r in model.regressions]),
                 eta=np.array([getattr(r, "eta", 0.0) for r in model.regressions]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _worker(rank, world, port, out_path, kind="bernoulli"):
    _run_model(world, rank, port, out_path, kind)


@pytest.mark.timeout(300)
def test_two_ranks_equal_one(tmp_path):
    import torch.multiprocessing as mp
    one = str(tmp_path / "one.npz")
    two = str(tmp_path / "two.npz")
    _run_model(1, 0, 0, one)
    mp.spawn(_worker, args=(2, _free_port(), two), nprocs=2, join=True)
    a, b = np.load(one), np.load(two)
    for k in a.files:
        np.testing.assert_allclose(a[k], b[k], rtol=1e-12, atol=1e-12, err_msg=k)
    # the log-likelihood total does not depend on the number of ranks in ANY bit: the ranks all_reduce the N per-neuron values (own entries,
    # zeros elsewhere) and every rank sums them in neuron order (round 6; a scalar all_reduce regrouped the additions)
    np.testing.assert_array_equal(a["lls"], b["lls"])
    np.testing.assert_array_equal(a["held"], b["held"])
    assert a["A"].shape == (5, 5) and a["W"].shape == (5, 5, 2) and a["means"].shape == (400, 5)
    assert np.isclose(a["held"][0], a["held"][1], rtol=1e-12) and a["held"][0] == a["held"][3] and a["held"][2] < a["held"][0]
    assert np.all(np.isfinite(a["lls"]))


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_gaussian(tmp_path):
    """the Gaussian model adds one exchanged quantity: the noise variances eta (all_gather of one double per neuron)"""
    import torch.multiprocessing as mp
    one = str(tmp_path / "one.npz")
    two = str(tmp_path / "two.npz")
    _run_model(1, 0, 0, one, "gaussian")
    mp.spawn(_worker, args=(2, _free_port(), two, "gaussian"), nprocs=2, join=True)
    a, b = np.load(one), np.load(two)
    for k in a.files:
        np.testing.assert_allclose(a[k], b[k], rtol=1e-12, atol=1e-12, err_msg=k)
    assert np.all(a["eta"] > 0) and len(set(a["eta"].tolist())) == 5
