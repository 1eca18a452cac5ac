"""(test infrastructure: imports the oracle, hence under tests/)  Measures how often the device's PG draw differs from the oracle's on the shared stream (profiles/r06_pg_mismatch.json):
(a) same z on both sides (accept/reject knife edges only); (b) sweep level -- the device draws from ITS activation (MFMA summation order),
the oracle from NumPy's (dgemv order), so psi differs by ulps as well.  The parity tests allow 2 draws per comparison (tests/_pg_agree.py).
Usage (GPU box): python tests/probe_pg_mismatch.py > gpurun_out/pg_mismatch.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyglm_oracle as orc                      # noqa: E402
from tests.test_gpu_pg_edges import _dev_draw               # noqa: E402
from tests.test_gpu_parity import _random_problem, _hyp     # noqa: E402


def main():
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    out = {}
    rng = np.random.default_rng(0)
    n = 10_000_000
    for b in (1.0, 3.0, 2.5):
        mis = 0
        for c in range(4):
            z = rng.standard_normal(n // 4) * 3.0
            got = _dev_draw(b, z, 5, orc.stream_id(c, 1))
            want = orc.pg_draw(np.full(z.size, b), z, 5, orc.stream_id(c, 1))
            mis += int((np.abs(got - want) > 1e-12 * want).sum())
        out["same_z_b=%g" % b] = dict(draws=n, mismatches=mis, rate=mis / n)
    # sweep level, Bernoulli: the shapes of tests/test_gpu_parity.py::test_sweep_vs_oracle, every neuron
    sweep = {}
    tot_m = tot_n = tot_psi = 0
    for N, B, T, rho, batch in [(12, 2, 700, 0.5, None), (60, 3, 1500, 0.5, 16), (40, 4, 900, 1.0, 7), (33, 5, 1200, 0.3, 33), (4, 1, 10000, 0.5, None),
                                (10, 3, 6100, 0.5, 4), (25, 5, 2300, 0.4, 25), (64, 5, 40000, 0.5, None)]:
        basis, X, Y, r2 = _random_problem(N, B, T, seed=N * 7 + B)
        kw = dict(rho=rho, S_w=4.0, mu_w=0.0, mu_b=-1.5, S_b=2.0)
        a = r2.random((N, N)) < 0.5
        W = r2.standard_normal((N, N, B)) * a[:, :, None]
        b = r2.standard_normal(N) - 1.5
        eng = GibbsEngine(N, B, batch=batch)
        eng.add_data(Y, X=X)
        regs = [orc.Regression(N, B, **kw) for _ in range(N)]
        rho_a, Jw, hw, Jb, hb, c0 = _hyp(regs)
        perm, u, z = make_draws(123, 4, range(N), N, N * B)
        eng.sweep(a, W, b, rho_a, Jw, hw, Jb, hb, c0, perm, u, z, seed=123, sweep=4)
        om = eng.datasets[0].OK[:T, :N].cpu().numpy()
        psi_dev = eng.psi(a, W, b)
        m = p = 0
        for nn in range(N):
            r = regs[nn]
            r.a, r.W, r.b = a[nn], W[nn], b[nn:nn + 1]
            psi = r.activation(X)
            want = orc.pg_draw(None, psi, 123, orc.stream_id(nn, 4))
            m += int((np.abs(om[:, nn] - want) > 1e-12 * want).sum())
            p += int((psi_dev[:, nn] != psi).sum())
        sweep["N%d_B%d_T%d" % (N, B, T)] = dict(draws=N * T, mismatches=m, rate=m / (N * T), psi_not_bit_equal=p / (N * T))
        tot_m, tot_n, tot_psi = tot_m + m, tot_n + N * T, tot_psi + p
        del eng
    out["sweep_level_bernoulli"] = dict(cases=sweep, draws=tot_n, mismatches=tot_m, rate=tot_m / tot_n, psi_not_bit_equal=tot_psi / tot_n)
    # sweep level, negative binomial: tests/test_gpu_model.py::test_negative_binomial_sweep_vs_oracle's shapes at 30x its T
    nb = {}
    for xi, tol in [(3.0, 1e-12), (2.5, 1e-12), (0.7, 1e-8)]:
        r2 = np.random.default_rng(4)
        N, B, T = 10, 2, 27000
        basis = orc.cosine_basis(B, L=15) / 15
        Y = r2.negative_binomial(xi, 0.8, size=(T, N)).astype(float)
        X = orc.convolve_with_basis(Y, basis)
        kw = dict(rho=0.5, S_w=2.0, mu_w=0.0, mu_b=-1.0, S_b=1.0)
        a = r2.random((N, N)) < 0.3
        W = r2.standard_normal((N, N, B)) * 0.2 * a[:, :, None]
        b = np.full(N, -1.5)
        eng = GibbsEngine(N, B, obs="negbin", xi=xi)
        eng.add_data(Y, X=X)
        regs = [orc.Regression(N, B, obs="negbin", xi=xi, **kw) for _ in range(N)]
        hyp = prior_terms(np.array([r.S_w for r in regs]), np.array([r.mu_w for r in regs]), np.ones(N), np.full(N, -1.0))
        perm, u, z = make_draws(8, 0, range(N), N, N * B)
        eng.sweep(a, W, b, np.full((N, N), 0.5), *hyp, perm, u, z, seed=8, sweep=0)
        om = eng.datasets[0].OK[:T, :N].cpu().numpy()
        m = 0
        for nn, r in enumerate(regs):
            r.a, r.W, r.b = a[nn].copy(), W[nn].copy(), b[nn:nn + 1].copy()
            want = orc.pg_draw(Y[:, nn] + xi, r.activation(X), 8, orc.stream_id(nn, 0))
            m += int((np.abs(om[:, nn] - want) > tol * want).sum())
        nb["xi=%g" % xi] = dict(draws=N * T, tol=tol, mismatches=m, rate=m / (N * T), mean_shape=float(Y.mean() + xi))
        del eng
    out["sweep_level_negbin"] = nb
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
