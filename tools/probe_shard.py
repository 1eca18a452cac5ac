#!/usr/bin/env python
"""What ONE rank of an N-GPU run does, timed on one GPU: the sweep of a shard of nloc = 1024 / G neurons of cfg3 (engine level, synthetic
hyper-parameters), stage by stage.  Every sweep starts from the SAME state (77 % of the blocks active, as in the bench chain) and the prior
keeps almost every block on (rho = 0.9999), so the initial tableau sweep and the Cholesky see full-size active sets.
python tools/probe_shard.py [G=8] [sweeps=3]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
from pyglm_amd.utils.basis import cosine_basis
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, B, T = 1024, 5, 100000
nloc, D = N // G, N * B
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
eng = GibbsEngine(N, B, 0, nloc)
eng.add_data(Y, basis=cosine_basis(B, L=100) / 100)
a = rng.random((nloc, N)) < 0.77
W = rng.standard_normal((nloc, N, B)) * 0.05 * a[:, :, None]
b = np.full(nloc, -2.0)
hyp = prior_terms(np.tile(np.eye(B) * 10.0, (nloc, N, 1, 1)), np.zeros((nloc, N, B)), np.ones(nloc), np.full(nloc, -2.0))
rho = np.full((nloc, N), 0.9999)
eng.profile = True
for s in range(sweeps + 1):
    perm, u, z = make_draws(1, s, range(nloc), N, D)
    if s == 1:
        eng.collect_timings(); torch.cuda.synchronize(); t0 = time.perf_counter()
    a1, W1, b1, ll = eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=1, sweep=s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / sweeps
st = eng.collect_timings()
print("shard of %d neurons (1/%d of cfg3), batch %d: %.3f s per sweep; x%d = %.2f s" % (nloc, G, eng.nb, dt, G, dt * G))
print({k: round(v["ms"] / sweeps, 1) for k, v in st.items()}, "density after: %.3f" % a1.mean())
