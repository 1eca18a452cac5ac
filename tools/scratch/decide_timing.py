import sys, subprocess, os
import numpy as np, torch
sys.path.insert(0, ".")
from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
from pyglm_amd.utils.basis import cosine_basis
dens, rho_v = float(sys.argv[1]), float(sys.argv[2])
nloc = 256
N, B, T = 1024, 5, 4096
D = N * B
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
eng = GibbsEngine(N, B, 0, nloc, gram="fp64")
eng.add_data(Y, basis=cosine_basis(B, L=100) / 100)
a = rng.random((nloc, N)) < dens
W = rng.standard_normal((nloc, N, B)) * 0.05 * a[:, :, None]
b = np.full(nloc, -2.0)
hyp = prior_terms(np.tile(np.eye(B) * 10.0, (nloc, N, 1, 1)), np.zeros((nloc, N, B)), np.ones(nloc), np.full(nloc, -2.0))
rho = np.full((nloc, N), rho_v)
for s in range(2):
    perm, u, z = make_draws(1, s, range(nloc), N, D)
    eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=1, sweep=s)
torch.cuda.synchronize()
kk = 513 * 513
L = eng.Lws.reshape(-1)[: nloc * kk].reshape(nloc, kk)[:, 200000:200000 + 8 * 16].cpu().numpy().view(np.int64).reshape(nloc, 16, 8)
us = L[..., [0, 1, 2, 3, 6]] / 100.0
print("per window, microseconds: mean over neurons / max over neurons (the launch waits for the slowest)")
for w in (0, 5, 10, 15):
    print("window %2d: gather %5.0f/%5.0f  eval %5.0f/%5.0f (%.1f rounds)  flips %5.0f/%5.0f (%.1f flips, %.0f us each)  tail %5.0f/%5.0f  total %5.0f/%5.0f" % (
        w, us[:, w, 0].mean(), us[:, w, 0].max(), us[:, w, 1].mean(), us[:, w, 1].max(), L[:, w, 4].mean(), us[:, w, 2].mean(), us[:, w, 2].max(),
        L[:, w, 5].mean(), us[:, w, 2].sum() / max(1, L[:, w, 5].sum()), us[:, w, 3].mean(), us[:, w, 3].max(), us[:, w, 4].mean(), us[:, w, 4].max()))
