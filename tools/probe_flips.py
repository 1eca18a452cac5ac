"""One batch of the cfg3 shape (N=1024, B=5, T from PT) through engine.sweep, for kernel-level traces of the flips / weights stages."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
from pyglm_amd.utils.basis import cosine_basis

N, B, T, nloc = (int(os.environ.get(k, d)) for k, d in (("PN", 1024), ("PB", 5), ("PT", 20000), ("PNLOC", 210)))
dens = float(os.environ.get("PDENS", 0.8))
rng = np.random.default_rng(0)
basis = cosine_basis(B, L=100) / 100
Y = (rng.random((T, N)) < 0.08).astype(np.float64)
eng = GibbsEngine(N, B, 0, nloc, batch=nloc)
eng.add_data(Y, basis=basis)
a = rng.random((nloc, N)) < dens
W = rng.standard_normal((nloc, N, B)) * 0.05 * a[:, :, None]
b = np.full(nloc, -2.0)
hyp = prior_terms(np.tile(np.eye(B) * float(os.environ.get("PSW", 1e-3)), (nloc, N, 1, 1)), np.zeros((nloc, N, B)), np.ones(nloc), np.full(nloc, -2.0))
rho = np.full((nloc, N), dens)
for sw in range(int(os.environ.get("PSWEEPS", 2))):
    perm, u, z = make_draws(5, sw, range(nloc), N, N * B)
    eng.profile = True
    a, W, b, ll = eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=5, sweep=sw)
    st = eng.collect_timings()
    print("sweep %d: density %.2f  stages(ms) %s" % (sw, a.mean(), {k: round(v["ms"], 1) for k, v in st.items()}), flush=True)
