#!/usr/bin/env python
"""The flip and weight stages of a cfg3-shaped batch (N = 1024, B = 5, 256 neurons) timed on their own: T is short (the Gram is then a
small part of the sweep; the tableau and the Cholesky do not depend on T), every sweep starts from the same state -- `dens` of the blocks
active, prior rho = `rho` -- so that runs are comparable.   python tools/probe_flipweights.py [dens=0.6] [rho=0.6] [sweeps=3] [nloc=256]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import os
import pyglm_amd._lib as _l
if os.environ.get("PGL_PROBE_LIB") == "ab":         # the -DPGL_AB build (make -C pyglm_amd/csrc ab): tuning knobs from the environment
    _l.LIB_PATH = _l.LIB_PATH.replace("libpyglm_hip.so", "libpyglm_hip_ab.so")
from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
from pyglm_amd.utils.basis import cosine_basis
dens = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
rho_v = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nloc = int(sys.argv[4]) if len(sys.argv) > 4 else 256
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
eng.profile = True
for s in range(sweeps + 1):
    perm, u, z = make_draws(1, s, range(nloc), N, D)
    if s == 1:
        eng.collect_timings(); torch.cuda.synchronize(); t0 = time.perf_counter()
    a1, W1, b1, ll = eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=1, sweep=s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / sweeps
st = eng.collect_timings()
print("batch of %d neurons, density %.2f -> %.3f, %.3f s per sweep" % (nloc, dens, a1.mean(), dt))
print({k: round(v["ms"] / sweeps, 1) for k, v in st.items()}, "checksum %.12g" % float(np.abs(W1).sum()))
