#!/usr/bin/env python
"""Stage times of a sweep of a small model (HIP events inside pgl_sweep): python tools/probe_small_stages.py [N=4] [B=1] [T=10000]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import os
import pyglm_amd._lib as _l
if os.environ.get("PGL_PROBE_LIB"):                 # another build of the library next to the shipped one (lib/libpyglm_hip_<name>.so)
    _l.LIB_PATH = _l.LIB_PATH.replace("libpyglm_hip.so", "libpyglm_hip_%s.so" % os.environ["PGL_PROBE_LIB"])
from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
N, B, T = (int(x) for x in (sys.argv[1:4] + ["4", "1", "10000"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
X = rng.random((T, N, B)) * 0.1
e = GibbsEngine(N, B)
e.add_data(Y, X=X)
a = rng.random((N, N)) < 0.5
W = rng.standard_normal((N, N, B)) * 0.1 * a[:, :, None]
b = np.full(N, -2.0)
hyp = prior_terms(np.tile(np.eye(B) * 2.0, (N, N, 1, 1)), np.zeros((N, N, B)), np.ones(N) * 2, np.full(N, -2.0))
rho = np.full((N, N), 0.5)
e.profile = True
n = 50
for it in range(n + 3):
    perm, u, z = make_draws(1, it, range(N), N, N * B)
    if it == 3:
        e.collect_timings()
    a, W, b, ll = e.sweep(a, W, b, rho, *hyp, perm, u, z, seed=1, sweep=it)
st = e.collect_timings()
print("N=%d B=%d T=%d, ms per sweep by stage:" % (N, B, T), {k: round(v["ms"] / n, 3) for k, v in st.items() if v["ms"] > 0}, "sum %.3f" % (sum(v["ms"] for k, v in st.items() if "." not in k) / n))
