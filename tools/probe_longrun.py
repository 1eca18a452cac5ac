"""Longer chains at benchmark scale: every sweep must finish without a non-positive-definite flag and with a finite log-likelihood."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth, CONFIGS
from pyglm_amd.models import SparseBernoulliGLM
cfg = dict(CONFIGS[sys.argv[1]]); nsweep = int(sys.argv[2])
N, B, T, L = cfg["N"], cfg["B"], cfg["T"], cfg["L"]
np.random.seed(0)
basis, Y = synth(N, B, T, L)
model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=0)
model.add_data(Y)
t0 = time.perf_counter()
for s in range(nsweep):
    model.resample_model()
    if s % max(1, nsweep // 10) == 0 or s == nsweep - 1:
        ll = model.log_likelihood()
        A = model.adjacency
        assert np.isfinite(ll) and np.all(np.isfinite(model.weights))
        print("%s sweep %d: ll %.1f density %.3f |W|max %.2f  (%.1f s)" % (sys.argv[1], s, ll, A.mean(), np.abs(model.weights).max(), time.perf_counter() - t0), flush=True)
print("ok")
