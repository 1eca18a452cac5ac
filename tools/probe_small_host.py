#!/usr/bin/env python
"""Where a sweep of a small model spends its host time (BASELINE configs[0]: N = 4, B = 1, T = 10 000): cProfile over resample_model."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
from pyglm_amd.models import SparseBernoulliGLM
from pyglm_amd.utils.basis import cosine_basis
N, B, T = (int(x) for x in (sys.argv[1:4] + ["4", "1", "10000"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
m = SparseBernoulliGLM(N, basis=cosine_basis(B, L=100) / 100, B=B)
m.add_data(Y)
for _ in range(5):
    m.resample_model()
t0 = time.perf_counter()
for _ in range(100):
    m.resample_model()
print("%.3f ms per resample_model" % ((time.perf_counter() - t0) * 10))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    m.resample_model()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
