"""BASELINE configs[0] (N = 4, B = 1, L = 100, T = 10 000; examples/synthetic.py:61-68) is host-bound: where do the host's microseconds of
one resample_model() go?  cProfile over 300 sweeps after a warm-up, top entries by own time and by cumulative time
-> profiles/r06_cfg1_host_profile.txt.   Usage (GPU box): python tools/profile_cfg1_host.py > gpurun_out/r06_cfg1_host_profile.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from bench import synth
    from pyglm_amd.models import SparseBernoulliGLM
    N, B, T, L = 4, 1, 10000, 100
    np.random.seed(0)
    basis, Y = synth(N, B, T, L)
    model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=0)
    model.add_data(Y)
    for _ in range(50):
        model.resample_model()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        model.resample_model()
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / n * 1e3
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        model.resample_model()
    pr.disable()
    print("configs[0]: %.3f ms per resample_model() unprofiled (%d sweeps); cProfile of the same %d sweeps, times are totals over them" % (plain, n, n))
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(18)
        print("---- by %s\n%s" % (key, "\n".join(l for l in s.getvalue().splitlines()[4:] if l.strip())))


if __name__ == "__main__":
    main()
