#!/usr/bin/env python
"""Where a cfg3 sweep's wall time goes outside the device stages: resample_regressions (engine stages + host) vs resample_network (host only,
identical on every rank -- the part of a sweep that does not shrink with the number of GPUs).  Optional argv[1] = number of local neurons
(simulates the shard of an N-GPU run on one device)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from pyglm_amd.models import SparseBernoulliGLM


def main():
    N, B, T, L = 1024, 5, 100000, 100
    np.random.seed(0)
    basis, Y = bench.synth(N, B, T, L)
    model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=0)
    if len(sys.argv) > 1:
        print("note: shard emulation needs torch.distributed; run bench.py with PGL_BENCH_DEVICE for that")
    model.add_data(Y)
    torch.cuda.synchronize()
    model.resample_model()
    eng = model.engine
    for it in range(2):
        eng.profile = True
        eng.collect_timings()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.resample_regressions()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.resample_network()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        st = eng.collect_timings()
        dev = sum(v["ms"] for k, v in st.items() if k in ("activation", "pg_loglik", "gram", "gram.planes", "gram.int8", "gram.crt", "flips", "weights"))
        print("sweep %d: regressions %.1f ms (device stages %.1f, rest %.1f), network %.1f ms" % (it, (t1 - t0) * 1e3, dev, (t1 - t0) * 1e3 - dev, (t2 - t1) * 1e3), flush=True)


if __name__ == "__main__":
    main()
