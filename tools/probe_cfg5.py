"""BASELINE.json configs[4] shape (N=4096, B=8, T=200000; D=32768) on ONE GPU for a few local neurons: one full engine sweep with
flips, stage timings, and the size-independent checks J mu = h / quadratic form of the draw.  PN/PT/PB/PNLOC override the shape."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
from pyglm_amd.utils.basis import cosine_basis

N, B, T, nloc = (int(os.environ.get(k, d)) for k, d in (("PN", 4096), ("PB", 8), ("PT", 200000), ("PNLOC", 2)))
rng = np.random.default_rng(0)
basis = cosine_basis(B, L=100) / 100
Y = (rng.random((T, N)) < 0.08).astype(np.float64)
t0 = time.perf_counter()
eng = GibbsEngine(N, B, 0, nloc, batch=nloc)
eng.add_data(Y, basis=basis)
torch.cuda.synchronize()
print("setup %.1f s, mem %.1f GB" % (time.perf_counter() - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
a = rng.random((nloc, N)) < 0.5
W = rng.standard_normal((nloc, N, B)) * 0.05 * a[:, :, None]
b = np.full(nloc, -2.0)
hyp = prior_terms(np.tile(np.eye(B) * float(os.environ.get('PSW', 1.0)), (nloc, N, 1, 1)), np.zeros((nloc, N, B)), np.ones(nloc), np.full(nloc, -2.0))
perm, u, z = make_draws(5, 0, range(nloc), N, N * B)
eng.profile = True
t0 = time.perf_counter()
a1, W1, b1, ll = eng.sweep(a, W, b, np.full((nloc, N), 0.5), *hyp, perm, u, z, seed=5, sweep=0)
dt = time.perf_counter() - t0
st = eng.collect_timings()
print("sweep %.1f s  stages(ms) %s" % (dt, {k: round(v["ms"], 1) for k, v in st.items()}), flush=True)
g = st["gram"]
print("gram %.1f TFLOP/s; active blocks %s -> %s" % (g["work"] / g["ms"] * 1e-9, a.sum(1).tolist(), a1.sum(1).tolist()))
D = N * B
for i in range(nloc):
    M = torch.tril(eng.Jbuf[i, :D + 2, :D + 2])
    m = torch.from_numpy(np.concatenate((np.repeat(a1[i], B), [True]))).cuda()
    idx = torch.nonzero(m)[:, 0]
    Js = M[:D + 1, :D + 1][idx][:, idx]
    Js = Js + torch.tril(Js, -1).t()
    h = M[D + 1, :D + 1][idx]
    k = int(m.sum().item())
    mu = torch.cholesky_solve(h[:, None], torch.linalg.cholesky(Js))[:, 0]          # independent fp64 posterior mean
    x = torch.from_numpy(np.concatenate((W1[i][a1[i]].ravel(), [b1[i]]))).cuda()
    d = x - mu
    q = (d @ (Js @ d)).item()
    zz = float(z[i, :k] @ z[i, :k])
    print("neuron %d: dim %d  (x-mu)'J(x-mu)/z'z - 1 = %.2e" % (i, k, q / zz - 1))
    assert abs(q / zz - 1) < 1e-6
    # the final tableau against the definition: M_SS = -(J_SS)^-1 on a probe vector, M_RS = J_RS J_SS^-1 on a few inactive rows
    Mt = eng.Mtab[i]
    v = torch.from_numpy(np.random.default_rng(3).standard_normal(k)).cuda()
    low = torch.tril(Mt[:D + 1, :D + 1])
    Ms = low[idx][:, idx]
    Ms = Ms + torch.tril(Ms, -1).t()
    e1 = (Js @ (Ms @ v) + v).abs().max().item() / v.abs().max().item()
    print("          |J_SS M_SS v + v| / |v| = %.2e" % e1)
    assert e1 < 1e-6
print("OK")
