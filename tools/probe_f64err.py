"""Measurement (GPU): error of the fp64-MFMA Gram kernel relative to |a_i||b_j|, with the integer path as the reference, on the
bench's own kind of data (basis-filtered Bernoulli(0.08) spikes, omega ~ PG(1, psi)).  Decides how many residue planes the integer
path needs to stay at or below the fp64 kernel's own error.   python tools/probe_f64err.py [T] [N]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyglm_amd.engine import GibbsEngine                      # noqa: E402
from pyglm_amd.utils.basis import cosine_basis                # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B, nl = 5, 8
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
basis = cosine_basis(B, L=100) / 100
res = {}
for gram in ("int8", "fp64"):
    eng = GibbsEngine(N, B, 0, nl, gram=gram, batch=nl)
    ds = eng.add_data(Y, basis=basis)
    a = np.ones((nl, N), bool)
    W = rng.standard_normal((nl, N, B)) * 0.1 if gram == "int8" else W
    b = np.full(nl, -2.0)
    eng._upload_weights(a, W, b)
    with torch.cuda.device(eng.dev):
        eng._psi_pass(True, 3, 0)
        eng._gram(0, nl, 0)
        torch.cuda.synchronize()
    D = N * B
    res[gram] = eng.Jslots[0][:nl, :D, :D].cpu().numpy()
    X = ds.X[:T, :D].cpu().numpy()
    Om = ds.OK[:T, :nl].cpu().numpy()
    del eng
    torch.cuda.empty_cache()
low = np.tril(np.ones((D, D), bool))
na = np.sqrt((X * X).sum(0))
print("T=%d D=%d  column max/rms of X: median %.2f max %.2f;  omega max/rms: %.2f  (max %.3f, min %.4f)" % (
    T, D, np.median(np.abs(X).max(0) / (na / np.sqrt(T))), (np.abs(X).max(0) / (na / np.sqrt(T))).max(),
    (Om.max(0) / np.sqrt((Om ** 2).mean(0))).max(), Om.max(), Om.min()))
for g in range(nl):
    nb = np.sqrt(((Om[:, g:g + 1] * X) ** 2).sum(0))
    den = np.outer(na, nb)
    e = (np.abs(res["fp64"][g] - res["int8"][g]) / den)[low]
    rel = (np.abs(res["fp64"][g] - res["int8"][g]) / np.abs(res["int8"][g]))[low]
    print("neuron %d: fp64-kernel error / |a||b|: max %.3e  rms %.3e  median %.3e ;  relative to |J_ij|: max %.3e rms %.3e; cos(a,b) median %.3f" % (
        g, e.max(), np.sqrt((e ** 2).mean()), np.median(e), rel.max(), np.sqrt((rel ** 2).mean()), np.median((np.abs(res["int8"][g]) / den)[low])))
