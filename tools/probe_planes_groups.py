#!/usr/bin/env python
"""Does the plane conversion's store rate depend on how many planes a workgroup writes side by side?  The same 8 neurons converted as
1 x 8, 2 x 4, 4 x 2 and 8 x 1 calls (X is re-read per call: 4 GB of reads against 53 GB of stores)."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from pyglm_amd.engine import GibbsEngine
from pyglm_amd._lib import call, ptr, load
from pyglm_amd.utils.basis import cosine_basis
k, reps = 13, 8
N, B, T, nl = 1024, 5, 100000, 8
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
eng = GibbsEngine(N, B, 0, nl, batch=nl, gram="int8", planes=k)
ds = eng.add_data(Y, basis=cosine_basis(B, L=100) / 100)
eng._upload_weights(np.ones((nl, N), bool), rng.standard_normal((nl, N, B)) * 0.05, np.full(nl, -2.0))
with torch.cuda.device(eng.dev):
    eng._psi_pass(True, 3, 0)
    D, Dp = eng.D, eng.Dp
    _, _, G, PB, R, stat = eng._i8_scratch[:6]
    ldo = 2 * eng.ldn
    call("pgl_i8_colstats", ptr(ds.X), Dp, ptr(ds.OK), ldo, T, D, nl, ptr(stat[0]), ptr(stat[1]), None)
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), nl * D, T, k, ptr(stat[2]), None)
    lib = load()
    plane = lib.pgl_i8_padded_rows(D) * ((T + 63) // 64 * 64)
    ref = None
    for g in (8, 4, 2, 1):
        def run():
            for g0 in range(0, nl, g):
                call("pgl_i8_planes_t", ptr(ds.Xt), ds.Tp, ctypes.c_void_p(ds.OK.data_ptr() + 8 * g0), ldo,
                     ctypes.c_void_p(stat[2].data_ptr() + 8 * g0 * D), ctypes.c_void_p(PB.data_ptr() + g0 * k * plane), T, D, g, k, 0, None)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        chk = int(PB[: nl * k * plane].view(torch.int32)[::4097].to(torch.int64).sum())
        ref = chk if ref is None else ref
        print("%d call(s) of %d neurons: %.3f ms per 8 neurons  (checksum %s)" % (nl // g, g, e0.elapsed_time(e1) / reps, "same" if chk == ref else "DIFFERENT"), flush=True)
