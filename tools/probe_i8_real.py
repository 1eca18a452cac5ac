#!/usr/bin/env python
"""Times one group launch of the integer Gram (stats / planes / product / CRT) on the bench's own data (cfg3: basis-filtered Bernoulli(0.08)
spikes, omega ~ PG(1, psi)) -- residue planes of real data are not uniformly random bytes, and under the package power limit that
matters.  python tools/probe_i8_real.py [planes=13] [reps=6] [only=stage]
PGL_PROBE_LIB=ab loads lib/libpyglm_hip_ab.so (`make -C pyglm_amd/csrc ab`), whose tuning knobs read the environment (pgl_common.h)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import os                                           # noqa: E402
import pyglm_amd._lib as _l                         # noqa: E402
if os.environ.get("PGL_PROBE_LIB"):                 # "ab", or a variant built by hand next to it ("ab1": lib/libpyglm_hip_ab1.so)
    _l.LIB_PATH = _l.LIB_PATH.replace("libpyglm_hip.so", "libpyglm_hip_%s.so" % os.environ["PGL_PROBE_LIB"])
from pyglm_amd.engine import GibbsEngine            # noqa: E402
from pyglm_amd._lib import call, ptr                # noqa: E402
from pyglm_amd.utils.basis import cosine_basis      # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 13
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
only = sys.argv[3] if len(sys.argv) > 3 else None
N, B, T, nl = (int(os.environ.get(k, d)) for k, d in (('PN', 1024), ('PB', 5), ('PT', 100000), ('PNL', 8)))      # PN=128 PT=50000 PNL=64: BASELINE configs[1]
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
eng = GibbsEngine(N, B, 0, nl, batch=nl, gram="int8", planes=k)
ds = eng.add_data(Y, basis=cosine_basis(B, L=100) / 100)
eng._upload_weights(np.ones((nl, N), bool), rng.standard_normal((nl, N, B)) * 0.05, np.full(nl, -2.0))
with torch.cuda.device(eng.dev):
    eng._psi_pass(True, 3, 0)
    D, Dp, ldj = eng.D, eng.Dp, eng.ldj
    _, _, G, PB, R, stat = eng._i8_scratch[:6]
    om = ctypes.c_void_p(ds.OK.data_ptr())
    ldo = 2 * eng.ldn
    J = eng.Jslots[0]
    def stats():
        for c0 in range(0, nl, 8):          # (the statistics pass takes at most 8 weight columns)
            call("pgl_i8_colstats", ptr(ds.X), Dp, ctypes.c_void_p(om.value + 8 * c0), ldo, T, D, min(8, nl - c0), ptr(stat[0][c0:]), ptr(stat[1][c0:]), None)
        call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), nl * D, T, k, ptr(stat[2]), None)
    stages = [("stats", stats),
              ("planes X", lambda: call("pgl_i8_planes", ptr(ds.X), Dp, om, ldo, ptr(stat[2]), ptr(PB), T, D, nl, k, 0, None)),
              ("planes", lambda: call("pgl_i8_planes_t", ptr(ds.Xt), ds.Tp, om, ldo, ptr(stat[2]), ptr(PB), T, D, nl, k, 0, None)),
              ("gram", lambda: call("pgl_i8_gram", ptr(ds.PA), ptr(PB), ptr(R), T, D, nl, k, None)),
              ("crt", lambda: call("pgl_i8_crt", ptr(R), ptr(ds.sA), ptr(stat[2]), ptr(J), ldj, ldj * ldj, T, D, nl, k, 0, None))]
    for name, fn in stages:
        if only and name not in ("stats", "planes", only):
            continue
        if name == "gram" and os.environ.get("PZERO"):        # PZERO=1: the product on all-zero planes (same addresses, no switching in the multipliers)
            ds.PA.zero_(); PB.zero_()
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        extra = "   %.1f TOP/s algorithmic" % (k * nl * T * D * (D + 1.0) / ms * 1e-9) if name == "gram" else ""
        print("%-8s %9.3f ms%s" % (name, ms, extra), flush=True)
    zeros = float((PB[: ds.PA.numel()] == 0).float().mean())
    print("zero bytes in the first neuron's planes: %.3f" % zeros)
    print("checksum of the group's Gram matrices: %.17g" % float(torch.tril(J[:D, :D]).double().sum()))
