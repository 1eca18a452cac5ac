#!/usr/bin/env python
"""Graphics clock, socket power and memory-controller activity WHILE the fp64 tail of a sweep runs (flips + weight draw of a cfg3-shaped
batch, tools/probe_flipweights.py's workload): is the rank-512 pass at 0.82 of the fp64 MFMA peak short of the peak at 2.4 GHz, or of the
clock the package power limit leaves it?   python tools/probe_flip_clock.py [dens=0.44] [rho=0.7] [sweeps=4]"""
import sys, threading, time
import numpy as np, torch
sys.path.insert(0, ".")
from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
from pyglm_amd.utils.basis import cosine_basis
import amdsmi
dens = float(sys.argv[1]) if len(sys.argv) > 1 else 0.44
rho_v = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N, B, T, nloc = 1024, 5, 4096, 256
rng = np.random.default_rng(0)
Y = (rng.random((T, N)) < 0.08).astype(float)
eng = GibbsEngine(N, B, 0, nloc, gram="fp64")
eng.add_data(Y, basis=cosine_basis(B, L=100) / 100)
a = rng.random((nloc, N)) < dens
W = rng.standard_normal((nloc, N, B)) * 0.05 * a[:, :, None]
b = np.full(nloc, -2.0)
hyp = prior_terms(np.tile(np.eye(B) * 10.0, (nloc, N, 1, 1)), np.zeros((nloc, N, B)), np.ones(nloc), np.full(nloc, -2.0))
rho = np.full((nloc, N), rho_v)
amdsmi.amdsmi_init()
H = amdsmi.amdsmi_get_processor_handles()[0]
rows, stop = [], [False]
def loop():
    while not stop[0]:
        p = amdsmi.amdsmi_get_power_info(H); c = amdsmi.amdsmi_get_clock_info(H, amdsmi.AmdSmiClkType.GFX); u = amdsmi.amdsmi_get_gpu_activity(H)
        rows.append((time.perf_counter(), float(p.get("current_socket_power", 0)), float(c["clk"]), u.get("umc_activity")))
        time.sleep(0.02)
perm, u_, z = make_draws(1, 0, range(nloc), N, N * B)
eng.sweep(a, W, b, rho, *hyp, perm, u_, z, seed=1, sweep=0)
torch.cuda.synchronize()
th = threading.Thread(target=loop, daemon=True); th.start()
marks = []
for s in range(1, sweeps + 1):
    perm, u_, z = make_draws(1, s, range(nloc), N, N * B)
    eng.profile = True
    t0 = time.perf_counter()
    eng.sweep(a, W, b, rho, *hyp, perm, u_, z, seed=1, sweep=s)
    torch.cuda.synchronize()
    st = eng.collect_timings()
    marks.append((t0, time.perf_counter(), {k: round(v["ms"], 1) for k, v in st.items() if k in ("gram", "flips", "flips.init", "flips.apply", "weights")}))
stop[0] = True; th.join()
# a sweep here is: fp64 Gram (T = 4096: ~0.38 s) then flips (~0.5 s) then weights: the samples of the second half of each sweep are the tail
for t0, t1, st in marks:
    dur = t1 - t0
    g = st.get("gram", 0.0) * 1e-3
    tail = [r for r in rows if t0 + g + 0.03 <= r[0] <= t1 - 0.02]
    head = [r for r in rows if t0 + 0.03 <= r[0] <= t0 + g - 0.03]
    f = lambda rs, i: (float(np.mean([r[i] for r in rs])) if rs else float("nan"))
    print("sweep %.3f s %s | fp64 Gram: %.0f MHz %.0f W umc %.0f %% (%d samples) | flips+weights: %.0f MHz %.0f W umc %.0f %% (%d samples)"
          % (dur, st, f(head, 2), f(head, 1), f(head, 3), len(head), f(tail, 2), f(tail, 1), f(tail, 3), len(tail)))
