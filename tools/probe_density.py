import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth, CONFIGS
from pyglm_amd.models import SparseBernoulliGLM
cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
nsweep = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, B, T, L = cfg["N"], cfg["B"], cfg["T"], cfg["L"]
np.random.seed(0)
basis, Y = synth(N, B, T, L)
model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=0)
model.add_data(Y)
model.engine.profile = True
print("init density %.3f" % model.adjacency.mean())
for s in range(nsweep):
    A_old = model.adjacency.copy()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.resample_model()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    A = model.adjacency
    st = model.engine.collect_timings()
    fl = (A != A_old).sum(1)
    print("sweep %d: %.3f s  density %.4f  max row %d  flips/row mean %.1f max %d | %s" % (s, dt, A.mean(), A.sum(1).max(), fl.mean(), fl.max(),
          " ".join("%s=%.0fms" % (k, v["ms"]) for k, v in st.items())), flush=True)
