"""Stand-alone timing of the omega-weighted Gram kernel (for A/B of kernel variants and PMC runs)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyglm_amd._lib import call, ptr
T = int(os.environ.get("PT", 100000)); D = int(os.environ.get("PD", 1280)); nz = int(os.environ.get("PNZ", 64)); reps = int(os.environ.get("PREPS", 3))
Tp, Dp = (T + 15) // 16 * 16, (D + 1 + 15) // 16 * 16
ldj = (D + 2 + 15) // 16 * 16
torch.manual_seed(0)
X = torch.zeros(Tp, Dp, dtype=torch.float64, device="cuda"); X[:T, :D] = torch.rand(T, D, dtype=torch.float64, device="cuda") * 0.2
W = torch.zeros(Tp, nz, dtype=torch.float64, device="cuda"); W[:T] = torch.rand(T, nz, dtype=torch.float64, device="cuda") * 0.25
J = torch.zeros(nz, ldj, ldj, dtype=torch.float64, device="cuda")
flops = float(nz) * T * D * (D + 1)
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    call("pgl_weighted_gram", ptr(X), Dp, Dp, ptr(W), nz, Tp, D, nz, ptr(J), ldj, ldj * ldj, 0, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("gram T=%d D=%d nz=%d: %.4f s  %.2f TFLOP/s (algorithmic)" % (T, D, nz, dt, flops / dt * 1e-12), flush=True)
if os.environ.get("PCHECK"):
    ref = (X[:T, :D].t() * W[:T, 1]) @ X[:T, :D]
    got = torch.tril(J[1, :D, :D])
    print("max rel err", ((got - torch.tril(ref)).abs().max() / ref.abs().max()).item())
