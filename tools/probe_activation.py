import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyglm_amd._lib import call, ptr
T, D, N = 100000, 5136, 1024
torch.manual_seed(0)
Xt = torch.rand(D, T, dtype=torch.float64, device="cuda")
for name, Wt in [("random", torch.randn(D, N, dtype=torch.float64, device="cuda")), ("zeros", torch.zeros(D, N, dtype=torch.float64, device="cuda"))]:
    Psi = torch.zeros(T, N, dtype=torch.float64, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        call("pgl_activation", ptr(Xt), T, ptr(Wt), N, ptr(Psi), N, T, D, N, None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(name, "activation %.3f s  %.1f TFLOP/s" % (dt, 2.0 * T * D * N / dt * 1e-12))
    ref = (Xt[:, :4096].t() @ Wt)
    print(name, "max abs err vs torch on first 4096 rows", (Psi[:4096] - ref).abs().max().item(), "ref scale", ref.abs().max().item())
