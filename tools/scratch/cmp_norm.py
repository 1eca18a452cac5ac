"""the scales of omega_n X from the batch contraction (pgl_sweep_t.i8_norm) against those of the per-group column statistics, on the bench's data"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from bench import synth
from pyglm_amd.models import SparseBernoulliGLM
N, B, T = 256, 5, 30000
D = N * B
basis, Y = synth(N, B, T, 100)
res = []
for use in (True, False):
    np.random.seed(0)
    model = SparseBernoulliGLM(N, basis=basis, regression_kwargs=dict(S_w=10.0, mu_b=-2.0), seed=0, engine_kwargs=dict(gram="int8", batch=128))
    model.add_data(Y)
    eng = model.engine
    if not use:
        eng._i8_norm = None
    for _ in range(2):
        model.resample_model()
    torch.cuda.synchronize()
    sB = eng._i8_scratch[5][2].cpu().numpy().copy()          # scales of the last group
    J = eng.Jbuf[:, :D + 2, :D + 2].cpu().numpy().copy()
    res.append((sB, J, model.log_likelihood()))
    ds = eng.datasets[0]
    if use:
        part = (eng.nb + (eng.nb & 1)) * eng.Dp
        ommax = eng._i8_norm[part:part + N].cpu().numpy()
        xmax = ds.xmax.cpu().numpy()
        print("omega max: min %.3f max %.3f; xmax: min %.4f max %.4f" % (ommax.min(), ommax.max(), xmax[:D].min(), xmax[:D].max()))
        ss = eng._i8_norm[:part].reshape(-1, eng.Dp)[:8, :D].cpu().numpy()
        print("bound / norm of the last group's columns: max %.3f" % (ommax[-8:, None] * xmax[None, :D] / np.sqrt(ss[-8 + 0:][:8] if False else ss[:8])).max())
    del model, eng
(s1, J1, l1), (s0, J0, l0) = res
print("scales differing:", int((s1 != s0).sum()), "of", s1.size, "; ratio values", np.unique(s1[s1 != s0] / s0[s1 != s0]))
print("J entries differing:", int((np.tril(J1) != np.tril(J0)).sum()), "max rel", np.abs(J1 - J0).max() / np.abs(J0).max())
print("ll", l1, l0)
