"""The fused C entry of the boundary (include/pyglm_hip.h: pgl_sweep / pgl_get_state) driven WITHOUT pyglm_amd/engine.py: every buffer is
allocated here, the design matrix comes from pgl_design_matrix, and one call runs the golden model sweep captured from the reference
(fixture G9/G11: SparseBernoulliGLM N = 4, B = 2, T = 600 with recorded omega / permutations / uniforms / normals)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _prior_terms(S_w, mu_w, S_b, mu_b):
    Jw = np.linalg.inv(S_w)
    hw = np.einsum("nmij,nmj->nmi", Jw, mu_w)
    c0 = 0.5 * np.linalg.slogdet(Jw)[1] - 0.5 * np.einsum("nmi,nmi->nm", mu_w, hw)
    return Jw, hw, 1.0 / S_b, mu_b / S_b, c0


@pytest.mark.parametrize("batch,timed", [(4, False), (3, True)])
def test_golden_model_sweep_through_the_c_abi_alone(golden, batch, timed):
    import torch
    from pyglm_amd import _lib
    from pyglm_amd._lib import call, ptr
    g = golden
    lib = _lib.load()
    dev = torch.device("cuda:0")
    Y, basis = g["M_Y"], g["M_basis"]
    T, N = Y.shape
    B = basis.shape[1]
    D, nloc, nb = N * B, N, batch
    Dp, ldn, ldj = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    call("pgl_sweep_dims", N, B, nloc, ctypes.byref(Dp), ctypes.byref(ldn), ctypes.byref(ldj))
    Dp, ldn, ldj = Dp.value, ldn.value, ldj.value
    Tp = (T + 15) // 16 * 16
    kmax = lib.pgl_flip_kmax()

    def z(*shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype, device=dev)

    def up(x, dtype=np.float64):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=dtype)).to(dev)
    # data set: design matrix on the device (pyglm/utils/basis.py:5-34), spikes, outputs
    X, Xt = z(Tp, Dp), z(Dp, Tp)
    S = up(Y)
    bas = up(basis)
    call("pgl_design_matrix", ptr(S), N, ptr(bas), ptr(X), Dp, ptr(Xt), Tp, T, N, B, basis.shape[0], 1, None)
    torch.cuda.synchronize()
    np.testing.assert_allclose(X[:T, :D].cpu().numpy().reshape(T, N, B), g["M_X"], rtol=1e-10, atol=1e-13)
    Yd = z(T, ldn)
    Yd[:, :nloc] = S
    om = up(g["M_omegas"].T)                              # (T, nloc): the omega the reference consumed
    Psi, OK, llpart = z(T, ldn), z(Tp, 2 * ldn), z(lib.pgl_pg_loglik_partials(T), nloc)        # (named: raw pointers do not keep them alive)
    ds = (_lib.Dataset * 1)(_lib.Dataset(T, Tp, ptr(X), ptr(Xt), ptr(Yd), ptr(Psi), ptr(OK), ptr(llpart), 0, 0, 0, None, None, ptr(om)))
    # chain state and the sweep's inputs
    a, W, b = up(g["M_A0"], np.int32), up(g["M_W0"].reshape(N, D)), up(g["M_b0"])
    S_w = np.tile(10.0 * np.eye(B), (N, N, 1, 1))
    Jw, hw, Jb, hb, c0 = _prior_terms(S_w, np.zeros((N, N, B)), np.ones(N), np.full(N, -2.0))
    inp = dict(rho=up(np.full((N, N), 0.5)), Jw=up(Jw), hw=up(hw), Jb=up(Jb), hb=up(hb), c0=up(c0), perm=up(g["M_perms"], np.int32), u=up(g["M_us"]),
               z=up(g["M_zs"]))
    out = dict(ll=z(nloc), status=z(nloc, dtype=torch.int32), logodds=z(nloc, N))
    scr = dict(Wt=z(Dp, ldn), bias=z(nloc), border=z(2 * ldn, Dp), skip=z(nloc, dtype=torch.int32), Jbuf=z(nb, ldj, ldj), Mtab=z(nb, ldj, ldj),
               Ac=z(nb, ldj, ldj), hc=z(2, nb, ldj), Tinv=z(nb, 64, 64), G=z(nb, kmax, kmax), Lws=z(nb, (kmax + 1) ** 2), Ut=z(nb, kmax, ldj),
               Wt_ws=z(nb, kmax, ldj), d_idx=z(nb, kmax, dtype=torch.int32), d_sign=z(nb, kmax), d_cnt=z(nb, dtype=torch.int32),
               batch_k=z(nb, dtype=torch.int32), act=z(nb, D + 1, dtype=torch.int32), na=z(nb, dtype=torch.int32))
    times = _lib.StageTimes()
    sw = _lib.Sweep(N=N, B=B, n0=0, nloc=nloc, nb=nb, obs=0, xi=1.0, visit_order=1, planes=13, i8_group=0, datasets=ds, ndatasets=1,
                    a=ptr(a), W=ptr(W), b=ptr(b), **{k: ptr(v) for k, v in inp.items()}, **{k: ptr(v) for k, v in out.items()},
                    **{k: ptr(v) for k, v in scr.items()}, times=ctypes.pointer(times) if timed else None)
    call("pgl_sweep", ctypes.byref(sw), 5, 0, None)
    a1 = np.empty((nloc, N), dtype=np.int32)
    W1 = np.empty((nloc, N, B))
    b1, ll, status = np.empty(nloc), np.empty(nloc), np.empty(nloc, dtype=np.int32)
    call("pgl_get_state", ctypes.byref(sw), a1.ctypes.data, W1.ctypes.data, b1.ctypes.data, ll.ctypes.data, status.ctypes.data, None)
    assert not status.any()
    np.testing.assert_allclose(ll.sum(), g["M_ll0"], rtol=1e-11)                # log-likelihood of the state before the sweep
    np.testing.assert_array_equal(a1.astype(bool), g["M_A1"])                   # decisions: exact
    np.testing.assert_allclose(W1, g["M_W1"], rtol=1e-8, atol=1e-10)            # posteriors: << 1e-5 rel
    np.testing.assert_allclose(b1, g["M_b1"], rtol=1e-8, atol=1e-10)
    assert np.isfinite(out["logodds"].cpu().numpy()).all()                      # rho = 1/2 everywhere: every block was proposed
    if timed:
        call("pgl_stage_times_collect", ctypes.byref(times))
        names = [lib.pgl_stage_name(i).decode() for i in range(_lib.NSTAGES)]
        got = {n: (times.ms[i], times.calls[i]) for i, n in enumerate(names) if times.calls[i]}
        # batches of 3 + 1; a model this small takes flips + weight draw as ONE launch per batch (pgl_small.hip), timed under "flips"
        assert got["gram"][1] == 2 and got["flips"][1] == 2 and "weights" not in got and got["activation"][1] == 1
        assert all(ms >= 0.0 for ms, _ in got.values()) and times.pending is None
    # a second sweep continues the chain from the device-resident state, now with the library's own PG draws
    ds[0].omega_override = None
    call("pgl_sweep", ctypes.byref(sw), 5, 1, None)
    a2 = np.empty_like(a1)
    W2 = np.empty_like(W1)
    call("pgl_get_state", ctypes.byref(sw), a2.ctypes.data, W2.ctypes.data, None, None, status.ctypes.data, None)
    assert not status.any() and not np.array_equal(W2, W1) and np.all(W2[a2 == 0] == 0)


def test_native_binder_runs_sweeps_without_python():
    """examples/c_sweep/sweep_demo: C-style host code over include/pyglm_hip.h (hipMalloc'ed buffers, its own random inputs) runs a
    few sweeps of a small model through pgl_sweep / pgl_get_state in a process of its own -- no Python, no torch"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_sweep", "sweep_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["hipcc", "-O2", "-I", os.path.join(root, "include"), exe + ".cpp", "-L", os.path.join(root, "pyglm_amd", "lib"),
                               "-lpyglm_hip", "-Wl,-rpath,$ORIGIN/../../pyglm_amd/lib", "-o", exe])
    out = subprocess.run([exe, "24", "3", "4000", "6"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("sweep")]
    lls = [float(l.split("before")[1].split(",")[0]) for l in lines]
    assert len(lls) == 6 and out.stdout.strip().endswith("ok")
    assert max(lls[2:]) > lls[0]                       # the chain moves towards the data
