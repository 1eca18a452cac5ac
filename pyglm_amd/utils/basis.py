"""Basis construction on the host (one-time, tiny) with the reference's names (pyglm/utils/basis.py).
The design matrix itself is built on the GPU at add_data (pgl_design_matrix); `convolve_with_basis` here is the
same routine exposed for users and goes through the device as well."""
import numpy as np


def cosine_basis(B, L=100, orth=False, norm=True, n_eye=0, a=1.0 / 120, b=0.5):
    """Raised-cosine bumps with log-warped centres (basis.py:61-106): the first n_eye columns are unit lags, the rest are
    0.5*(1+cos(clip((u-c)*pi/(2w)))) on u = log(a*t+b); `norm` rescales each column to sum to L."""
    n_cos = B - n_eye
    assert n_cos >= 0 and n_eye >= 0
    u = np.log(a * np.arange(L) + b)
    pick = np.floor(np.linspace(n_eye, L / 2.0, n_cos)).astype(int)
    centres = u[pick]
    if n_cos == 1:
        width = centres / 2
    else:
        width = (centres[-1] - centres[0]) / (n_cos - 1)
    basis = np.zeros((L, B))
    basis[:n_eye, :n_eye] = np.eye(n_eye)
    for k in range(n_cos):
        phase = np.minimum(np.pi, np.maximum(-np.pi, (u - centres[k]) * np.pi / width / 2.0))
        basis[:, n_eye + k] = 0.5 * (1.0 + np.cos(phase))
    if orth:
        import scipy.linalg
        return scipy.linalg.orth(basis)
    if norm:
        if (basis < 0).any():
            raise Exception("We can only normalize nonnegative impulse responses!")
        basis = basis / basis.sum(axis=0, keepdims=True) * L
    return basis


def interpolate_basis(basis, dt, dt_max, norm=True, allow_instantaneous=False):
    """Resample a basis defined on [0, dt_max] at resolution dt (basis.py:36-58)."""
    L, B = basis.shape
    t_new = np.arange(0.0, dt_max, step=dt)
    t_old = np.linspace(0.0, dt_max, L)
    out = np.column_stack([np.interp(t_new, t_old, basis[:, k]) for k in range(B)])
    if norm:
        out = out / (dt * out.sum(axis=0))
    if not allow_instantaneous:
        out = np.vstack((np.zeros((1, B)), out))
    return out


def convolve_with_basis(S, basis, device=None):
    """(T,N) counts -> (T,N,B) causally filtered regressors (basis.py:5-34), computed by pgl_design_matrix on the GPU."""
    from ..engine import GibbsEngine
    S = np.asarray(S, dtype=float)
    eng = GibbsEngine(S.shape[1], basis.shape[1], 0, 1, device=device, batch=1, design_only=True)
    eng.add_data(S, basis=basis)
    return eng.design_matrix()
