"""The batched rank-k products of the flips and the Cholesky (pgl_contract_tn_batched): the update pipeline (pgl_update.hip: 256 x 128
tiles, DMA-staged, persistent, border rows through the skinny kernel) against the generic tiles -- bit for bit -- and both against a
torch fp64 product.  The reference has no counterpart of these products (it refactors the active block per proposal,
pyglm/regression.py:343-378); what they must reproduce is plain linear algebra, C = beta C + alpha A'B on the stated tile sets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(A, B, C0, M, N, K, tri, alpha, beta, kernel, batch_k=None, a_cols=None, b_cols=None):
    import torch
    from pyglm_amd._lib import call, ptr
    C = C0.clone()
    nb = A.shape[0]
    bk = None if batch_k is None else torch.tensor(batch_k, dtype=torch.int32, device=A.device)
    call("pgl_contract_tn_batched", ptr(A), A.shape[2], A.shape[1] * A.shape[2], a_cols or A.shape[2], ptr(B), B.shape[2], B.shape[1] * B.shape[2],
         b_cols or B.shape[2], ptr(C), C.shape[2], C.shape[1] * C.shape[2], M, N, K, nb, ptr(bk), float(alpha), float(beta), tri, kernel, None)
    torch.cuda.synchronize()
    return C


def _mask(M, N, tri, device):
    import torch
    r = torch.arange(M, device=device)[:, None]
    c = torch.arange(N, device=device)[None, :]
    if tri == 1:
        return c <= r
    if tri == 2:
        return c >= r
    return torch.ones(M, N, dtype=torch.bool, device=device)


@pytest.mark.parametrize("tri,M,N", [(1, 770, 770), (1, 1030, 1030), (2, 700, 700), (0, 512, 1040), (0, 130, 300), (1, 200, 200)])
@pytest.mark.parametrize("alpha,beta", [(-1.0, 1.0), (1.0, 1.0), (1.0, 0.0), (-0.5, 0.0)])
def test_pipeline_equals_generic_tiles_bit_for_bit(tri, M, N, alpha, beta):
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    nb, K = 5, 96
    ld = (max(M, N) + 15) // 16 * 16 + 16
    A = torch.randn(nb, K, ld, dtype=torch.float64, device=dev, generator=g)
    B = torch.randn(nb, K, ld, dtype=torch.float64, device=dev, generator=g)
    C0 = torch.randn(nb, max(M, N), ld, dtype=torch.float64, device=dev, generator=g)
    bk = [96, 48, 0, 16, 96]
    out = {k: _run(A, B, C0, M, N, K, tri, alpha, beta, k, batch_k=bk) for k in (0, 1, 2)}
    m = _mask(M, N, tri, dev)
    for b in range(nb):
        ref = beta * C0[b, :M, :N] + alpha * A[b, :bk[b], :M].T @ B[b, :bk[b], :N] if bk[b] else C0[b, :M, :N]
        for k in (0, 1, 2):
            got = out[k][b, :M, :N]
            # (a skipped batch leaves C alone, also with beta = 0: the flips rely on that)
            assert torch.equal(got[m], out[0][b, :M, :N][m]), "kernel %d differs from the generic tiles in batch %d" % (k, b)
            np.testing.assert_allclose(got[m].cpu().numpy(), ref[m].cpu().numpy(), rtol=0, atol=1e-11)
        # nothing outside the stated rows / columns is touched
        assert torch.equal(out[2][b, M:, :], C0[b, M:, :]) and torch.equal(out[2][b, :, N:], C0[b, :, N:])


def test_pipeline_on_the_tableau_shape_with_border_rows():
    """D + 2 rows with D a multiple of 256: the bias and potential rows go through the skinny kernel; rank 512 and a trailing square"""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    nb, D, K = 3, 1280, 512
    Md, ld = D + 2, D + 16
    W = torch.randn(nb, K, ld, dtype=torch.float64, device=dev, generator=g)
    U = torch.randn(nb, K, ld, dtype=torch.float64, device=dev, generator=g)
    C0 = torch.randn(nb, ld, ld, dtype=torch.float64, device=dev, generator=g)
    for r0 in (0, 320):
        from pyglm_amd._lib import call, ptr
        res = []
        for kernel in (0, 2):
            C = C0.clone()
            off = lambda t, e: t.data_ptr() + 8 * e
            import ctypes
            call("pgl_contract_tn_batched", ctypes.c_void_p(off(W, r0)), ld, K * ld, ld - r0, ctypes.c_void_p(off(U, r0)), ld, K * ld, ld - r0,
                 ctypes.c_void_p(off(C, r0 * ld + r0)), ld, ld * ld, Md - r0, Md - r0, K, nb, None, -1.0, 1.0, 1, kernel, None)
            torch.cuda.synchronize()
            res.append(C)
        m = _mask(Md - r0, Md - r0, 1, dev)
        for b in range(nb):
            a = res[0][b, r0:Md, r0:Md][m]
            assert torch.equal(a, res[1][b, r0:Md, r0:Md][m])
            ref = C0[b, r0:Md, r0:Md] - W[b, :, r0:Md].T @ U[b, :, r0:Md]
            np.testing.assert_allclose(a.cpu().numpy(), ref[m].cpu().numpy(), rtol=0, atol=2e-10)
            assert torch.equal(res[1][b, :r0, :], C0[b, :r0, :]) and torch.equal(res[1][b, Md:, :], C0[b, Md:, :])
