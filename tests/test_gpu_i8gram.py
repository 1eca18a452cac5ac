"""The integer-MFMA Gram (pgl_i8_*: what gram='auto' runs at large shapes instead of pgl_weighted_gram) against NumPy integer arithmetic,
against the fp64 kernel and against an extended-precision reference."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODULI = [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197]
ELEM_BITS = 50


def _limit(k, T):
    """norm of the integer columns (pgl_i8gram.hip: pgl_k_i8_norm_limit): (limit (1 + 1e-9) + sqrt(T) + 1)^2 <= prod(p[:k]) / 2"""
    l2 = sum(np.log2(p) for p in MODULI[:k])
    return (2.0 ** ((l2 - 1.0) * 0.5) - np.sqrt(T) - 1.0) / (1.0 + 1e-9)


def _mix32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xffffffff
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x


def _dither(T, D, t0=0):
    """u(t, d) in [0, 1) of pgl_i8gram.hip (pgl_i8_dither): a fixed hash of the global time bin and the column, (T, D)"""
    t = (np.arange(T, dtype=np.uint64) + np.uint64(t0))[:, None]
    d = np.arange(D, dtype=np.uint64)[None, :]
    return _mix32((t + _mix32(d + 0x9e3779b9)) & 0xffffffff).astype(np.float64) / 4294967296.0


def _round_x(X, sA, t0=0):
    """the integers the planes of X hold: floor(v) + [frac(v) + u >= 1], v = x * scale (exact: the scale is a power of two)"""
    V = X * sA[None, :]
    fl = np.floor(V)
    return (fl + (((V - fl) + _dither(X.shape[0], X.shape[1], t0)) >= 1.0)).astype(np.int64)


def _nu(k, T):
    return int(np.floor(np.log2(_limit(k, T)) - 1e-12))


def _planes(P, n, Dq, Kp):
    """blocked plane layout [row / 16][K tile][row % 16][64 B] -> (n, Dq, Kp)"""
    return P.reshape(n, Dq // 16, Kp // 64, 16, 64).transpose(0, 1, 3, 2, 4).reshape(n, Dq, Kp)


def _data(T, D, G, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.random((T, D)) * (rng.random((T, D)) < 0.3) * 0.2
    X[:, 1] *= 1e-6                                           # a column on a very different scale
    X[:, 2] = 0.0                                             # an empty column
    if D > 6:
        X[T // 3, 5] = 3e7                                    # a column whose norm is one outlier element (1e8 x the rest)
    Om = 0.25 * rng.gamma(4.0, 0.25, size=(T, G))
    return X, Om


def _scales(Xd, Od, T, D, G, k):
    """column statistics + scales on the device (sA (D,), sB (G, D) as host arrays and device tensors), checked against their definition"""
    import torch
    from pyglm_amd._lib import call, ptr
    dev = Xd.device
    stat = torch.zeros(2, G, D, dtype=torch.float64, device=dev)
    sA = torch.zeros(D, dtype=torch.float64, device=dev)
    sB = torch.zeros(G, D, dtype=torch.float64, device=dev)
    call("pgl_i8_colstats", ptr(Xd), D, None, 0, T, D, 1, ptr(stat[0]), ptr(stat[1]), None)
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), D, T, k, ptr(sA), None)
    X = Xd.cpu().numpy()
    np.testing.assert_array_equal(stat[0, 0].cpu().numpy(), np.abs(X).max(0))
    np.testing.assert_allclose(stat[1, 0].cpu().numpy(), (X * X).sum(0), rtol=1e-11)
    call("pgl_i8_colstats", ptr(Xd), D, ptr(Od), G, T, D, G, ptr(stat[0]), ptr(stat[1]), None)
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), G * D, T, k, ptr(sB), None)
    torch.cuda.synchronize()
    Om = Od.cpu().numpy()
    lim, cap = _limit(k, T), 2.0 ** ELEM_BITS
    for g in range(G):
        V = Om[:, g:g + 1] * X
        np.testing.assert_array_equal(stat[0, g].cpu().numpy(), np.abs(V).max(0))
        np.testing.assert_allclose(stat[1, g].cpu().numpy(), (V * V).sum(0), rtol=1e-11)
    for V, sc in [(X, sA.cpu().numpy())] + [(Om[:, g:g + 1] * X, sB[g].cpu().numpy()) for g in range(G)]:
        amax, nrm = np.abs(V).max(0), np.sqrt((V * V).sum(0))
        live = amax > 0
        assert np.all(sc[~live] == 1.0)
        assert np.all(np.frexp(sc[live])[0] == 0.5)                                                 # powers of two: the scaling is exact
        assert np.all(amax[live] * sc[live] < cap) and np.all(nrm[live] * sc[live] <= lim)         # element bound, norm bound
        assert np.all((amax[live] * sc[live] * 2 >= cap) | (nrm[live] * sc[live] * 2 * (1 + 3e-12) > lim))                # and maximal
    return sA, sB


def test_norm_bits_and_minimum_planes():
    from pyglm_amd._lib import load
    lib = load()
    assert lib.pgl_i8_max_planes() == 15
    for k in range(8, 16):
        for T in (300, 100000, 5000000):
            assert lib.pgl_i8_norm_bits(k, T) == _nu(k, T) and abs(lib.pgl_i8_norm_limit(k, T) / _limit(k, T) - 1) < 1e-12
    assert [lib.pgl_i8_norm_bits(k, 100000) for k in (12, 13, 14, 15)] == [46, 50, 54, 58]
    assert lib.pgl_i8_min_planes(100000) == 13 and lib.pgl_i8_min_planes(50) == 13


@pytest.mark.parametrize("k", [13, 15])
def test_residue_planes_match_numpy(k):
    import torch
    from pyglm_amd._lib import call, ptr, load
    T, D, G = 300, 37, 2
    X, Om = _data(T, D, G)
    Xd, Od = torch.from_numpy(X).cuda(), torch.from_numpy(Om).cuda()
    lib = load()
    Dq, Kp = lib.pgl_i8_padded_rows(D), 320
    assert Dq in (256, 320) and lib.pgl_i8_plane_bytes(D, T) == 15 * Dq * Kp and lib.pgl_i8_residue_bytes(D) == 15 * Dq * Dq
    sA, sB = _scales(Xd, Od, T, D, G, k)
    PA = torch.full((k * Dq * Kp,), 77, dtype=torch.int8, device="cuda:0")
    PB = torch.full((G * k * Dq * Kp,), 77, dtype=torch.int8, device="cuda:0")
    call("pgl_i8_planes", ptr(Xd), D, None, 0, ptr(sA), ptr(PA), T, D, 1, k, 0, None)
    call("pgl_i8_planes", ptr(Xd), D, ptr(Od), G, ptr(sB), ptr(PB), T, D, G, k, 0, None)
    torch.cuda.synchronize()
    PA = _planes(PA.cpu().numpy(), k, Dq, Kp).astype(np.int64)
    PB = _planes(PB.cpu().numpy(), G * k, Dq, Kp).reshape(G, k, Dq, Kp).astype(np.int64)
    IA = _round_x(X, sA.cpu().numpy())                                                  # (T, D), |.| <= 2^50: X is rounded with the dither
    assert np.abs(IA).max() <= 2 ** ELEM_BITS
    V = X * sA.cpu().numpy()[None, :]
    assert np.all(np.abs(IA - V) < 1.0) and np.all(IA[V == np.floor(V)] == V[V == np.floor(V)])      # integers (zeros) stay exact
    up = (IA - np.floor(V))[(V != np.floor(V))]
    assert 0.4 < up.mean() < 0.6                                                         # ... and the rest goes up about half the time
    for q, p in enumerate(MODULI[:k]):
        got = PA[q, :D, :T]
        assert not ((got - IA.T) % p).any() and got.min() >= -128 and got.max() <= 127   # a signed-byte representative of the residue
        assert not PA[q, D:].any() and not PA[q, :, T:].any()                          # padding rows / time bins are zero
    for g in range(G):
        IB = np.rint((Om[:, g:g + 1] * X) * sB[g].cpu().numpy()[None, :]).astype(np.int64)
        for q in (0, 7, k - 1):
            p = MODULI[q]
            got = PB[g, q, :D, :T]
            assert not ((got - IB.T) % p).any() and got.min() >= -128 and got.max() <= 127


@pytest.mark.parametrize("T,D,G", [(300, 37, 2), (1300, 70, 8), (777, 16, 1)])
def test_planes_from_the_transposed_copy_are_the_same_bytes(T, D, G):
    """pgl_i8_planes_t (what pgl_sweep uses: coalesced rows of Xt) against pgl_i8_planes, ragged time / column counts, leading dimensions
    with padding"""
    import torch
    from pyglm_amd._lib import call, ptr, load
    k = 13
    X, Om = _data(T, D, G)
    Xd, Od = torch.from_numpy(X).cuda(), torch.from_numpy(Om).cuda()
    ldt = T + 5
    Xt = torch.zeros(D, ldt, dtype=torch.float64, device="cuda:0")
    Xt[:, :T] = Xd.t()
    Xt[:, T:] = 9.0                                                                     # never read
    lib = load()
    Dq, Kp = lib.pgl_i8_padded_rows(D), max(256, -(-T // 64) * 64)
    sA, sB = _scales(Xd, Od, T, D, G, k)
    out = []
    for name, Xa, ld in (("pgl_i8_planes", Xd, D), ("pgl_i8_planes_t", Xt, ldt)):
        PA = torch.full((k * Dq * Kp,), 77, dtype=torch.int8, device="cuda:0")
        PB = torch.full((G * k * Dq * Kp,), 77, dtype=torch.int8, device="cuda:0")
        call(name, ptr(Xa), ld, None, 0, ptr(sA), ptr(PA), T, D, 1, k, 0, None)
        call(name, ptr(Xa), ld, ptr(Od), G, ptr(sB), ptr(PB), T, D, G, k, 0, None)
        torch.cuda.synchronize()
        out.append((PA.cpu(), PB.cpu()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert out[0][1].float().abs().sum() > 0


@pytest.mark.parametrize("T,D,G,k", [(5000, 300, 3, 13), (20000, 520, 2, 13), (20000, 260, 2, 14), (140000, 40, 1, 13), (300, 1700, 1, 13),
                                     (5000, 300, 8, 15), (5000, 300, 2, 12)])
def test_integer_gram_matches_fp64_kernel_and_reference(T, D, G, k):
    import torch
    from pyglm_amd._lib import call, ptr, load
    X, Om = _data(T, D, G, seed=T)
    dev = "cuda:0"
    Xd, Od = torch.from_numpy(X).to(dev), torch.from_numpy(Om).to(dev)
    lib = load()
    sAd, sBd = _scales(Xd, Od, T, D, G, k)
    PA = torch.empty(lib.pgl_i8_plane_bytes(D, T) // 15 * k, dtype=torch.int8, device=dev)
    PB = torch.empty(G * lib.pgl_i8_plane_bytes(D, T) // 15 * k, dtype=torch.int8, device=dev)
    R = torch.empty(G * lib.pgl_i8_residue_bytes(D) // 15 * k, dtype=torch.int8, device=dev)
    ldj = (D + 2 + 15) // 16 * 16
    J = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    call("pgl_i8_planes", ptr(Xd), D, None, 0, ptr(sAd), ptr(PA), T, D, 1, k, 0, None)
    call("pgl_i8_planes", ptr(Xd), D, ptr(Od), G, ptr(sBd), ptr(PB), T, D, G, k, 0, None)
    call("pgl_i8_gram", ptr(PA), ptr(PB), ptr(R), T, D, G, k, None)
    call("pgl_i8_crt", ptr(R), ptr(sAd), ptr(sBd), ptr(J), ldj, ldj * ldj, T, D, G, k, 0, None)
    torch.cuda.synchronize()
    Ji = J.cpu().numpy()[:, :D, :D]
    sA, sB = sAd.cpu().numpy(), sBd.cpu().numpy()
    # (a) the exact integer answer: S = A'B on the scaled integers (Python ints), J = S / (sA sB)
    IA = _round_x(X, sA)
    lim = _limit(k, T)
    for g in (0, G - 1):
        IB = np.rint((Om[:, g:g + 1] * X) * sB[g][None, :]).astype(np.int64)
        # Cauchy-Schwarz keeps every entry inside the symmetric CRT range
        nA = np.sqrt((IA.astype(np.longdouble) ** 2).sum(0))
        nB = np.sqrt((IB.astype(np.longdouble) ** 2).sum(0))
        assert float(nA.max() * nB.max()) < 0.5 * float(np.prod([np.longdouble(p) for p in MODULI[:k]]))
        live = np.abs(X).max(0) > 0
        assert np.all(nA[live] >= 0.5 * min(lim, 2.0 ** ELEM_BITS) * (1 - 1e-6) - np.sqrt(T)) and np.all(nA <= lim * (1 + 1e-9) + np.sqrt(T))
        cols = [0, 1, 2, 5, D // 2, D - 1]
        S = IA.astype(object).T.dot(IB[:, cols].astype(object))                       # exact big-integer product, (D, len(cols))
        for c, j in enumerate(cols):
            rows = np.arange(j, D)
            want = np.array([float(S[i, c]) for i in rows]) / sA[rows] / sB[g, j]
            np.testing.assert_allclose(Ji[g, rows, j], want, rtol=2e-15, atol=0)     # CRT + Horner: a few ulp of the exact value
    # (b) against an extended-precision reference and the fp64 kernel: error relative to |a_i||b_j|
    Xl = X.astype(np.longdouble)
    Tp = (T + 15) // 16 * 16
    Xp = torch.zeros(Tp, (D + 1 + 15) // 16 * 16, dtype=torch.float64, device=dev)
    Xp[:T, :D] = Xd
    Wp = torch.zeros(Tp, G + (G & 1), dtype=torch.float64, device=dev)
    Wp[:T, :G] = Od
    Jn = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    call("pgl_weighted_gram", ptr(Xp), Xp.shape[1], Xp.shape[1], ptr(Wp), Wp.shape[1], Tp, D, G, ptr(Jn), ldj, ldj * ldj, 0, None)
    torch.cuda.synchronize()
    Jn = Jn.cpu().numpy()[:, :D, :D]
    low = np.tril(np.ones((D, D), dtype=bool))
    # standard deviation of the operand-rounding error relative to |a_i||b_j|: sqrt(|A_i|^-2 / 6 + |B_j|^-2 / 12) (dithered rounding of X:
    # variance f (1 - f), 1/6 on average; to nearest: 1/12); the smallest norms are half the element bound (2^49) or half the limit
    sigma = np.sqrt(1.0 / 6.0 + 1.0 / 12.0) / (0.5 * min(lim, 2.0 ** ELEM_BITS))
    for g in range(G if D <= 300 else 1):
        ref = np.asarray((Xl * Om[:, g].astype(np.longdouble)[:, None]).T @ Xl, dtype=np.longdouble)
        na = np.sqrt((X * X).sum(0))
        nb = np.sqrt(((Om[:, g:g + 1] * X) ** 2).sum(0))
        den = np.maximum(np.outer(na, nb), 1e-300)
        e_int = float(np.max((np.abs(Ji[g] - ref) / den)[low]))
        e_f64 = float(np.max((np.abs(Jn[g] - ref) / den)[low]))
        assert e_int < 6.5 * sigma + 3e-16, (e_int, sigma, e_f64)
        if k >= 13:      # the default number of moduli and above: at the level of the fp64 kernel's own error, or better
            assert e_int < 5e-15 and e_int < 3 * max(e_f64, 1.2e-15), (e_int, e_f64)
    # accumulate flag (second data set)
    call("pgl_i8_crt", ptr(R), ptr(sAd), ptr(sBd), ptr(J), ldj, ldj * ldj, T, D, G, k, 1, None)
    torch.cuda.synchronize()
    np.testing.assert_allclose(np.tril(J.cpu().numpy()[0, :D, :D]), 2 * np.tril(Ji[0]), rtol=1e-15)


def test_columns_of_few_distinct_values_round_independently():
    """a design matrix of filtered spikes: a narrow first basis function leaves ~1000 distinct values in a column of 200 000 bins
    (BASELINE configs[4]: B = 8, L = 100), and to-nearest rounding sends every occurrence of a value the same way -- the errors add
    coherently (measured before the dither: 1.0e-14 |a_i||b_j| at the worst entry, rms 3.6e-15).  With the dithered rounding of X the error
    is that of independent roundings; checked against an extended-precision reference."""
    import torch
    from pyglm_amd._lib import call, ptr, load
    from pyglm_amd.utils.basis import cosine_basis
    rng = np.random.default_rng(11)
    T, N, B, G, k = 200000, 5, 8, 2, 13
    D = N * B
    S = (rng.random((T, N)) < 0.02).astype(np.float64)
    basis = cosine_basis(B, L=100, norm=True)
    X = np.zeros((T, D))
    for n in range(N):
        for b in range(B):
            X[1:, n * B + b] = np.convolve(S[:, n], basis[:, b])[:T - 1]
    assert len(np.unique(X[:, 0])) < 3000
    Om = 0.25 * rng.gamma(4.0, 0.25, size=(T, G))
    dev = "cuda:0"
    Xd, Od = torch.from_numpy(X).to(dev), torch.from_numpy(Om).to(dev)
    lib = load()
    sAd, sBd = _scales(Xd, Od, T, D, G, k)
    PA = torch.empty(lib.pgl_i8_plane_bytes(D, T) // 15 * k, dtype=torch.int8, device=dev)
    PB = torch.empty(G * lib.pgl_i8_plane_bytes(D, T) // 15 * k, dtype=torch.int8, device=dev)
    R = torch.empty(G * lib.pgl_i8_residue_bytes(D) // 15 * k, dtype=torch.int8, device=dev)
    ldj = (D + 2 + 15) // 16 * 16
    J = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    call("pgl_i8_planes", ptr(Xd), D, None, 0, ptr(sAd), ptr(PA), T, D, 1, k, 0, None)
    call("pgl_i8_planes", ptr(Xd), D, ptr(Od), G, ptr(sBd), ptr(PB), T, D, G, k, 0, None)
    call("pgl_i8_gram", ptr(PA), ptr(PB), ptr(R), T, D, G, k, None)
    call("pgl_i8_crt", ptr(R), ptr(sAd), ptr(sBd), ptr(J), ldj, ldj * ldj, T, D, G, k, 0, None)
    torch.cuda.synchronize()
    Ji = J.cpu().numpy()[:, :D, :D]
    Xl = X.astype(np.longdouble)
    low = np.tril(np.ones((D, D), dtype=bool))
    na = np.sqrt((X * X).sum(0))
    for g in range(G):
        ref = np.asarray((Xl * Om[:, g].astype(np.longdouble)[:, None]).T @ Xl, dtype=np.longdouble)
        nb = np.sqrt(((Om[:, g:g + 1] * X) ** 2).sum(0))
        err = (np.abs(Ji[g] - ref) / np.outer(na, nb))[low].astype(np.float64)
        print("few distinct values, neuron %d: max %.3g rms %.3g of |a_i||b_j|" % (g, err.max(), np.sqrt((err ** 2).mean())))
        assert err.max() < 2.5e-15 and np.sqrt((err ** 2).mean()) < 6e-16


@pytest.mark.parametrize("T,S", [(1000, 320), (4096, 1024), (777, 256)])
def test_time_slices_of_x_hold_the_integers_of_the_whole(T, S):
    """the dither of X is keyed by the GLOBAL time bin (argument t0 of pgl_i8_planes[_t]): a data set converted slice by slice (planes of X
    not resident) holds the bytes of one conversion"""
    import torch
    from pyglm_amd._lib import call, ptr, load
    D, k = 37, 13
    X, _ = _data(T, D, 1, seed=2)
    dev = "cuda:0"
    Xd = torch.from_numpy(X).to(dev)
    Xt = Xd.t().contiguous()
    lib = load()
    Dq = lib.pgl_i8_padded_rows(D)
    stat = torch.zeros(2, D, dtype=torch.float64, device=dev)
    sA = torch.zeros(D, dtype=torch.float64, device=dev)
    call("pgl_i8_colstats", ptr(Xd), D, None, 0, T, D, 1, ptr(stat[0]), ptr(stat[1]), None)
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), D, T, k, ptr(sA), None)
    Kp = max(256, -(-T // 64) * 64)
    whole = torch.zeros(k * Dq * Kp, dtype=torch.int8, device=dev)
    call("pgl_i8_planes_t", ptr(Xt), T, None, 0, ptr(sA), ptr(whole), T, D, 1, k, 0, None)
    whole = _planes(whole.cpu().numpy(), k, Dq, Kp)
    IA = _round_x(X, sA.cpu().numpy())
    assert not ((whole[0, :D, :T].astype(np.int64) - IA.T) % 256).any()
    for name, src, ld, step in (("pgl_i8_planes_t", Xt, T, 8), ("pgl_i8_planes", Xd, D, 8 * D)):
        for t0 in range(0, T, S):
            ts = min(S, T - t0)
            Ks = max(256, -(-ts // 64) * 64)
            part = torch.zeros(k * Dq * Ks, dtype=torch.int8, device=dev)
            call(name, ctypes.c_void_p(src.data_ptr() + step * t0), ld, None, 0, ptr(sA), ptr(part), ts, D, 1, k, t0, None)
            torch.cuda.synchronize()
            part = _planes(part.cpu().numpy(), k, Dq, Ks)
            np.testing.assert_array_equal(part[:, :, :ts], whole[:, :, t0:t0 + ts])
            if t0 > 0 and name == "pgl_i8_planes_t":          # ... and NOT those of a conversion that forgot where the slice starts
                other = torch.zeros(k * Dq * Ks, dtype=torch.int8, device=dev)
                call(name, ctypes.c_void_p(src.data_ptr() + step * t0), ld, None, 0, ptr(sA), ptr(other), ts, D, 1, k, 0, None)
                torch.cuda.synchronize()
                assert (_planes(other.cpu().numpy(), k, Dq, Ks)[:, :, :ts] != whole[:, :, t0:t0 + ts]).any()


def test_heavy_tailed_columns_keep_the_error_at_the_fp64_level():
    """the guard of gram='auto' is structural: scales come from column NORMS, so a column with one element 1e8 x its rms, or a neuron
    whose omega has a spike, costs no precision relative to |a_i||b_j| -- checked against an extended-precision reference, next to the
    fp64 kernel, through the engine's own group path (the default number of planes)"""
    import torch
    from pyglm_amd.engine import GibbsEngine
    from pyglm_amd._lib import call, ptr
    rng = np.random.default_rng(3)
    N, B, T = 52, 5, 6000
    D = N * B
    X = rng.random((T, N, B)) * (rng.random((T, N, B)) < 0.3) * 0.2
    X[1234, 3, 1] = 2e7                                          # one bin 1e8 x the column's rms
    X[:, 7, 0] *= 1e-9
    X[77, 7, 0] = 5.0                                            # a tiny column with a huge outlier
    Om = 0.25 * rng.gamma(4.0, 0.25, size=(T, 4))
    Om[4321, 1] = 4e6                                            # a spike in one neuron's omega
    eng = GibbsEngine(N, B, 0, 4, batch=4, gram="int8")
    ds = eng.add_data((rng.random((T, N)) < 0.1).astype(float), X=X)
    assert eng.planes is None and ds.planes == 13
    W = torch.zeros(ds.Tp, 4, dtype=torch.float64, device="cuda")
    W[:T] = torch.from_numpy(Om).cuda()
    J8 = torch.zeros(4, eng.ldj, eng.ldj, dtype=torch.float64, device="cuda")
    J64 = torch.zeros(4, eng.ldj, eng.ldj, dtype=torch.float64, device="cuda")
    eng._i8_group(ds, ptr(W), 4, 4, ptr(J8), 0)
    call("pgl_weighted_gram", ptr(ds.X), eng.Dp, eng.Dp, ptr(W), 4, ds.Tp, D, 4, ptr(J64), eng.ldj, eng.ldj * eng.ldj, 0, None)
    torch.cuda.synchronize()
    Xf = X.reshape(T, D)
    Xl = Xf.astype(np.longdouble)
    low = np.tril(np.ones((D, D), dtype=bool))
    for g in range(4):
        ref = np.asarray((Xl * Om[:, g].astype(np.longdouble)[:, None]).T @ Xl, dtype=np.longdouble)
        den = np.outer(np.sqrt((Xf * Xf).sum(0)), np.sqrt(((Om[:, g:g + 1] * Xf) ** 2).sum(0)))
        e_int = float(np.max((np.abs(J8[g, :D, :D].cpu().numpy() - ref) / den)[low]))
        e_f64 = float(np.max((np.abs(J64[g, :D, :D].cpu().numpy() - ref) / den)[low]))
        assert e_int < 5e-15 and e_int < 3 * max(e_f64, 1.2e-15), (g, e_int, e_f64)


@pytest.mark.parametrize("T", [100000, 20000])
def test_default_planes_against_the_fp64_kernels_own_error_on_bench_data(T):
    """the default number of moduli (13) where it is used: on the bench's kind of data -- basis-filtered Bernoulli spikes, whose columns
    take few distinct values (roundings of repeated values are NOT independent: the error is several times the random-rounding model),
    omega ~ PG(1, psi) -- the integer Gram must not be less accurate than the fp64 MFMA kernel it replaces; both are measured against the
    15-plane integer Gram (error ~1e-18), relative to |a_i||b_j|.  12 planes fail this (measured 1.45e-14 rms against 4.4e-15 at T = 1e5)."""
    import torch
    from pyglm_amd.engine import GibbsEngine
    from pyglm_amd.utils.basis import cosine_basis
    N, B, nl = 64, 5, 4
    rng = np.random.default_rng(T)
    Y = (rng.random((T, N)) < 0.08).astype(float)
    W = rng.standard_normal((nl, N, B)) * 0.1
    res = {}
    for name, kw in (("default", dict(gram="int8")), ("k15", dict(gram="int8", planes=15)), ("fp64", dict(gram="fp64"))):
        eng = GibbsEngine(N, B, 0, nl, batch=nl, **kw)
        ds = eng.add_data(Y, basis=cosine_basis(B, L=100) / 100)
        if name == "default":
            assert ds.planes == 13
        eng._upload_weights(np.ones((nl, N), bool), W, np.full(nl, -2.0))
        with torch.cuda.device(eng.dev):
            eng._psi_pass(True, 3, 0)
            eng._gram(0, nl, 0)
            torch.cuda.synchronize()
        D = N * B
        res[name] = eng.Jslots[0][:nl, :D, :D].cpu().numpy()
        X, Om = ds.X[:T, :D].cpu().numpy(), ds.OK[:T, :nl].cpu().numpy()
        del eng, ds
        torch.cuda.empty_cache()
    low = np.tril(np.ones((D, D), bool))
    na = np.sqrt((X * X).sum(0))
    rms = lambda e: float(np.sqrt((e ** 2).mean()))
    out = []
    for g in range(nl):
        den = np.outer(na, np.sqrt(((Om[:, g:g + 1] * X) ** 2).sum(0)))
        e_int = (np.abs(res["default"][g] - res["k15"][g]) / den)[low]
        e_f64 = (np.abs(res["fp64"][g] - res["k15"][g]) / den)[low]
        out.append((rms(e_int), float(e_int.max()), rms(e_f64), float(e_f64.max())))
    print("T = %d: integer Gram (13 planes) rms / max, fp64 kernel rms / max relative to |a||b|:" % T, out)
    for r_int, m_int, r_f64, m_f64 in out:
        assert r_int <= r_f64 and m_int <= m_f64, out


def test_dyadic_data_is_exact():
    """spike counts through an identity basis (the reference's default basis, models.py:14-17) are small integers: with power-of-two
    scales the integer operands are exact -- no operand rounding at all for dyadic weights -- and every entry of the Gram comes back to
    the last few ulp of ITS OWN value (1e-15 relative: the fp64 Horner evaluation of the ~100-bit mixed-radix integer), where the
    operand-rounding model only promises 2e-16 |a_i||b_j| -- orders of magnitude more for these nearly orthogonal columns"""
    import torch
    from pyglm_amd.engine import GibbsEngine
    from pyglm_amd._lib import ptr
    rng = np.random.default_rng(8)
    N, B, T = 64, 5, 4000
    X = (rng.random((T, N, B)) < 0.1).astype(float) * rng.integers(1, 4, size=(T, N, B))       # counts 0..3
    eng = GibbsEngine(N, B, 0, 2, batch=2, gram="int8")
    ds = eng.add_data((rng.random((T, N)) < 0.1).astype(float), X=X)
    W = torch.zeros(ds.Tp, 2, dtype=torch.float64, device="cuda")
    W[:T, 0] = 1.0
    W[:T, 1] = 0.25
    J = torch.zeros(2, eng.ldj, eng.ldj, dtype=torch.float64, device="cuda")
    eng._i8_group(ds, ptr(W), 2, 2, ptr(J), 0)
    torch.cuda.synchronize()
    Xf = X.reshape(T, -1)
    G = Xf.T @ Xf                                       # integers below 2^53: exact in fp64
    D = N * B
    low = np.tril(np.ones((D, D), bool))
    np.testing.assert_allclose(J[0, :D, :D].cpu().numpy()[low], G[low], rtol=1e-15, atol=0)
    np.testing.assert_allclose(J[1, :D, :D].cpu().numpy()[low], 0.25 * G[low], rtol=1e-15, atol=0)


@pytest.mark.parametrize("N,B,T,batch", [(60, 3, 1500, 16), (110, 4, 2500, None)])
def test_sweep_with_integer_gram_equals_fp64_sweep_and_oracle(N, B, T, batch):
    """a full engine sweep with gram='int8' against the default fp64 Gram (same decisions, weights to 1e-9) and against the oracle"""
    from oracle import pyglm_oracle as orc
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    from tests.test_gpu_parity import _random_problem
    basis, X, Y, rng = _random_problem(N, B, T, seed=N + 1)
    kw = dict(rho=0.5, S_w=4.0, mu_w=0.0, mu_b=-1.5, S_b=2.0)
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * a[:, :, None]
    b = rng.standard_normal(N) - 1.5
    regs = [orc.Regression(N, B, **kw) for _ in range(N)]
    hyp = prior_terms(np.array([r.S_w for r in regs]), np.array([r.mu_w for r in regs]), np.array([r.S_b[0, 0] for r in regs]),
                      np.array([r.mu_b[0] for r in regs]))
    rho = np.array([r.rho for r in regs])
    perm, u, z = make_draws(41, 3, range(N), N, N * B)
    outs, Js = [], []
    for gram in ("int8", "fp64"):
        eng = GibbsEngine(N, B, batch=batch, gram=gram)
        eng.add_data(Y[: T // 2], X=X[: T // 2])
        eng.add_data(Y[T // 2:], X=X[T // 2:])                  # two data sets: the second Gram accumulates
        outs.append(eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=41, sweep=3)[:3])
        Js.append(eng.posterior(0))
    np.testing.assert_allclose(Js[0][0], Js[1][0], rtol=0, atol=1e-13 * np.abs(Js[1][0]).max())
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-9, atol=1e-11)
    omegas = [eng.datasets[k].OK[: eng.datasets[k].T, :N].cpu().numpy() for k in range(2)]
    for n in (0, N // 2, N - 1):
        r = orc.Regression(N, B, **kw)
        r.a, r.W, r.b = a[n].copy(), W[n].copy(), b[n:n + 1].copy()
        r.resample([(X[: T // 2], Y[: T // 2, n]), (X[T // 2:], Y[T // 2:, n])], [omegas[0][:, n], omegas[1][:, n]], perm[n], u[n], z[n])
        np.testing.assert_array_equal(outs[0][0][n], r.a)
        np.testing.assert_allclose(outs[0][1][n], r.W, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("resident", [True, False])
def test_time_slices_add_up_to_the_same_bits(resident):
    """BASELINE configs[4] cannot hold a neuron's planes at once: the integer Gram then runs in time slices whose products add up in the
    residues (pgl_i8_gram_slice), with X's planes either resident or converted per slice.  The arithmetic is exact, so J and the whole
    sweep must come out bit for bit as without slices -- including a last slice that is shorter and not a multiple of 64."""
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    rng = np.random.default_rng(11)
    N, B, T = 70, 5, 2300
    D = N * B
    Y = (rng.random((T, N)) < 0.1).astype(float)
    X = rng.random((T, N, B)) * (rng.random((T, N, B)) < 0.3) * 0.2
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.1 * a[:, :, None]
    b = np.full(N, -2.0)
    hyp = prior_terms(np.tile(np.eye(B) * 4.0, (N, N, 1, 1)), np.zeros((N, N, B)), np.ones(N), np.full(N, -2.0))
    rho = np.full((N, N), 0.5)
    perm, u, z = make_draws(5, 0, range(N), N, D)
    res = []
    for slc in (None, 0, 640, 1024):          # 0: one slice, but X's planes (resident = False) converted per group instead of kept
        kw = {} if slc is None else dict(i8_slice=slc or None, i8_resident=resident)
        eng = GibbsEngine(N, B, gram="int8", batch=N, **kw)
        ds = eng.add_data(Y, X=X)
        assert ds.int8 and (eng._i8_scratch[6] == (slc or 0)) and ((ds.PA is None) == (slc is not None and not resident))
        assert (eng._i8_scratch[7] is not None) == (ds.PA is None)
        out = eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=5, sweep=0)
        res.append((eng.Jbuf[:, :D + 2, :D + 2].cpu().numpy().copy(), out))
        del eng
    for J, out in res[1:]:
        np.testing.assert_array_equal(np.tril(J), np.tril(res[0][0]))
        for x, y in zip(out, res[0][1]):
            np.testing.assert_array_equal(x, y)


def _sweep_problem(N, B, T, seed):
    from pyglm_amd.engine import make_draws, prior_terms
    rng = np.random.default_rng(seed)
    D = N * B
    Y = (rng.random((T, N)) < 0.1).astype(float)
    X = rng.random((T, N, B)) * (rng.random((T, N, B)) < 0.3) * 0.2
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.1 * a[:, :, None]
    b = np.full(N, -2.0)
    hyp = prior_terms(np.tile(np.eye(B) * 4.0, (N, N, 1, 1)), np.zeros((N, N, B)), np.ones(N), np.full(N, -2.0))
    return Y, X, (a, W, b, np.full((N, N), 0.5)) + hyp + make_draws(5, 0, range(N), N, D)


@pytest.mark.parametrize("N,B,T", [(70, 5, 2300), (150, 5, 2100)])
def test_group_sizes_give_the_same_bits(N, B, T):
    """Where a plane has only a few tiles the product kernel takes more than 8 neurons per launch (16 .. 64: per-XCD lists of several neurons,
    the last plane of each in K quarters; BASELINE configs[1] runs 64).  Exact integer arithmetic: J and the whole sweep must not depend on
    the group size -- multiples of 8, a ragged last group (flat work list), a group that is no multiple of 8."""
    from pyglm_amd.engine import GibbsEngine
    Y, X, args = _sweep_problem(N, B, T, seed=13)
    D = N * B
    res = []
    for G in (8, 16, 32, 64, 5):
        eng = GibbsEngine(N, B, gram="int8", batch=N, i8_group=G)
        ds = eng.add_data(Y, X=X)
        assert ds.int8 and eng._i8_scratch[2] == G and (eng._i8_scratch[8] is not None) == (G % 8 == 0)
        out = eng.sweep(*args, seed=5, sweep=0)
        res.append((eng.Jbuf[:, :D + 2, :D + 2].cpu().numpy().copy(), out))
        del eng
    for J, out in res[1:]:
        np.testing.assert_array_equal(np.tril(J), np.tril(res[0][0]))
        for x, y in zip(out, res[0][1]):
            np.testing.assert_array_equal(x, y)


def test_group_size_follows_the_item_count():
    """_i8_plan: 8 neurons per launch where a plane has many tiles (one neuron per XCD), 64 where it has three (D = 640: 312 items of 8 neurons
    would leave the second round half empty, 64 neurons run 9.75 full rounds)"""
    from pyglm_amd.engine import GibbsEngine
    r = GibbsEngine._i8_rounds
    assert r(8, 136, 13) == 55.25 and r(8, 3, 13) == 2.0 and r(64, 3, 13) == 9.75 and r(32, 3, 13) == 5.0 and r(5, 3, 13) == 1.0
    rng = np.random.default_rng(0)
    for N, B, want in [(128, 5, 64), (420, 5, 8)]:
        T = 17000
        eng = GibbsEngine(N, B)
        ds = eng.add_data((rng.random((T, N)) < 0.1).astype(float), X=rng.random((T, N, B)) * 0.1)
        assert ds.int8 and eng._i8_scratch[2] == want, (N, eng._i8_scratch[2])
        del eng


def test_data_sets_planned_with_different_slicing_share_the_shorter_slice():
    """one slice length per engine (pgl_sweep_t.i8_slice): a second data set whose planes only fit in slices makes the first one run in
    slices too (and the other way round a data set no longer than the slice runs whole) -- slices add up exactly, so nothing changes in
    the result.  (This combination used to raise.)"""
    from pyglm_amd.engine import GibbsEngine
    N, B, T = 70, 5, 2300
    D = N * B
    Y, X, args = _sweep_problem(N, B, T, seed=17)
    res = []
    for second_slice in (None, 640):
        eng = GibbsEngine(N, B, gram="int8", batch=N)
        d0 = eng.add_data(Y[:1500], X=X[:1500])
        assert d0.int8 and eng._i8_scratch[6] == 0
        eng._i8_over["slice"] = second_slice             # the second data set is planned as if its planes only fitted 640 bins at a time
        d1 = eng.add_data(Y[1500:], X=X[1500:])
        assert d1.int8 and eng._i8_scratch[6] == (second_slice or 0)
        out = eng.sweep(*args, seed=5, sweep=0)
        res.append((eng.Jbuf[:, :D + 2, :D + 2].cpu().numpy().copy(), out))
        del eng
    np.testing.assert_array_equal(np.tril(res[0][0]), np.tril(res[1][0]))
    for x, y in zip(res[0][1], res[1][1]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("N,B,T,batch", [(70, 5, 2300, 70), (70, 5, 9000, 22), (33, 4, 2500, 7)])
def test_batch_norms_give_the_scales_of_the_column_statistics(N, B, T, batch):
    """Inside the sweep the norms of the columns of omega_n X come from one contraction of the squared operands per batch and the largest
    element from the bound max omega * max |x| (pgl_sweep_t.i8_norm) instead of a pass over X per group of 8 neurons.  Scales are powers of
    two, so on ordinary data both routes give the same scales and with them the same J and the same sweep, bit for bit; a column that a
    single element dominates may get a smaller scale (never a larger one) -- then J still agrees to the integer path's accuracy.
    Batches of odd size and a ragged last batch and group included."""
    from pyglm_amd.engine import GibbsEngine, make_draws, prior_terms
    rng = np.random.default_rng(12)
    D = N * B
    Y = (rng.random((T, N)) < 0.1).astype(float)
    X = rng.random((T, N, B)) * (rng.random((T, N, B)) < 0.3) * 0.2
    X[7, 3, 1] = 40.0                         # one element that carries its whole column: the bound binds there
    a = rng.random((N, N)) < 0.5
    W = rng.standard_normal((N, N, B)) * 0.1 * a[:, :, None]
    b = np.full(N, -2.0)
    hyp = prior_terms(np.tile(np.eye(B) * 4.0, (N, N, 1, 1)), np.zeros((N, N, B)), np.ones(N), np.full(N, -2.0))
    rho = np.full((N, N), 0.5)
    perm, u, z = make_draws(5, 0, range(N), N, D)
    eng = GibbsEngine(N, B, gram="int8", batch=batch)
    ds = eng.add_data(Y, X=X)
    assert ds.int8 and ds.xmax is not None and eng._i8_norm is not None
    res = []
    for batch_norms in (True, False):
        keep = eng._i8_norm
        if not batch_norms:
            eng._i8_norm = None               # -> pgl_sweep_t.i8_norm = NULL: pgl_i8_colstats per group
        out = eng.sweep(a, W, b, rho, *hyp, perm, u, z, seed=5, sweep=0)
        eng._i8_norm = keep
        res.append((eng.Jbuf[:, :D + 2, :D + 2].cpu().numpy().copy(), out))
    (J1, out1), (J0, out0) = res                # J of the LAST batch of each run
    d = 3 * B + 1                               # the dominated column
    other = np.ones(D + 2, dtype=bool)
    other[d] = False
    m = np.ix_(range(J1.shape[0]), other, other)
    np.testing.assert_array_equal(np.tril(J1[m]), np.tril(J0[m]))
    np.testing.assert_allclose(J1, J0, rtol=0, atol=1e-12 * np.abs(J0).max())
    np.testing.assert_array_equal(out1[0], out0[0])                          # the same adjacency
    for x, y in zip(out1[1:], out0[1:]):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-9)


def test_auto_takes_the_integer_gram_where_it_pays():
    """gram='auto' (the default): int8 planes for large D and long T, the fp64 kernel for short data sets and for the Gaussian model; below
    1024 columns a cost model from measured rates decides (GibbsEngine._i8_pays: the 320-tile padding, the item count of a launch)"""
    from pyglm_amd.engine import GibbsEngine
    rng = np.random.default_rng(0)
    for N, B, T, obs, nloc, want in [(210, 5, 2100, "bernoulli", 4, True), (210, 5, 1500, "bernoulli", 4, False), (60, 3, 4000, "bernoulli", 4, False),
                                     (210, 5, 2100, "gaussian", 4, False),
                                     # below 1024 columns, whole models: D = 640 and 650 (padded to 960) and 320 pay, D = 500 (padded to 640) is a tie
                                     # and stays on the fp64 kernel, a short data set does too; a shard of two neurons of the D = 640 model takes the path
                                     # of the whole model (the two Gram paths differ in the last bits: 1 GPU and 8 must agree)
                                     (128, 5, 17000, "bernoulli", None, True), (130, 5, 17000, "bernoulli", None, True), (64, 5, 17000, "bernoulli", None, True),
                                     (100, 5, 17000, "bernoulli", None, False), (128, 5, 9000, "bernoulli", None, False), (128, 5, 17000, "bernoulli", 2, True)]:
        eng = GibbsEngine(N, B, n1=nloc, obs=obs, batch=nloc)
        ds = eng.add_data((rng.random((T, N)) < 0.1).astype(float), X=rng.random((T, N, B)) * 0.1)
        assert eng.gram == "auto" and ds.int8 == want, (N, B, T, obs, nloc)
        assert (eng._i8_scratch is not None) == want
        del eng


def test_non_finite_weights_give_nan_not_garbage():
    """a NaN / inf in a neuron's omega (a diverged chain) must surface as NaN in that neuron's Gram, as it does on the fp64 kernel"""
    import torch
    from pyglm_amd._lib import call, ptr, load
    T, D, G, k = 700, 40, 3, 13
    X, Om = _data(T, D, G, seed=5)
    dev = "cuda:0"
    Xd, Od = torch.from_numpy(X).to(dev), torch.from_numpy(Om).to(dev)
    Od[123, 1] = float("nan")
    Od[55, 2] = float("inf")
    lib = load()
    stat = torch.zeros(2, G, D, dtype=torch.float64, device=dev)
    sA = torch.zeros(D, dtype=torch.float64, device=dev)
    sB = torch.zeros(G, D, dtype=torch.float64, device=dev)
    call("pgl_i8_colstats", ptr(Xd), D, None, 0, T, D, 1, ptr(stat[0]), ptr(stat[1]), None)
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), D, T, k, ptr(sA), None)
    call("pgl_i8_colstats", ptr(Xd), D, ptr(Od), G, T, D, G, ptr(stat[0]), ptr(stat[1]), None)
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), G * D, T, k, ptr(sB), None)
    PA = torch.empty(lib.pgl_i8_plane_bytes(D, T), dtype=torch.int8, device=dev)
    PB = torch.empty(G * lib.pgl_i8_plane_bytes(D, T), dtype=torch.int8, device=dev)
    R = torch.empty(G * lib.pgl_i8_residue_bytes(D), dtype=torch.int8, device=dev)
    ldj = (D + 2 + 15) // 16 * 16
    J = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    call("pgl_i8_planes", ptr(Xd), D, None, 0, ptr(sA), ptr(PA), T, D, 1, k, 0, None)
    call("pgl_i8_planes", ptr(Xd), D, ptr(Od), G, ptr(sB), ptr(PB), T, D, G, k, 0, None)
    call("pgl_i8_gram", ptr(PA), ptr(PB), ptr(R), T, D, G, k, None)
    call("pgl_i8_crt", ptr(R), ptr(sA), ptr(sB), ptr(J), ldj, ldj * ldj, T, D, G, k, 0, None)
    torch.cuda.synchronize()
    Jh = J.cpu().numpy()
    low = np.tril(np.ones((D, D), dtype=bool))
    assert np.isfinite(Jh[0, :D, :D][low]).all()
    live = np.abs(X).max(0) > 0            # (an empty column contributes exact zeros: 0 * NaN never forms in the integer path)
    lowl = low & np.outer(live, live)
    assert not np.isfinite(Jh[1, :D, :D][lowl]).any() and not np.isfinite(Jh[2, :D, :D][lowl]).any()
    ref = (X * Om[:, 0:1]).T @ X
    np.testing.assert_allclose(Jh[0, :D, :D][low], ref[low], rtol=1e-11, atol=1e-13 * np.abs(ref).max())
