#!/usr/bin/env python
"""Accuracy study for DESIGN.md section 9 item 1 (CPU only, NumPy): the omega-weighted Gram J = X' diag(omega) X computed
  (a) as the product runs today (fp64 multiply-add; here NumPy/BLAS dgemm),
  (b) by integer modular arithmetic the way an int8-MFMA emulation would (Ozaki scheme II): A = sqrt(omega) * X, columns scaled by
      powers of two to `beta`-bit integers, one exact product per modulus p <= 256 on residues in [-p/2, p/2], CRT reconstruction,
against an extended-precision (x87 long double, 64-bit mantissa) reference on bench-like data (basis-filtered Bernoulli(0.08) spikes,
Polya-gamma-like omega).  Prints the worst error relative to ||a_i|| ||a_j|| for both, per beta / number of moduli."""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyglm_amd.utils.basis import cosine_basis

MODULI = [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193, 191, 181]   # pairwise coprime, <= 256


def design(T, N, B, L, rng):
    basis = cosine_basis(B, L=L) / L
    S = (rng.random((T, N)) < 0.08).astype(np.float64)
    X = np.zeros((T, N * B))
    for b in range(B):
        for n in range(N):
            X[1:, n * B + b] = np.convolve(S[:, n], basis[:, b])[: T - 1]
    return np.maximum(X, 0.0)


def crt(residues, moduli):
    """exact integers from symmetric residues (object arrays of Python ints)"""
    P = 1
    for p in moduli:
        P *= p
    acc = np.zeros(residues[0].shape, dtype=object)
    for r, p in zip(residues, moduli):
        q = P // p
        acc = acc + r.astype(object) * (q * pow(q, -1, p))
    acc = acc % P
    return np.where(acc > P // 2, acc - P, acc), P


def gram_modular(A, beta, nmod):
    T, D = A.shape
    cmax = np.abs(A).max(axis=0)
    e = np.where(cmax > 0, beta - 1 - np.floor(np.log2(np.maximum(cmax, 1e-300))).astype(np.int64), 0)     # |A 2^e| < 2^beta
    Ai = np.rint(np.ldexp(A, e[None, :].astype(np.int32))).astype(np.int64)
    moduli = MODULI[:nmod]
    res = []
    for p in moduli:
        r = Ai % p
        r = np.where(r > p // 2, r - p, r).astype(np.int64)
        c = (r.T @ r) % p                                        # exact: |sum| <= T * 128^2 << 2^63
        res.append(np.where(c > p // 2, c - p, c))
    S, P = crt(res, moduli)
    bound = max(int(v) for v in (Ai.astype(object) ** 2).sum(axis=0))       # |S_ij| <= |a_i| |a_j| <= max_i |a_i|^2 (exact integers)
    assert 2 * bound < P, "moduli product too small for beta=%d: need %d bits, have %d" % (beta, (2 * bound).bit_length(), P.bit_length())
    scale = np.ldexp(1.0, -(e[:, None] + e[None, :]).astype(np.int32))
    return S.astype(np.float64) * scale, P.bit_length()        # (S < 2^110: float conversion rounds once, 2^-53 relative)


def main():
    rng = np.random.default_rng(0)
    T, N, B, L = 20000, 24, 5, 100
    X = design(T, N, B, L, rng)
    omega = 0.25 * rng.gamma(4.0, 0.25, size=T)                 # PG(1, z)-like scale and spread
    t0 = time.time()
    Xl = X.astype(np.longdouble)
    ref = np.asarray((Xl * omega.astype(np.longdouble)[:, None]).T @ Xl, dtype=np.longdouble)
    print("reference (long double) %.1f s; T=%d D=%d" % (time.time() - t0, T, X.shape[1]))
    A = np.sqrt(omega)[:, None] * X
    nrm = np.sqrt((A * A).sum(axis=0))
    denom = (nrm[:, None] * nrm[None, :]).astype(np.longdouble)
    J64 = (X * omega[:, None]).T @ X
    print("fp64 dgemm                      : max |err| / (|a_i||a_j|) = %.2e" % float(np.max(np.abs(J64 - ref) / denom)))
    Jsq = A.T @ A
    print("fp64 dgemm on sqrt(omega) X     : max |err| / (|a_i||a_j|) = %.2e" % float(np.max(np.abs(Jsq - ref) / denom)))
    for beta, nmod in ((30, 9), (40, 12), (45, 13), (48, 14), (50, 15)):
        t0 = time.time()
        Jm, bits = gram_modular(A, beta, nmod)
        print("modular, beta=%2d bits, %2d moduli (%3d-bit product): max |err| / (|a_i||a_j|) = %.2e   (%.0f s)"
              % (beta, nmod, bits, float(np.max(np.abs(Jm - ref) / denom)), time.time() - t0))


if __name__ == "__main__":
    main()
