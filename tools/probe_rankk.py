#!/usr/bin/env python
"""Rate of the read-modify-write rank-k product  C[b] -= A[b]' B[b]  (lower 128-tiles of a 5122 x 5122 tableau per neuron: what the flips'
passes and the Cholesky's trailing updates are made of) as a function of K, on the generic tiles and on the update pipeline.  An item's
prologue (C tile in) and epilogue (C tile out) are the same for every K: if the rate keeps climbing with K the item boundaries are what
holds rank 512 at 0.82 of the fp64 MFMA peak; if it is flat, the K loop is.    python tools/probe_rankk.py [nb=64]"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from pyglm_amd._lib import call, ptr
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M, ld = 5122, 5136
dev = "cuda:0"
C = torch.zeros(nb, ld, ld, dtype=torch.float64, device=dev)
for kernel, name in ((0, "generic 128 x 128 tiles, 2 workgroups per CU"), (2, "update pipeline 256 x 128, persistent")):
    for K in (64, 128, 256, 512, 1024, 2048):
        A = torch.rand(nb, K, ld, dtype=torch.float64, device=dev) * 1e-3
        Bm = torch.rand(nb, K, ld, dtype=torch.float64, device=dev) * 1e-3
        def run():
            call("pgl_contract_tn_batched", ptr(A), ld, K * ld, ld, ptr(Bm), ld, K * ld, ld, ptr(C), ld, ld * ld, M, M, K, nb, None, -1.0, 1.0, 1, kernel, None)
        try:
            run()
        except Exception as e:
            print(name, K, "not supported:", str(e)[:80]); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(2, 2048 // K)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ntile = (M + 127) // 128
        flops = nb * (ntile * (ntile + 1) // 2) * 128 * 128 * 2.0 * K          # what the lower tiles multiply (the last tile row is mostly padding)
        useful = nb * float(M) * M * K
        print("%-48s K = %4d: %8.3f ms  %5.1f TFLOP/s useful (%5.1f issued)" % (name, K, ms, useful / ms * 1e-9, flops / ms * 1e-9), flush=True)
        del A, Bm
