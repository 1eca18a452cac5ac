#!/usr/bin/env python
"""Times the stages of the integer Gram (pgl_i8_colstats / scales / planes / gram / crt) at the cfg3 shape (D = 5120, T = 100000, G = 8)
for a given number of residue planes.   python tools/probe_i8.py [planes=13] [G=8] [reps=5]"""
import sys
import torch
sys.path.insert(0, ".")
from pyglm_amd._lib import call, ptr, load


def main(k=13, G=8, reps=5, D=5120, T=100000):
    lib = load()
    dev = torch.device("cuda:0")
    pb, rb = lib.pgl_i8_plane_bytes(D, T) // 15 * k, lib.pgl_i8_residue_bytes(D) // 15 * k
    PA = torch.empty(pb, dtype=torch.int8, device=dev)
    PB = torch.empty(G * pb, dtype=torch.int8, device=dev)
    R = torch.empty(G * rb, dtype=torch.int8, device=dev)
    ldj = (D + 2 + 15) // 16 * 16
    J = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    stat = torch.zeros(2, G, D, dtype=torch.float64, device=dev)
    sA = torch.ones(D, dtype=torch.float64, device=dev)
    sB = torch.ones(G, D, dtype=torch.float64, device=dev)

    def timeit(name, fn, n=reps, work=None, unit=""):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("%-44s %9.3f ms%s" % (name, ms, "   %.2f %s" % (work / ms * 1e-9, unit) if work else ""), flush=True)

    ops = float(k) * G * T * D * (D + 1)
    for buf in (PA, PB):
        for c in range(0, buf.numel(), 1 << 30):
            n = min(1 << 30, buf.numel() - c)
            buf[c:c + n] = torch.randint(-127, 128, (n,), dtype=torch.int8, device=dev)
    timeit("gram, random planes (k=%d, G=%d)" % (k, G), lambda: call("pgl_i8_gram", ptr(PA), ptr(PB), ptr(R), T, D, G, k, None), work=ops, unit="TOP/s")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    Dp = (D + 1 + 15) // 16 * 16
    X = torch.rand(T, Dp, dtype=torch.float64, device=dev, generator=g) * 0.2
    Om = torch.rand(T, G, dtype=torch.float64, device=dev, generator=g) * 0.25
    call("pgl_i8_colstats", ptr(X), Dp, None, 0, T, D, 1, ptr(stat[0]), ptr(stat[1]), None)
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), D, T, k, ptr(sA), None)
    timeit("colstats omega X (G = %d)" % G, lambda: call("pgl_i8_colstats", ptr(X), Dp, ptr(Om), G, T, D, G, ptr(stat[0]), ptr(stat[1]), None),
           work=8.0 * T * D, unit="GB/s")
    call("pgl_i8_scales", ptr(stat[0]), ptr(stat[1]), G * D, T, k, ptr(sB), None)
    timeit("planes X (G = 1)", lambda: call("pgl_i8_planes", ptr(X), Dp, None, 0, ptr(sA), ptr(PA), T, D, 1, k, 0, None), work=(8.0 + k) * T * D, unit="GB/s")
    timeit("planes omega X (G = %d)" % G, lambda: call("pgl_i8_planes", ptr(X), Dp, ptr(Om), G, ptr(sB), ptr(PB), T, D, G, k, 0, None),
           work=(8.0 + k * G) * T * D, unit="GB/s")
    timeit("gram, converted planes", lambda: call("pgl_i8_gram", ptr(PA), ptr(PB), ptr(R), T, D, G, k, None), work=ops, unit="TOP/s")
    timeit("crt", lambda: call("pgl_i8_crt", ptr(R), ptr(sA), ptr(sB), ptr(J), ldj, ldj * ldj, T, D, G, k, 0, None))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    main(*a)
