#!/usr/bin/env python
"""Times pgl_i8_planes / pgl_i8_gram at the cfg3 shape (D = 5120, T = 100000, G = 4) on random planes and on planes converted from data."""
import sys
import torch
sys.path.insert(0, ".")
from pyglm_amd._lib import call, ptr, load


def main(D=5120, T=100000, G=4):
    lib = load()
    dev = torch.device("cuda:0")
    pb, rb = lib.pgl_i8_plane_bytes(D, T), lib.pgl_i8_residue_bytes(D)
    PA = torch.empty(pb, dtype=torch.int8, device=dev)
    PB = torch.empty(G * pb, dtype=torch.int8, device=dev)
    R = torch.empty(G * rb, dtype=torch.int8, device=dev)
    ldj = (D + 2 + 15) // 16 * 16
    J = torch.zeros(G, ldj, ldj, dtype=torch.float64, device=dev)
    xmax = torch.ones(D, dtype=torch.float64, device=dev)
    wmax = torch.ones(G, dtype=torch.float64, device=dev)

    def timeit(name, fn, n=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("%-40s %.3f ms" % (name, e0.elapsed_time(e1) / n), flush=True)

    def gram():
        call("pgl_i8_gram", ptr(PA), ptr(PB), ptr(R), T, D, G, None)
        call("pgl_i8_crt", ptr(R), ptr(xmax), ptr(wmax), ptr(J), ldj, ldj * ldj, T, D, G, 0, None)

    for buf in (PA, PB):
        for c in range(0, buf.numel(), 1 << 30):
            n = min(1 << 30, buf.numel() - c)
            buf[c:c + n] = torch.randint(-127, 128, (n,), dtype=torch.int8, device=dev)
    timeit("gram+crt, random planes", gram, n=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
    PA.zero_()
    PB.zero_()
    timeit("gram+crt, zero planes", gram)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    Dp = (D + 1 + 15) // 16 * 16
    X = torch.rand(T, Dp, dtype=torch.float64, device=dev, generator=g) * 0.2
    Om = torch.rand(T, G, dtype=torch.float64, device=dev, generator=g) * 0.25
    xmax.zero_()
    wmax.zero_()
    call("pgl_i8_colmax", ptr(X), Dp, T, D, ptr(xmax), None)
    call("pgl_i8_colmax", ptr(Om), G, T, G, ptr(wmax), None)
    timeit("planes X (G = 1)", lambda: call("pgl_i8_planes", ptr(X), Dp, None, 0, ptr(xmax), None, ptr(PA), T, D, 1, None))
    timeit("planes omega X (G = %d)" % G, lambda: call("pgl_i8_planes", ptr(X), Dp, ptr(Om), G, ptr(xmax), ptr(wmax), ptr(PB), T, D, G, None))
    timeit("gram+crt, converted planes", gram)


if __name__ == "__main__":
    main()
