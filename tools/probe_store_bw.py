#!/usr/bin/env python
"""HBM store / copy ceilings on this box (torch kernels), to put i8_planes_kernel's 4.0-4.4 TB/s in context."""
import torch
n = 6 * 10 ** 9            # 6e9 bytes... int8 elements
for name, fn, byts in [("fill int8 (store only)", lambda a, b: a.fill_(3), 1.0), ("copy int8 (read + store)", lambda a, b: a.copy_(b), 2.0),
                       ("fill f64", lambda a, b: a.view(torch.float64).fill_(1.0), 1.0)]:
    a = torch.empty(n, dtype=torch.int8, device="cuda:0")
    b = torch.empty(n, dtype=torch.int8, device="cuda:0")
    fn(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-28s %7.3f ms  %.2f TB/s" % (name, ms, byts * n / ms * 1e-9))
    del a, b
