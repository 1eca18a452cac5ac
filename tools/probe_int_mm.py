"""What the vendor int8 GEMM (hipBLASLt behind torch._int_mm) sustains on this box for the shape of one residue plane
(5120 x T x 5120, random bytes), to put the hand-written kernel's rate in context (same power limit, same clocks)."""
import sys, time, torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
D = 5120
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
A = torch.randint(-128, 128, (D, T), dtype=torch.int8, device=dev, generator=g)
Bs = [torch.randint(-128, 128, (T, D), dtype=torch.int8, device=dev, generator=g) for _ in range(4)]
Bt = [torch.randint(-128, 128, (D, T), dtype=torch.int8, device=dev, generator=g) for _ in range(4)]
for name, ops in (("A[D,T] @ B[T,D]", [(A, b) for b in Bs]), ("A[D,T] @ Bt[D,T].T", [(A, b.t()) for b in Bt])):
    try:
        for a, b in ops: torch._int_mm(a, b)
        torch.cuda.synchronize()
        for rep in (8, 40, 104):            # 104 = the 13 planes x 8 neurons of one launch of ours (full squares here, not triangles)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(rep):
                a, b = ops[r % len(ops)]
                torch._int_mm(a, b)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / rep
            print("%s: %d products back to back: %.3f ms each, %.2f POP/s" % (name, rep, ms, 2.0 * D * D * T / ms / 1e9 / 1e3), flush=True)
    except Exception as e:
        print(name, "failed:", repr(e)[:300])
