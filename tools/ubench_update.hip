// Microbenchmark + bit-for-bit check of the rank-k update pipeline (pyglm_amd/csrc/pgl_update.hip) against the generic fp64 kernel
// (pgl_gemm.hip) on the shapes the sweep uses at cfg3.  Build (from the repo root):
//   hipcc -O2 --offload-arch=gfx950 -I pyglm_amd/csrc tools/ubench_update.hip -L pyglm_amd/lib -lpyglm_hip -Wl,-rpath,$PWD/pyglm_amd/lib -o tools/bin/ubench_update
// Run: tools/bin/ubench_update [nb]      (nb = neurons per batch, default 256)
#include "pgl_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(double* p, size_t n, unsigned seed, double scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = ((double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5) * scale;
    }
}

// mode 1: lower triangle (col <= row), 2: upper, 0: all, within rows < dm[b], cols < dn[b]
__global__ void diff_kernel(const double* a, const double* b, long ld, long stride, int M, int N, int mode, const int* dim, int dim_off, unsigned long long* bad,
                            double* maxabs) {
    const int batch = blockIdx.z;
    const int row = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
    int Mv = M, Nv = N;
    if (dim) { Mv = dim[batch] - dim_off; Nv = Mv; }
    if (row >= Mv || col >= Nv) return;
    if (mode == 1 && col > row) return;
    if (mode == 2 && col < row) return;
    const size_t e = (size_t)batch * stride + (size_t)row * ld + col;
    const double x = a[e], y = b[e];
    if (__double_as_longlong(x) != __double_as_longlong(y) && !(x != x && y != y)) atomicAdd(bad, 1ull);
}

static double time_ms(hipStream_t st, int reps, const std::function<void()>& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 256;
    const int D = 5120, Md = D + 2, ldj = 5136, KMAX = 512;
    const size_t sq = (size_t)ldj * ldj;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    double *C0, *C1, *C2, *W, *U, *G;
    CK(hipMalloc(&C0, nb * sq * 8)); CK(hipMalloc(&C1, nb * sq * 8)); CK(hipMalloc(&C2, nb * sq * 8));
    CK(hipMalloc(&W, (size_t)nb * KMAX * ldj * 8)); CK(hipMalloc(&U, (size_t)nb * KMAX * ldj * 8)); CK(hipMalloc(&G, (size_t)nb * KMAX * KMAX * 8));
    int *bk, *na;
    CK(hipMalloc(&bk, nb * 4)); CK(hipMalloc(&na, nb * 4));
    unsigned long long* bad; double* mx;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&mx, 8));
    fill_kernel<<<4096, 256, 0, st>>>(C0, nb * sq, 1, 2.0);
    fill_kernel<<<4096, 256, 0, st>>>(W, (size_t)nb * KMAX * ldj, 2, 1.0);
    fill_kernel<<<4096, 256, 0, st>>>(U, (size_t)nb * KMAX * ldj, 3, 1.0);
    fill_kernel<<<4096, 256, 0, st>>>(G, (size_t)nb * KMAX * KMAX, 4, 1.0);
    CK(hipStreamSynchronize(st));
    const int n_cu = pgl_device_cus(pgl_device());
    printf("nb = %d, CUs = %d\n", nb, n_cu);

    auto check = [&](const char* name, PglGemmKind kind, PglGemmArgs a, int mode, double flop, bool inplaceC) {
        // old kernel -> C1, new -> C2, both from C0
        const size_t bytes = nb * sq * 8;
        double t_old, t_new;
        for (int which = 0; which < 2; ++which) {
            double* Cw = which ? C2 : C1;
            if (inplaceC) CK(hipMemcpyAsync(Cw, C0, bytes, hipMemcpyDeviceToDevice, st));
            PglGemmArgs q = a;
            q.C = Cw + (a.C - C0);
            if (a.A >= C0 && a.A < C0 + nb * sq) q.A = Cw + (a.A - C0);
            if (a.B >= C0 && a.B < C0 + nb * sq) q.B = Cw + (a.B - C0);
            q.pipe = which;
            int rc = pgl_launch_gemm(kind, q, st);
            if (rc) { printf("%s: launch failed (%d)\n", name, rc); return; }
            CK(hipStreamSynchronize(st));
        }
        CK(hipMemsetAsync(bad, 0, 8, st));
        const double* c1 = C1 + (a.C - C0); const double* c2 = C2 + (a.C - C0);
        diff_kernel<<<dim3((a.N + 255) / 256, a.M, nb), 256, 0, st>>>(c1, c2, a.ldc, a.strideC, a.M, a.N, mode, a.batch_dim && a.dim_mode == 0 ? a.batch_dim : nullptr,
                                                                      a.dim_off, bad, mx);
        unsigned long long hb = 0;
        CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        // timing: repeated in place on C1 / C2 (values drift, time does not)
        for (int which = 0; which < 2; ++which) {
            double* Cw = which ? C2 : C1;
            PglGemmArgs q = a;
            q.C = Cw + (a.C - C0);
            if (a.A >= C0 && a.A < C0 + nb * sq) q.A = Cw + (a.A - C0);
            if (a.B >= C0 && a.B < C0 + nb * sq) q.B = Cw + (a.B - C0);
            q.pipe = which;
            const double ms = time_ms(st, 3, [&] { pgl_launch_gemm(kind, q, st); });
            (which ? t_new : t_old) = ms;
        }
        printf("%-44s old %8.3f ms (%5.1f TF)   new %8.3f ms (%5.1f TF)   x%.3f   mismatching elements: %llu\n", name, t_old, flop / t_old * 1e-9, t_new,
               flop / t_new * 1e-9, t_old / t_new, hb);
        fflush(stdout);
    };

    std::vector<int> h(nb);
    auto set_i = [&](int* d, auto fn) { for (int i = 0; i < nb; ++i) h[i] = fn(i); CK(hipMemcpy(d, h.data(), nb * 4, hipMemcpyHostToDevice)); };

    // ---- flips: M -= W'U on the lower triangle, rank K (batch_k), trailing square from r0
    for (int K : {512, 256, 128, 64}) {
        for (int r0 : {0, 1920}) {
            if (r0 && K == 512) continue;
            set_i(bk, [&](int i) { return K; });
            PglGemmArgs t{};
            const int ncol = ldj - r0;
            t.A = W + r0; t.lda = ldj; t.strideA = (long)KMAX * ldj;
            t.B = U + r0; t.ldb = ldj; t.strideB = (long)KMAX * ldj;
            t.C = C0 + (long)r0 * ldj + r0; t.ldc = ldj; t.strideC = sq;
            t.M = Md - r0; t.N = Md - r0; t.K = KMAX; t.a_cols = ncol; t.b_cols = ncol; t.nbatch = nb;
            t.alpha = -1.0; t.beta = 1.0; t.tri = 1; t.batch_k = bk;
            char name[128];
            snprintf(name, sizeof name, "flip update  K=%d r0=%d", K, r0);
            check(name, PGL_GEMM_TRI1, t, 1, (double)nb * (Md - r0) * (double)(Md - r0) * K, true);
        }
    }
    // ---- ragged ranks: batch_k varies per neuron (a proposal window)
    {
        set_i(bk, [&](int i) { return ((i * 37) % 21) * 16; });
        double ksum = 0; for (int i = 0; i < nb; ++i) ksum += h[i];
        PglGemmArgs t{};
        t.A = W; t.lda = ldj; t.strideA = (long)KMAX * ldj; t.B = U; t.ldb = ldj; t.strideB = (long)KMAX * ldj;
        t.C = C0; t.ldc = ldj; t.strideC = sq; t.M = Md; t.N = Md; t.K = KMAX; t.a_cols = ldj; t.b_cols = ldj; t.nbatch = nb;
        t.alpha = -1.0; t.beta = 1.0; t.tri = 1; t.batch_k = bk;
        check("flip update  ragged K (0..320)", PGL_GEMM_TRI1, t, 1, ksum * Md * (double)Md, true);
    }
    // ---- W = G U  (beta = 0, M = batch_k rows used)
    {
        set_i(bk, [&](int i) { return 512; });
        PglGemmArgs w{};
        w.A = G; w.lda = KMAX; w.strideA = (long)KMAX * KMAX; w.B = U; w.ldb = ldj; w.strideB = (long)KMAX * ldj;
        w.C = C0; w.ldc = ldj; w.strideC = sq;           // (written into the C buffers so that the comparison machinery applies)
        w.M = KMAX; w.N = ldj; w.K = KMAX; w.a_cols = KMAX; w.b_cols = ldj; w.nbatch = nb; w.alpha = 1.0; w.beta = 0.0; w.tri = 0;
        w.batch_k = bk; w.batch_dim = bk; w.dim_off = 0; w.dim_mode = 2;
        check("W = G U  512 x 5136, K=512", PGL_GEMM_PLAIN, w, 0, (double)nb * 2.0 * KMAX * KMAX * ldj, true);
        set_i(bk, [&](int i) { return ((i * 37) % 33) * 16; });
        double s2 = 0; for (int i = 0; i < nb; ++i) s2 += (double)h[i] * h[i];
        check("W = G U  ragged", PGL_GEMM_PLAIN, w, 0, 2.0 * s2 * ldj, true);
    }
    // ---- Cholesky trailing update (upper form): C[c0.., c0..] -= P'P, P = rows [q0, q0+256) of the same matrix, per-neuron sizes
    for (int q0 : {0, 1024, 2560}) {
        const int Kc = 256, c0 = q0 + Kc, na_max = 4000;
        set_i(na, [&](int i) { return 4000 - (i % 16) * 60; });
        double fl = 0; for (int i = 0; i < nb; ++i) { const double r = h[i] - c0; if (r > 0) fl += r * r * Kc; }
        PglGemmArgs t{};
        const double* P = C0 + (long)q0 * ldj + c0;
        const int rem = na_max - c0;
        t.A = P; t.lda = ldj; t.strideA = sq; t.B = P; t.ldb = ldj; t.strideB = sq;
        t.C = C0 + (long)c0 * ldj + c0; t.ldc = ldj; t.strideC = sq;
        t.M = rem; t.N = rem; t.K = Kc; t.a_cols = rem + (rem & 1); t.b_cols = t.a_cols; t.nbatch = nb;
        t.alpha = -1.0; t.beta = 1.0; t.tri = 2; t.batch_dim = na; t.dim_off = c0; t.dim_mode = 0;
        char name[128];
        snprintf(name, sizeof name, "chol trailing K=256 q0=%d (rem<=%d)", q0, rem);
        check(name, PGL_GEMM_TRI1, t, 2, fl, true);
    }
    // ---- does the LAYOUT of C matter for the read-modify-write updates?  One tile column (5122 x 128 outputs per neuron, rank 512) with C rows
    // 41 KB apart (the tableau's row-major layout: a 128 x 128 tile is 128 pieces of 1 KiB) against the same product with C contiguous
    // (ldc = 128: a tile is one 128-KiB run -- what a tile-major tableau would give).  Same operands, same flops, generic kernel.
    for (int K : {512, 256, 64}) {
        set_i(bk, [&](int i) { return K; });
        for (int contiguous = 0; contiguous < 2; ++contiguous) {
            PglGemmArgs t{};
            t.A = W; t.lda = ldj; t.strideA = (long)KMAX * ldj;
            t.B = U; t.ldb = ldj; t.strideB = (long)KMAX * ldj;
            t.C = C1; t.ldc = contiguous ? 128 : ldj; t.strideC = sq;
            t.M = Md - 2; t.N = 128; t.K = KMAX; t.a_cols = ldj; t.b_cols = ldj; t.nbatch = nb;
            t.alpha = -1.0; t.beta = 1.0; t.tri = 0; t.batch_k = bk; t.pipe = 0;
            const double ms = time_ms(st, 5, [&] { pgl_launch_gemm(PGL_GEMM_PLAIN, t, st); });
            printf("C layout test: 5120 x 128 strip, rank %3d, C %s: %8.3f ms (%5.1f TF)\n", K, contiguous ? "contiguous (ldc = 128)  " : "row-major (ldc = 5136)  ", ms,
                   (double)nb * 2.0 * (Md - 2) * 128.0 * K / ms * 1e-9);
        }
    }
    return 0;
}
