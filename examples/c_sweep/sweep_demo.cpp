// A binder of libpyglm_hip.so that is not Python: plain C-style host code over include/pyglm_hip.h.
//
// It does what `SparseBernoulliGLM(N, basis=...).add_data(Y); for it: resample_model()` does for the regressions of one shard
// (pyglm/models.py:66-80, 169-171): synthetic Bernoulli spikes -> pgl_design_matrix -> `nsweeps` calls of pgl_sweep with fresh
// permutations / uniforms / normals from a host generator -> pgl_get_state.  Prints the log-likelihood of the state before every
// sweep and the final density of the adjacency; exit code 0 if every sweep left finite weights and clean status flags.
//
//   hipcc -O2 -I include examples/c_sweep/sweep_demo.cpp -L pyglm_amd/lib -lpyglm_hip -Wl,-rpath,$PWD/pyglm_amd/lib -o examples/c_sweep/sweep_demo
//   examples/c_sweep/sweep_demo [N=24] [B=3] [T=4000] [nsweeps=5]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pyglm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define PGL(x) do { if ((x) != 0) { fprintf(stderr, "libpyglm_hip: %s (line %d)\n", pgl_last_error(), __LINE__); exit(3); } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double unif(void) {                      // xorshift64*
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return ((rng_state * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0) + 1e-17;
}
static double normal(void) { return sqrt(-2.0 * log(unif())) * cos(6.283185307179586 * unif()); }

template <typename T> static T* dev_alloc(size_t n) { T* p; CK(hipMalloc((void**)&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
template <typename T> static void upload(T* d, const T* h, size_t n) { CK(hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice)); }

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 24, B = argc > 2 ? atoi(argv[2]) : 3, T = argc > 3 ? atoi(argv[3]) : 4000;
    const int nsweeps = argc > 4 ? atoi(argv[4]) : 5, L = 12;
    const int D = N * B, nloc = N, nb = N < 16 ? N : 16;
    int Dp, ldn, ldj;
    PGL(pgl_sweep_dims(N, B, nloc, &Dp, &ldn, &ldj));
    const int Tp = (T + 15) / 16 * 16, kmax = pgl_flip_kmax();

    // ---- spikes, a raised-cosine-like basis, the design matrix on the device
    double* hY = (double*)calloc((size_t)T * N, sizeof(double));
    for (long i = 0; i < (long)T * N; ++i) hY[i] = unif() < 0.1 ? 1.0 : 0.0;
    double* hbasis = (double*)calloc((size_t)L * B, sizeof(double));
    for (int l = 0; l < L; ++l)
        for (int b = 0; b < B; ++b) { const double c = (l - (b + 0.5) * L / B) / (0.5 * L / B); hbasis[l * B + b] = fabs(c) < 1 ? 0.5 * (1 + cos(3.141592653589793 * c)) / L : 0.0; }
    double *S = dev_alloc<double>((size_t)T * N), *basis = dev_alloc<double>((size_t)L * B), *X = dev_alloc<double>((size_t)Tp * Dp), *Xt = dev_alloc<double>((size_t)Dp * Tp);
    upload(S, hY, (size_t)T * N);
    upload(basis, hbasis, (size_t)L * B);
    PGL(pgl_design_matrix(S, N, basis, X, Dp, Xt, Tp, T, N, B, L, 1, NULL));
    double* Y = dev_alloc<double>((size_t)T * ldn);
    CK(hipMemcpy2D(Y, (size_t)ldn * sizeof(double), S, (size_t)N * sizeof(double), (size_t)N * sizeof(double), T, hipMemcpyDeviceToDevice));

    pgl_dataset_t ds;
    memset(&ds, 0, sizeof ds);
    ds.T = T; ds.Tp = Tp; ds.X = X; ds.Xt = Xt; ds.Y = Y;
    ds.Psi = dev_alloc<double>((size_t)T * ldn);
    ds.OK = dev_alloc<double>((size_t)Tp * 2 * ldn);
    ds.llpart = dev_alloc<double>((size_t)pgl_pg_loglik_partials(T) * nloc);

    // ---- prior in natural form (pyglm/regression.py:138-151): S_w = 4 I, mu_w = 0, S_b = 1, mu_b = -2, rho = 1/2
    const double sw = 4.0;
    double* hJw = (double*)calloc((size_t)nloc * N * B * B, sizeof(double));
    for (long k = 0; k < (long)nloc * N; ++k)
        for (int i = 0; i < B; ++i) hJw[k * B * B + i * B + i] = 1.0 / sw;
    double *hrho = (double*)malloc(sizeof(double) * nloc * N), *hc0 = (double*)malloc(sizeof(double) * nloc * N);
    for (long k = 0; k < (long)nloc * N; ++k) { hrho[k] = 0.5; hc0[k] = 0.5 * B * log(1.0 / sw); }     // 1/2 log|J_w| - 1/2 mu' J_w mu
    double *hJb = (double*)malloc(sizeof(double) * nloc), *hhb = (double*)malloc(sizeof(double) * nloc);
    for (int n = 0; n < nloc; ++n) { hJb[n] = 1.0; hhb[n] = -2.0; }

    pgl_sweep_t sw_;
    memset(&sw_, 0, sizeof sw_);
    sw_.N = N; sw_.B = B; sw_.n0 = 0; sw_.nloc = nloc; sw_.nb = nb; sw_.obs = 0; sw_.xi = 1.0; sw_.visit_order = 1;
    sw_.datasets = &ds; sw_.ndatasets = 1;
    sw_.a = dev_alloc<int>((size_t)nloc * N); sw_.W = dev_alloc<double>((size_t)nloc * D); sw_.b = dev_alloc<double>(nloc);
    double* d;
    d = dev_alloc<double>((size_t)nloc * N); upload(d, hrho, (size_t)nloc * N); sw_.rho = d;
    d = dev_alloc<double>((size_t)nloc * N * B * B); upload(d, hJw, (size_t)nloc * N * B * B); sw_.Jw = d;
    sw_.hw = dev_alloc<double>((size_t)nloc * N * B);                                         // J_w mu_w = 0
    d = dev_alloc<double>(nloc); upload(d, hJb, nloc); sw_.Jb = d;
    d = dev_alloc<double>(nloc); upload(d, hhb, nloc); sw_.hb = d;
    d = dev_alloc<double>((size_t)nloc * N); upload(d, hc0, (size_t)nloc * N); sw_.c0 = d;
    int* dperm = dev_alloc<int>((size_t)nloc * N);
    double *du = dev_alloc<double>((size_t)nloc * N), *dz = dev_alloc<double>((size_t)nloc * (D + 1));
    sw_.perm = dperm; sw_.u = du; sw_.z = dz;
    sw_.ll = dev_alloc<double>(nloc); sw_.status = dev_alloc<int>(nloc);
    sw_.Wt = dev_alloc<double>((size_t)Dp * ldn); sw_.bias = dev_alloc<double>(nloc); sw_.border = dev_alloc<double>((size_t)2 * ldn * Dp);
    sw_.skip = dev_alloc<int>(nloc);
    sw_.Jbuf = dev_alloc<double>((size_t)nb * ldj * ldj); sw_.Mtab = dev_alloc<double>((size_t)nb * ldj * ldj); sw_.Ac = dev_alloc<double>((size_t)nb * ldj * ldj);
    sw_.hc = dev_alloc<double>((size_t)2 * nb * ldj); sw_.Tinv = dev_alloc<double>((size_t)nb * 64 * 64);
    sw_.G = dev_alloc<double>((size_t)nb * kmax * kmax); sw_.Lws = dev_alloc<double>((size_t)nb * (kmax + 1) * (kmax + 1));
    sw_.Ut = dev_alloc<double>((size_t)nb * kmax * ldj); sw_.Wt_ws = dev_alloc<double>((size_t)nb * kmax * ldj);
    sw_.d_idx = dev_alloc<int>((size_t)nb * kmax); sw_.d_sign = dev_alloc<double>((size_t)nb * kmax); sw_.d_cnt = dev_alloc<int>(nb);
    sw_.batch_k = dev_alloc<int>(nb); sw_.act = dev_alloc<int>((size_t)nb * (D + 1)); sw_.na = dev_alloc<int>(nb);

    // ---- initial state: a ~ Bernoulli(1/2), W = 0.1 N(0, 1) where active, b = -2 (uploaded once: the state then lives on the device)
    int* ha = (int*)malloc(sizeof(int) * nloc * N);
    double *hW = (double*)calloc((size_t)nloc * D, sizeof(double)), *hb = (double*)malloc(sizeof(double) * nloc);
    for (int n = 0; n < nloc; ++n) {
        hb[n] = -2.0;
        for (int m = 0; m < N; ++m) {
            ha[n * N + m] = unif() < 0.5;
            for (int k = 0; k < B; ++k) hW[(size_t)n * D + m * B + k] = ha[n * N + m] ? 0.1 * normal() : 0.0;
        }
    }
    upload(sw_.a, ha, (size_t)nloc * N); upload(sw_.W, hW, (size_t)nloc * D); upload(sw_.b, hb, nloc);

    int* hperm = (int*)malloc(sizeof(int) * nloc * N);
    double *hu = (double*)malloc(sizeof(double) * nloc * N), *hz = (double*)malloc(sizeof(double) * nloc * (D + 1)), *hll = (double*)malloc(sizeof(double) * nloc);
    int* hstatus = (int*)malloc(sizeof(int) * nloc);
    int ok = 1;
    for (int it = 0; it < nsweeps; ++it) {
        for (int n = 0; n < nloc; ++n) {                               // npr.permutation / uniforms / normals of regression.py:286, 315, 334
            for (int m = 0; m < N; ++m) hperm[n * N + m] = m;
            for (int m = N - 1; m > 0; --m) { const int j = (int)(unif() * (m + 1)); const int t = hperm[n * N + m]; hperm[n * N + m] = hperm[n * N + j]; hperm[n * N + j] = t; }
            for (int m = 0; m < N; ++m) hu[n * N + m] = unif();
            for (int k = 0; k <= D; ++k) hz[(size_t)n * (D + 1) + k] = normal();
        }
        upload(dperm, hperm, (size_t)nloc * N); upload(du, hu, (size_t)nloc * N); upload(dz, hz, (size_t)nloc * (D + 1));
        PGL(pgl_sweep(&sw_, 1234, (uint64_t)it, NULL));
        PGL(pgl_get_state(&sw_, ha, hW, hb, hll, hstatus, NULL));
        double ll = 0.0, wmax = 0.0;
        long active = 0;
        for (int n = 0; n < nloc; ++n) { ll += hll[n]; if (hstatus[n]) ok = 0; }
        for (long k = 0; k < (long)nloc * N; ++k) active += ha[k] != 0;
        for (long k = 0; k < (long)nloc * D; ++k) { if (!isfinite(hW[k])) ok = 0; if (fabs(hW[k]) > wmax) wmax = fabs(hW[k]); }
        printf("sweep %d: log-likelihood before %.3f, density after %.3f, max |W| %.3f\n", it, ll, (double)active / ((double)nloc * N), wmax);
    }
    printf(ok ? "ok\n" : "FAILED\n");
    return ok ? 0 : 1;
}
