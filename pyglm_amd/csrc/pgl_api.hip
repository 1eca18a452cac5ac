// extern "C" entry points of libpyglm_hip.so (declared and documented in include/pyglm_hip.h).
#include "pgl_common.h"
#include "../../include/pyglm_hip.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";
void pgl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pgl_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}
int pgl_device_cus(int dev) {
    static std::atomic<int> cus[PGL_MAX_DEVICES];
    const int slot = dev & (PGL_MAX_DEVICES - 1);
    int n = cus[slot].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[slot].store(n, std::memory_order_relaxed);
    }
    return n;
}
int pgl_set_dynamic_lds(const void* fn, size_t bytes, PglPerDevice& flag) {
    const int dev = pgl_device();
    if (flag.done(dev)) return PGL_OK;
    if (bytes > 160 * 1024) { pgl_set_error("LDS request %zu > 160 KiB", bytes); return PGL_ERR_ARG; }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { pgl_set_error("hipFuncSetAttribute(LDS=%zu): %s", bytes, hipGetErrorString(e)); return PGL_ERR_HIP; }
    flag.mark(dev);
    return PGL_OK;
}

int pgl_grow_dynamic_lds(const void* fn, size_t bytes, PglPerDeviceSize& have) {
    std::atomic<size_t>& cur = have.set[pgl_device() & (PGL_MAX_DEVICES - 1)];
    if (bytes <= cur.load(std::memory_order_acquire)) return PGL_OK;
    if (bytes > 160 * 1024) { pgl_set_error("LDS request %zu > 160 KiB", bytes); return PGL_ERR_ARG; }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { pgl_set_error("hipFuncSetAttribute(LDS=%zu): %s", bytes, hipGetErrorString(e)); return PGL_ERR_HIP; }
    cur.store(bytes, std::memory_order_release);
    return PGL_OK;
}

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int pgl_abi_version(void) { return PGL_ABI_VERSION; }
const char* pgl_last_error(void) { return g_err; }

int pgl_row_stats(const int* a, const double* W, double* out, int N, int B, int nloc, int n0, void* st) {
    PGL_CHECK_ARG(a && W && out && N > 0 && B > 0 && B <= 32 && nloc > 0 && n0 >= 0);
    return pgl_k_row_stats(a, W, out, N, B, nloc, n0, ST(st));
}

int pgl_philox_words(uint64_t seed, uint32_t purpose, uint32_t j, uint64_t elem0, uint64_t stream, uint32_t* out, size_t n, void* st) {
    PGL_CHECK_ARG(out != nullptr || n == 0);
    return pgl_k_philox_words(seed, purpose, j, elem0, stream, out, n, ST(st));
}

int pgl_pg_draw(const double* b, const double* z, double* out, size_t len, uint64_t seed, uint64_t stream, uint64_t elem0, void* st) {
    PGL_CHECK_ARG((z != nullptr && out != nullptr) || len == 0);
    return pgl_k_pg_draw(b, z, out, len, seed, stream, elem0, ST(st));
}

int pgl_design_matrix(const double* S, long lds, const double* basis, double* X, long ldx, double* Xt, long ldxt, int T, int N, int B, int R,
                      int clip, void* st) {
    PGL_CHECK_ARG(S && basis && X && T > 0 && N > 0 && B > 0 && R > 0);
    PGL_CHECK_ARG(ldx >= (long)N * B + 1 && lds >= N && (Xt == nullptr || ldxt >= T));
    return pgl_k_basis_conv(S, lds, basis, X, ldx, Xt, ldxt, T, N, B, R, clip, ST(st));
}

int pgl_transpose(const double* src, long ld_src, double* dst, long ld_dst, int rows, int cols, void* st) {
    PGL_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows);
    return pgl_k_transpose(src, ld_src, dst, ld_dst, rows, cols, ST(st));
}

int pgl_activation(const double* Xt, long ldxt, const double* Wt, long ldw, double* Psi, long ldpsi, int T, int Dk, int nloc, void* st) {
    PGL_CHECK_ARG(Xt && Wt && Psi && T > 0 && nloc > 0 && Dk > 0 && Dk % 16 == 0);
    PGL_CHECK_ARG(ldxt >= T && ldw >= nloc && ldpsi >= nloc);
    PglGemmArgs a{};
    a.A = Xt; a.lda = ldxt; a.strideA = 0;
    a.B = Wt; a.ldb = ldw; a.strideB = 0;
    a.C = Psi; a.ldc = ldpsi; a.strideC = 0;
    a.M = T; a.N = nloc; a.K = Dk;
    a.a_cols = (int)(ldxt & ~1L); a.b_cols = (int)(ldw & ~1L);
    a.nbatch = 1; a.alpha = 1.0; a.beta = 0.0; a.tri = 0;
    return pgl_launch_gemm(PGL_GEMM_PLAIN, a, ST(st));
}

int pgl_pg_loglik(double* Psi, long ldpsi, const double* bias, const double* Y, long ldy, double* Omega, long ldo, double* Kappa, long ldk,
                  double* llpart, double* ll_out, int accumulate, int T, int nloc, int obs, double xi, uint64_t seed, uint64_t sweep,
                  uint64_t neuron0, uint64_t elem0, void* st) {
    PGL_CHECK_ARG(Psi && Y && llpart && ll_out && T > 0 && nloc > 0 && (obs == 0 || obs == 1));
    PGL_CHECK_ARG(obs == 0 || xi > 0);
    return pgl_k_pg_loglik(Psi, ldpsi, bias, Y, ldy, Omega, ldo, Kappa, ldk, llpart, ll_out, accumulate, T, nloc, obs, xi, seed, sweep, neuron0,
                           elem0, ST(st));
}
int pgl_pg_loglik_partials(int T) { return pgl_k_pg_loglik_nblk(T); }

int pgl_gaussian_stats(double* Psi, long ldpsi, const double* bias, const double* Y, long ldy, const double* inv_eta, double* Omega, long ldo,
                       double* Kappa, long ldk, double* part, double* sse_out, int accumulate, int T, int nloc, void* st) {
    PGL_CHECK_ARG(Psi && Y && inv_eta && part && sse_out && T > 0 && nloc > 0);
    return pgl_k_gaussian_stats(Psi, ldpsi, bias, Y, ldy, inv_eta, Omega, ldo, Kappa, ldk, part, sse_out, accumulate, T, nloc, ST(st));
}

int pgl_scaled_gram(const double* G0, long ldg, const double* inv_eta, double* J, long ldj, long strideJ, int D, int nz, void* st) {
    PGL_CHECK_ARG(G0 && inv_eta && J && D > 0 && nz > 0 && ldg >= D && ldj >= D && ldg % 2 == 0 && ldj % 2 == 0 && strideJ % 2 == 0);
    return pgl_k_scaled_gram(G0, ldg, inv_eta, J, ldj, strideJ, D, nz, ST(st));
}

int pgl_weighted_gram(const double* X, long ldx, int x_cols, const double* W, long ldw, int Tp, int D, int nz, double* J, long ldj, long strideJ,
                      int accumulate, void* st) {
    PGL_CHECK_ARG(X && W && J && Tp > 0 && Tp % 16 == 0 && D > 0 && nz > 0 && ldj >= D && ldw >= nz && x_cols <= ldx);
    PglGemmArgs a{};
    a.A = X; a.lda = ldx; a.strideA = 0;
    a.B = X; a.ldb = ldx; a.strideB = 0;
    a.C = J; a.ldc = ldj; a.strideC = strideJ;
    a.W = W; a.ldw = ldw;
    a.M = D; a.N = D; a.K = Tp;
    a.a_cols = x_cols & ~1; a.b_cols = x_cols & ~1;
    a.nbatch = (nz + 1) / 2; a.nz_total = nz;
    a.alpha = 1.0; a.beta = accumulate ? 1.0 : 0.0; a.tri = 1;
    return pgl_launch_gemm(PGL_GEMM_GRAM2, a, ST(st));
}

int pgl_contract_tn(const double* A, long lda, int a_cols, const double* B, long ldb, int b_cols, double* C, long ldc, int M, int N, int K,
                    double alpha, double beta, void* st) {
    PGL_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && K % 16 == 0 && ldc >= N);
    PglGemmArgs a{};
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.a_cols = a_cols & ~1; a.b_cols = b_cols & ~1;
    a.nbatch = 1; a.alpha = alpha; a.beta = beta; a.tri = 0;
    return pgl_launch_gemm(PGL_GEMM_PLAIN, a, ST(st));
}

int pgl_contract_tn_batched(const double* A, long lda, long strideA, int a_cols, const double* B, long ldb, long strideB, int b_cols, double* C,
                            long ldc, long strideC, int M, int N, int K, int nbatch, const int* batch_k, double alpha, double beta, int tri,
                            int kernel, void* st) {
    PGL_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && K % 16 == 0 && ldc >= N && nbatch > 0 && tri >= 0 && tri <= 2 && kernel >= 0 && kernel <= 2);
    PGL_CHECK_ARG(tri == 0 || M == N);
    PglGemmArgs a{};
    a.A = A; a.lda = lda; a.strideA = strideA; a.B = B; a.ldb = ldb; a.strideB = strideB; a.C = C; a.ldc = ldc; a.strideC = strideC;
    a.M = M; a.N = N; a.K = K; a.a_cols = a_cols & ~1; a.b_cols = b_cols & ~1;
    a.nbatch = nbatch; a.alpha = alpha; a.beta = beta; a.tri = tri; a.batch_k = batch_k; a.pipe = kernel;
    return pgl_launch_gemm(tri ? PGL_GEMM_TRI1 : PGL_GEMM_PLAIN, a, ST(st));
}

int pgl_assemble_posterior(double* J, long ldj, long strideJ, const double* border_omega, const double* border_kappa, long ldb, const double* Jw,
                           const double* hw, const int* label, const double* Jb, const double* hb, int nb, int N, int B, void* st) {
    PGL_CHECK_ARG(J && border_omega && border_kappa && Jw && hw && Jb && hb && nb > 0 && N > 0 && B > 0);
    PGL_CHECK_ARG(ldj >= (long)N * B + 2 && ldb >= (long)N * B + 1);
    return pgl_k_assemble_post(J, ldj, strideJ, border_omega, border_kappa, ldb, Jw, hw, label, Jb, hb, nb, N, B, ST(st));
}

// ---- integer-MFMA Gram (what the engine uses at large shapes instead of pgl_weighted_gram)
size_t pgl_i8_plane_bytes(int D, int T) { return pgl_k_i8_plane_bytes(D, T); }
size_t pgl_i8_residue_bytes(int D) { return pgl_k_i8_residue_bytes(D); }
int pgl_i8_max_planes(void) { return pgl_k_i8_max_planes(); }
int pgl_i8_padded_rows(int D) { return pgl_k_i8_padded_rows(D); }
int pgl_i8_min_planes(int T) { return pgl_k_i8_min_planes(T); }
int pgl_i8_norm_bits(int nplanes, int T) { return pgl_k_i8_nu(nplanes, T); }
double pgl_i8_norm_limit(int nplanes, int T) { return pgl_k_i8_norm_limit(nplanes, T); }
#define PGL_CHECK_PLANES(np, T) PGL_CHECK_ARG((np) >= 1 && (np) <= pgl_k_i8_max_planes() && pgl_k_i8_nu((np), (T)) >= 8)
int pgl_i8_colstats(const double* X, long ldx, const double* Om, long ldo, int T, int D, int G, double* amax, double* sumsq, void* st) {
    PGL_CHECK_ARG(X && amax && sumsq && T > 0 && D > 0 && G >= 1 && G <= 8 && ldx >= D && (Om == nullptr ? G == 1 : ldo >= G));
    return pgl_k_i8_colstats(X, ldx, Om, ldo, T, D, G, amax, sumsq, ST(st));
}
int pgl_i8_scales(const double* amax, const double* sumsq, long ncols, int T, int nplanes, double* scale, void* st) {
    PGL_CHECK_ARG(amax && sumsq && scale && ncols > 0 && T > 0);
    PGL_CHECK_PLANES(nplanes, T);
    return pgl_k_i8_scales(amax, sumsq, ncols, T, nplanes, scale, ST(st));
}
int pgl_i8_planes(const double* X, long ldx, const double* Om, long ldo, const double* scale, void* planes, int T, int D, int G, int nplanes,
                  long t0, void* st) {
    PGL_CHECK_ARG(X && scale && planes && T > 0 && D > 0 && G > 0 && ldx >= D && (Om == nullptr || ldo >= G) && t0 >= 0);
    PGL_CHECK_ARG(Om != nullptr || G == 1);
    PGL_CHECK_PLANES(nplanes, T);
    return pgl_k_i8_planes(X, ldx, 0, Om, ldo, scale, static_cast<int8_t*>(planes), T, D, G, nplanes, t0, ST(st));
}
int pgl_i8_planes_t(const double* Xt, long ldt, const double* Om, long ldo, const double* scale, void* planes, int T, int D, int G, int nplanes,
                    long t0, void* st) {
    PGL_CHECK_ARG(Xt && scale && planes && T > 0 && D > 0 && G > 0 && ldt >= T && (Om == nullptr || ldo >= G) && t0 >= 0);
    PGL_CHECK_ARG(Om != nullptr || G == 1);
    PGL_CHECK_PLANES(nplanes, T);
    return pgl_k_i8_planes(Xt, ldt, 1, Om, ldo, scale, static_cast<int8_t*>(planes), T, D, G, nplanes, t0, ST(st));
}
int pgl_i8_gram(const void* planes_x, const void* planes_wx, void* residues, int T, int D, int G, int nplanes, void* st) {
    PGL_CHECK_ARG(planes_x && planes_wx && residues && T > 0 && D > 0 && G > 0);
    PGL_CHECK_PLANES(nplanes, T);
    return pgl_k_i8_gram(static_cast<const int8_t*>(planes_x), 0, 0, static_cast<const int8_t*>(planes_wx), static_cast<int8_t*>(residues), nullptr, T, D, G,
                         nplanes, 0, ST(st));
}
int pgl_i8_gram_slice(const void* planes_x, int T_x, int t0, const void* planes_wx, void* residues, int T_slice, int T_total, int D, int G, int nplanes,
                      int accumulate, void* st) {
    PGL_CHECK_ARG(planes_x && planes_wx && residues && T_slice > 0 && D > 0 && G > 0 && t0 >= 0 && t0 % 64 == 0 && T_total >= T_slice);
    PGL_CHECK_ARG(T_x == 0 || t0 + T_slice <= T_x);
    PGL_CHECK_PLANES(nplanes, T_total);
    return pgl_k_i8_gram(static_cast<const int8_t*>(planes_x), T_x > 0 ? pgl_k_i8_kp(T_x) : 0, T_x > 0 ? t0 / 64 : 0, static_cast<const int8_t*>(planes_wx),
                         static_cast<int8_t*>(residues), nullptr, T_slice, D, G, nplanes, accumulate, ST(st));
}
int pgl_i8_crt(const void* residues, const double* scale_x, const double* scale_wx, double* J, long ldj, long strideJ, int T, int D, int G,
               int nplanes, int accumulate, void* st) {
    PGL_CHECK_ARG(residues && scale_x && scale_wx && J && T > 0 && D > 0 && G > 0 && ldj >= D);
    PGL_CHECK_PLANES(nplanes, T);
    return pgl_k_i8_crt(static_cast<const int8_t*>(residues), nullptr, scale_x, scale_wx, J, ldj, strideJ, D, G, nplanes, accumulate, ST(st));
}

static PglFlipState to_state(const pgl_flip_t* s) {
    return PglFlipState{s->M, s->ldj, s->strideM, s->nb, s->N, s->B, s->perm, s->u, s->rho, s->c0, s->a, s->skip,
                        s->d_idx, s->d_sign, s->d_cnt, s->batch_k, s->G, s->Lws, s->Ut, s->Wt, s->ldu, s->status, s->visit_order, s->logodds};
}
int pgl_flip_kmax(void) { return pgl_k_flip_kmax(); }
int pgl_flip_window_blocks(int B) { return pgl_k_flip_window_blocks(B); }
int pgl_flip_apply(const pgl_flip_t* s, void* st) {
    PGL_CHECK_ARG(s && s->M && s->d_idx && s->d_sign && s->d_cnt && s->batch_k && s->G && s->Ut && s->Wt && s->status);
    PGL_CHECK_ARG(s->ldu >= (long)s->N * s->B + 2 && s->ldu % 2 == 0 && s->ldj >= (long)s->N * s->B + 2 && s->nb > 0);
    return pgl_k_flip_apply(to_state(s), 0, 128, -1, ST(st));
}
int pgl_flip_apply_chunk(const pgl_flip_t* s, int max_pivots, void* st) {
    PGL_CHECK_ARG(s && s->M && s->d_idx && s->d_sign && s->d_cnt && s->batch_k && s->G && s->Lws && s->Ut && s->Wt && s->status);
    PGL_CHECK_ARG(s->ldu >= (long)s->N * s->B + 2 && s->ldu % 2 == 0 && s->ldj >= (long)s->N * s->B + 2 && s->nb > 0 && max_pivots > 0);
    return pgl_k_flip_apply(to_state(s), 0, max_pivots, -1, ST(st));
}
int pgl_flip_visit_order(const pgl_flip_t* s, const double* J, long ldj_src, long strideJ, void* st) {
    PGL_CHECK_ARG(s && s->M && s->perm && J && s->visit_order && s->nb > 0 && ldj_src >= (long)s->N * s->B + 2 && s->ldj >= (long)s->N * s->B + 2);
    PGL_CHECK_ARG(s->B >= 1 && s->B <= 32);
    return pgl_k_flip_permute(to_state(s), J, ldj_src, strideJ, ST(st));
}
int pgl_flip_apply_window(const pgl_flip_t* s, int window, void* st) {
    PGL_CHECK_ARG(s && s->M && s->d_idx && s->d_sign && s->d_cnt && s->batch_k && s->G && s->Ut && s->Wt && s->status);
    PGL_CHECK_ARG(s->ldu >= (long)s->N * s->B + 2 && s->ldu % 2 == 0 && s->ldj >= (long)s->N * s->B + 2 && s->nb > 0);
    PGL_CHECK_ARG(window >= 0);
    return pgl_k_flip_apply(to_state(s), 1, 0, window, ST(st));
}
int pgl_flip_decide(const pgl_flip_t* s, int window, void* st) {
    PGL_CHECK_ARG(s && s->M && s->perm && s->u && s->rho && s->c0 && s->a && s->d_idx && s->d_sign && s->d_cnt && s->status && s->Lws);
    PGL_CHECK_ARG(s->B >= 1 && s->B <= 32 && window >= 0);
    return pgl_k_flip_decide(to_state(s), window, ST(st));
}

static PglCholState to_cstate(const pgl_chol_t* s) {
    return PglCholState{s->J, s->ldj, s->strideJ, s->a, s->act, s->ldact, s->na, s->Ac, s->ldc, s->strideC, s->hc, s->Tinv, s->z, s->ldz,
                        s->W, s->b, s->nb, s->N, s->B, s->status};
}
int pgl_active_index(const pgl_chol_t* s, void* st) {
    PGL_CHECK_ARG(s && s->a && s->act && s->na && s->nb > 0 && s->ldact >= (long)s->N * s->B + 1);
    return pgl_k_chol_index(to_cstate(s), ST(st));
}
int pgl_sample_weights(const pgl_chol_t* s, int na_max, void* st) {
    PGL_CHECK_ARG(s && s->J && s->act && s->na && s->Ac && s->hc && s->Tinv && s->z && s->W && s->b && s->status);
    PGL_CHECK_ARG(na_max >= 1 && na_max <= s->N * s->B + 1 && s->ldc >= na_max + 1 && s->ldc % 2 == 0 && s->ldz >= na_max);
    return pgl_k_chol_sample(to_cstate(s), na_max, ST(st));
}

}  // extern "C"
