// Shared declarations for the libpyglm_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>

#define PGL_OK 0
#define PGL_ERR_ARG 1
#define PGL_ERR_HIP 2

void pgl_set_error(const char* fmt, ...);

// Tuning knobs for same-box A/B runs exist only in a -DPGL_AB build (`make ab` -> lib/libpyglm_hip_ab.so, never loaded by the package):
// the shipped library reads no environment variable.
#ifdef PGL_AB
#include <cstdlib>
inline int pgl_ab_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int pgl_ab_int(const char*, int dflt) { return dflt; }
#endif

// Library state is kept PER DEVICE (a process may drive several GPUs, one engine each): the dynamic-LDS attribute of a kernel belongs to the
// device it was set on, and so do the CU count and the scheduler scratch of the persistent launches.
constexpr int PGL_MAX_DEVICES = 32;
int pgl_device();                          // current HIP device (0 if the query fails)
int pgl_device_cus(int dev);               // compute units of `dev`, cached
struct PglPerDevice {                      // "done once on this device" flags, lock-free
    std::atomic<uint32_t> mask{0};
    bool done(int dev) const { return (mask.load(std::memory_order_acquire) >> (dev & 31)) & 1u; }
    void mark(int dev) { mask.fetch_or(1u << (dev & 31), std::memory_order_release); }
};
// hipFuncSetAttribute(MaxDynamicSharedMemorySize = bytes) once per (kernel, device)
int pgl_set_dynamic_lds(const void* fn, size_t bytes, PglPerDevice& flag);
// same for a kernel whose request varies between launches: raised whenever a launch needs more than was set on this device
struct PglPerDeviceSize { std::atomic<size_t> set[PGL_MAX_DEVICES]; };
int pgl_grow_dynamic_lds(const void* fn, size_t bytes, PglPerDeviceSize& have);

#define PGL_CHECK_ARG(cond)                                                              \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            pgl_set_error("%s:%d: argument check failed: %s", __FILE__, __LINE__, #cond); \
            return PGL_ERR_ARG;                                                          \
        }                                                                                \
    } while (0)

#define PGL_CHECK_LAUNCH()                                                                         \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess) {                                                                    \
            pgl_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return PGL_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// ---- fp64 MFMA "TN" contraction (pgl_gemm.hip):  C[m][n] = beta*C + alpha * sum_k w[k] * A[k][m] * B[k][n]
struct PglGemmArgs {
    const double* A; long lda; long strideA;   // A is K x M, row-major (k-major); rows padded to K%16==0
    const double* B; long ldb; long strideB;   // B is K x N, row-major
    double* C; long ldc; long strideC;         // C is M x N, row-major
    const double* W; long ldw;                 // weighted mode: W[k][z_global], z_global = batch*WZ + z
    int M, N, K;
    int a_cols, b_cols;                        // readable columns of A / B rows (even); loads beyond give 0
    int nbatch;                                // batches (weighted mode: groups of WZ weight columns)
    int nz_total;                              // weighted mode: number of valid weight columns / outputs
    double alpha, beta;
    int tri;                                   // 0: all tiles; 1: tiles tm >= tn (lower); 2: tiles tm <= tn (upper)
    const int* batch_k;                        // optional per-batch K (multiple of 16; 0 = skip batch)
    int* sched;                                // persistent launch: 8 per-XCD work counters, zeroed before the launch
    const int* batch_dim; int dim_off;         // optional per-batch size d = max(0, batch_dim[b] - dim_off)
    int dim_mode;                              // 0: M = N = d;  1: M = g.M fixed, N = d;  2: M = d, N = g.N fixed;  3: M = min(g.M, d), N = d
    const int* batch_row_off;                  // optional: batch b reads B and writes C from row batch_row_off[b] on (panels stacked per neuron; not for the TRI kinds' skinny rows)
    int pipe;                                  // 1: a rank-k product of the flips / the Cholesky: may take the update pipeline (pgl_update.hip) where that is faster; 2: must
    // weighted Gram of a small model (one 128 x 128 tile per neuron): K cut into ksplit slices of ks_rows rows, one work item each, whose
    // products go to Cpart[z][slice] (ldc as C) and are added up in slice order by pgl_gram_split (pgl_gemm.hip)
    int ksplit; long ks_rows; double* Cpart; long part_stride_z, part_stride_s;
};
enum PglGemmKind { PGL_GEMM_GRAM2 = 0, PGL_GEMM_PLAIN = 1, PGL_GEMM_TRI1 = 2, PGL_GEMM_SQUARES = 3 };   // SQUARES: PLAIN on the squared elements of A and B
int pgl_launch_gemm(PglGemmKind kind, const PglGemmArgs& a, hipStream_t st);
// X' diag(w_z) X for nz weight columns of a model with D <= 512 columns, K (time) cut into S = ceil(Tp / ks_rows) slices that run as separate work
// items: part = scratch of nz * part_stride_z doubles, part_stride_z >= S * ldj * ldj.  The slices' sums are added in slice order (a fixed
// order: the same bits whatever the launch geometry)
int pgl_gram_split(const double* X, long ldx, int x_cols, const double* W, long ldw, int Tp, int D, int nz, double* J, long ldj, long strideJ, int accumulate,
                   long ks_rows, double* part, long part_stride_z, hipStream_t st);
// the update pipeline (pgl_update.hip): 256 x 128 tiles, DMA-staged, persistent; pgl_launch_gemm routes products marked `pipe` to it
bool pgl_update_supported(const PglGemmArgs& a);
int pgl_launch_update(const PglGemmArgs& a, hipStream_t st);
int pgl_launch_update_rows(const PglGemmArgs& a, int row0, int nrows, hipStream_t st);
// 8 zeroed per-XCD work counters for one persistent launch on stream st (a ring of slots owned by the library; pgl_gemm.hip)
int* pgl_sched_slot(hipStream_t st);

// ---- host-side views of the C-ABI structs (pgl_flip_t / pgl_chol_t of include/pyglm_hip.h), shared by the translation units
struct PglFlipState {
    double* M; long ldj; long strideM; int nb, N, B;
    const int* perm; const double* u; const double* rho; const double* c0; int* a; const int* skip;
    int* d_idx; double* d_sign; int* d_cnt; int* batch_k; double* G; double* Lws; double* Ut; double* Wt; long ldu; int* status;
    int permuted; double* logodds;
};
struct PglCholState {
    const double* J; long ldj; long strideJ; const int* a; int* act; long ldact; int* na;
    double* Ac; long ldc; long strideC; double* hc; double* Tinv; const double* z; long ldz; double* W; double* b; int nb, N, B; int* status;
};

// kernels' host launchers (defined next to their kernels)
int pgl_k_philox_words(uint64_t, uint32_t, uint32_t, uint64_t, uint64_t, uint32_t*, size_t, hipStream_t);
int pgl_k_pg_draw(const double*, const double*, double*, size_t, uint64_t, uint64_t, uint64_t, hipStream_t);
int pgl_k_row_stats(const int* a, const double* W, double* out, int N, int B, int nloc, int n0, hipStream_t st);
int pgl_k_pg_loglik(double*, long, const double*, const double*, long, double*, long, double*, long, double*, double*, int, int, int, int, double,
                    uint64_t, uint64_t, uint64_t, uint64_t, hipStream_t);
int pgl_k_pg_loglik_nblk(int);
int pgl_k_gaussian_stats(double*, long, const double*, const double*, long, const double*, double*, long, double*, long, double*, double*, int, int,
                         int, hipStream_t);
int pgl_k_scaled_gram(const double*, long, const double*, double*, long, long, int, int, hipStream_t);
int pgl_k_basis_conv(const double*, long, const double*, double*, long, double*, long, int, int, int, int, int, hipStream_t);
int pgl_k_transpose(const double*, long, double*, long, int, int, hipStream_t);
int pgl_k_assemble_post(double*, long, long, const double*, const double*, long, const double*, const double*, const int*, const double*,
                        const double*, int, int, int, hipStream_t);
size_t pgl_k_i8_plane_bytes(int, int);
size_t pgl_k_i8_residue_bytes(int);
int pgl_k_i8_max_planes(void);
int pgl_k_i8_padded_rows(int);
int pgl_k_i8_min_planes(int);
int pgl_k_i8_nu(int, int);
double pgl_k_i8_norm_limit(int, int);
int pgl_k_i8_colstats(const double*, long, const double*, long, int, int, int, double*, double*, hipStream_t);
int pgl_k_i8_scales(const double*, const double*, long, int, int, double*, hipStream_t);
int pgl_k_i8_colmax(const double*, long, int, int, double*, hipStream_t);
int pgl_k_i8_scales_bound(const double*, long, const double*, const double*, int, int, int, int, double*, hipStream_t);
size_t pgl_k_i8_stats_scratch_doubles(int, int);
int pgl_k_i8_colstats_scales(const double*, long, const double*, long, int, int, int, int, double*, double*, hipStream_t);
int pgl_k_i8_planes(const double*, long, int transposed, const double*, long, const double*, int8_t*, int, int, int, int, long t_base, hipStream_t);
int pgl_k_i8_gram(const int8_t*, long, int, const int8_t*, int8_t*, int8_t*, int, int, int, int, int, hipStream_t);
long pgl_k_i8_kp(int);
int pgl_k_i8_crt(const int8_t*, const int8_t*, const double*, const double*, double*, long, long, int, int, int, int, hipStream_t);
int pgl_k_flip_apply(const PglFlipState&, int, int, int, hipStream_t);
int pgl_k_flip_apply_pair(const PglFlipState&, int phase, int window, int* ws, hipStream_t);
int pgl_k_flip_permute(const PglFlipState&, const double*, long, long, hipStream_t);
int pgl_k_flip_decide(const PglFlipState&, int, hipStream_t);
int pgl_k_flip_pivot_list(const PglFlipState&, int*, long, int*, hipStream_t);
int pgl_k_flip_pivot_chunk(const PglFlipState&, const int*, long, const int*, int, int, hipStream_t);
int pgl_k_flip_kmax(void);
int pgl_k_flip_window_blocks(int);
int pgl_k_chol_index(const PglCholState&, hipStream_t);
// flips + weight draw of a small model (D + 2 <= pgl_k_small_max_rows(), B <= 16) as one launch, one workgroup per neuron (pgl_small.hip)
bool pgl_k_small_fits(int N, int B);
int pgl_k_small_max_rows(void);
int pgl_k_small_tail(const double* J, long ldj, long strideJ, int nb, int N, int B, const int* perm, const double* u, const double* rho, const double* c0,
                     int* a, const int* skip, const double* z, long ldz, double* W, double* b, int* status, double* logodds, hipStream_t st);
int pgl_k_chol_sample(const PglCholState&, int, hipStream_t);
