/*
 * pyglm_hip.h -- C ABI of libpyglm_hip.so, the MI355X (gfx950) Gibbs hot path of slinderman/pyglm.
 *
 * The reference is pure Python; its native boundary on this path is (a) the third-party `pypolyagamma`
 * sampler and (b) BLAS/LAPACK reached through NumPy/SciPy.  Each entry point below names the reference
 * call site(s) it replaces (paths relative to /root/reference).  Conventions:
 *   - every pointer is a DEVICE pointer unless marked host; the caller owns every data buffer, the library borrows it
 *     for the call only.  The one thing the library allocates itself: per device, on first use, an 8 KiB ring of work
 *     counters for its persistent launches (pgl_gemm.hip, sched_slot), kept until the process ends (and, inside the diagnostic
 *     pgl_ubench_mfma only, 1 MiB of scratch that it frees again before it returns);
 *   - calls act on the CURRENT HIP device (hipSetDevice by the caller); library state (kernel attributes, CU count, the
 *     counter ring) is kept per device, so one process may drive several GPUs;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it;
 *   - return value 0 = ok, non-zero = error, message from pgl_last_error() (host, thread-local);
 *   - no hidden RNG state: stochastic entries take (seed, stream-id, element) counters (see oracle/pg_oracle.c
 *     header for the stream specification);
 *   - matrices are row-major fp64 with explicit leading dimensions (in elements).  "k-major" means the
 *     contraction index is the row index.  Leading dimensions and readable column counts must be even and base
 *     pointers 16-byte aligned for operands of the MFMA contraction.
 */
#ifndef PYGLM_HIP_H
#define PYGLM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PGL_ABI_VERSION 11

int pgl_abi_version(void);
const char* pgl_last_error(void);

/* ---- random stream -------------------------------------------------------------------------------------------- */
/* out[4*i..4*i+3] = Philox4x32-10(counter = (j | purpose<<24, elem0+i, stream lo, stream hi), key = seed). Test hook. */
int pgl_philox_words(uint64_t seed, uint32_t purpose, uint32_t j, uint64_t elem0, uint64_t stream, uint32_t* out, size_t n, void* hip_stream);

/* out[i] ~ PG(b[i], z[i]) (b == NULL -> 1; any real b >= 0.  1 <= b <= 64 is exact: floor(b) - 1 Devroye draws of PG(1, z) plus one draw of
 * PG(1 + frac(b), z) from Windle's alternate rejection sampler (floor(b) Devroye draws for an integer b); the truncated sum-of-gammas series
 * only for b < 1 and b > 64 -- pgl_rng.h).
 * Replaces pypolyagamma.pgdrawvpar(ppgs, n, z, out) at pyglm/regression.py:504-507 (samplers built at :474-477). */
int pgl_pg_draw(const double* b, const double* z, double* out, size_t len, uint64_t seed, uint64_t stream, uint64_t elem0, void* hip_stream);

/* ---- design matrix -------------------------------------------------------------------------------------------- */
/* X[t][n*B+b] = sum_l basis[l][b] * S[t-1-l][n], clipped at 0 if clip; X[t][N*B] = 1 (bias regressor); optional
 * transposed copy Xt[d][t].  Replaces convolve_with_basis, pyglm/utils/basis.py:5-34 (called from models.py:74-75). */
int pgl_design_matrix(const double* S, long lds, const double* basis, double* X, long ldx, double* Xt, long ldxt, int T, int N, int B, int R,
                      int clip, void* hip_stream);
/* dst[c][r] = src[r][c] */
int pgl_transpose(const double* src, long ld_src, double* dst, long ld_dst, int rows, int cols, void* hip_stream);

/* ---- activation, PG draw, log-likelihood ---------------------------------------------------------------------- */
/* Psi[t][n] = sum_d Xt[d][t] * Wt[d][n] for nloc neurons at once (Xt: K x T k-major with K = Dk rows, Wt: K x nloc).
 * Replaces the per-neuron dgemv X.dot(W) at pyglm/regression.py:195-201 (bias is added by pgl_pg_loglik). Dk % 16 == 0. */
int pgl_activation(const double* Xt, long ldxt, const double* Wt, long ldw, double* Psi, long ldpsi, int T, int Dk, int nloc, void* hip_stream);

/* In one pass over Psi (T x nloc): psi += bias; kappa = a(y) - b(y)/2 (regression.py:510-511); omega ~ PG(b(y), psi)
 * (:496-508) on stream (neuron0+n, sweep), element elem0+t; ll_out[n] (+)= sum_t log c + a psi - b log1p(exp psi) (:491-494).
 * obs 0 = Bernoulli (:514-522), 1 = negative binomial a=y, b=y+xi. Omega/Kappa may be NULL (log-likelihood only).
 * llpart: scratch of pgl_pg_loglik_partials(T) * nloc doubles. */
int pgl_pg_loglik(double* Psi, long ldpsi, const double* bias, const double* Y, long ldy, double* Omega, long ldo, double* Kappa, long ldk,
                  double* llpart, double* ll_out, int accumulate, int T, int nloc, int obs, double xi, uint64_t seed, uint64_t sweep,
                  uint64_t neuron0, uint64_t elem0, void* hip_stream);
int pgl_pg_loglik_partials(int T);

/* Gaussian observations (SparseGaussianRegression, pyglm/regression.py:380-446): Psi[t][n] += bias[n] (mean, :430-431); when Omega is
 * non-NULL Omega[t][n] = inv_eta[n] (:421-423) and Kappa[t][n] = Y[t][n] * inv_eta[n] (:425-426); sse_out[n] (+)= sum_t (y - psi)^2, the
 * statistic of both log_likelihood (:399-403) and _resample_eta (:433-445). part: scratch as for pgl_pg_loglik. */
int pgl_gaussian_stats(double* Psi, long ldpsi, const double* bias, const double* Y, long ldy, const double* inv_eta, double* Omega, long ldo,
                       double* Kappa, long ldk, double* part, double* sse_out, int accumulate, int T, int nloc, void* hip_stream);

/* ---- likelihood statistics ------------------------------------------------------------------------------------ */
/* J[z][i][j] = inv_eta[z] * G0[i][j] for j <= i < D (row pairs up to the diagonal): the Gaussian model's omega is constant in t, so
 * X'OX of pyglm/regression.py:251-252 is (1/eta) X'X with X'X = G0 formed once per dataset by pgl_weighted_gram with unit weights. */
int pgl_scaled_gram(const double* G0, long ldg, const double* inv_eta, double* J, long ldj, long strideJ, int D, int nz, void* hip_stream);

/* J[z][i][j] (+)= sum_t W[t][z] * X[t][i] * X[t][j], i >= j tiles only (lower triangle valid), z in [0, nz), i,j in [0, D).
 * Replaces XO = X*omega[:,None]; J += XO.T.dot(X) at pyglm/regression.py:251-252 without the T x D temporary.
 * X: Tp x ldx (Tp % 16 == 0, rows >= T zero); W: Tp x ldw weights (rows >= T zero). */
int pgl_weighted_gram(const double* X, long ldx, int x_cols, const double* W, long ldw, int Tp, int D, int nz, double* J, long ldj, long strideJ,
                      int accumulate, void* hip_stream);
/* C[r][c] (+)= sum_t A[t][r] * X[t][c]  (plain k-major contraction; the border sums X'omega, sum omega, X'kappa, sum kappa of
 * pyglm/regression.py:253-260 with A = [Omega | Kappa] and X carrying the ones column). */
int pgl_contract_tn(const double* A, long lda, int a_cols, const double* B, long ldb, int b_cols, double* C, long ldc, int M, int N, int K,
                    double alpha, double beta, void* hip_stream);
/* The same contraction for a batch of independent operands, C[b] = beta C[b] + alpha A[b]' B[b], b < nbatch (operands strideX doubles apart):
 * the rank-k products of the collapsed flips (M -= W'U on every neuron's sweep tableau; pyglm/regression.py:282-320, where the reference
 * refactors the active block per proposal) and of the blocked Cholesky of the weight draw (pyglm/regression.py:323-340).
 * tri: 0 all tiles, 1 the tiles of the lower triangle (M == N), 2 of the upper.  batch_k: optional per-batch K (multiples of 16, 0 = skip).
 * kernel: 0 generic tiles, 1 the update pipeline (256 x 128 tiles, DMA-staged, persistent) where it is the faster one, 2 always the
 * pipeline -- every choice gives the same bits (tests/test_gpu_update.py). */
int pgl_contract_tn_batched(const double* A, long lda, long strideA, int a_cols, const double* B, long ldb, long strideB, int b_cols, double* C,
                            long ldc, long strideC, int M, int N, int K, int nbatch, const int* batch_k, double alpha, double beta, int tri,
                            int kernel, void* hip_stream);
/* J_post = J_lkhd + J_prior, h_post = h_lkhd + h_prior in the (D+2)-square layout [J, bias row D, potential row D+1].
 * Replaces _prior_sufficient_statistics + the additions at pyglm/regression.py:210-223, 253-260, 270-271.
 * border: 2*nloc_b rows (omega sums then kappa sums) x ldb; Jb [nb], hb [nb]; the block-diagonal prior either dense, Jw [nb][N][B][B] and
 * hw [nb][N][B] with label = NULL, or as tables of the K distinct blocks, Jw [K][B][B] and hw [K][B], with label [nb][N] naming the
 * block of (postsynaptic n, presynaptic m) -- a network prior (pyglm/networks.py) pushes a handful of distinct blocks. */
int pgl_assemble_posterior(double* J, long ldj, long strideJ, const double* border_omega, const double* border_kappa, long ldb, const double* Jw,
                           const double* hw, const int* label, const double* Jb, const double* hb, int nb, int N, int B, void* hip_stream);

/* ---- the same weighted Gram on the INTEGER matrix cores -------------------------------------------
 * X'OX of pyglm/regression.py:251-252 computed exactly on operands rounded, column by column, to integers
 *     A[t][i] = round(x_ti sA_i),  B_g[t][j] = rint((omega_gt x_tj) sB_gj),      sA, sB powers of two,
 * (the columns of X, whose values repeat from bin to bin, are rounded with a dither u(t, i) in [0, 1) -- a fixed hash of the GLOBAL time bin and
 * the column: floor(v) + [frac(v) + u >= 1], unbiased and independent across bins, exact for integer v; omega_gt x_tj to the nearest integer)
 * one int8 GEMM per modulus for the first `nplanes` of 15 pairwise coprime moduli <= 256 (256, 255, 253, 251, 247, 241, 239, 233, 229,
 * 227, 223, 217, 211, 199, 197), int32 accumulation (re-reduced every 128 000 bins), exact Chinese-remainder reconstruction,
 * J_ij = (S_ij / sA_i) / sB_gj.  The scales are set from each column's Euclidean norm and largest element so that the integer columns have
 * norms in (limit / 2, limit], limit = pgl_i8_norm_limit(nplanes, T) (2^46.9 / 2^50.8 / 2^54.6 / 2^58.4 for 12 / 13 / 14 / 15 moduli;
 * pgl_i8_norm_bits = floor(log2)) unless their largest element would reach 2^50; Cauchy-Schwarz then keeps every |S_ij| inside the CRT
 * range for any data, and with independent roundings the error of J_ij has standard deviation sqrt((|A_i|^-2 + |B_gj|^-2) / 12)
 * |a_i||b_gj| (2-4e-16 |a_i||b_gj| at 13 moduli; a few times that where a column repeats few distinct values): pinned to the column norms.
 *   pgl_i8_colstats amax[g][c] = max_t |v_tc|, sumsq[g][c] = sum_t v_tc^2 with v = X (Om = NULL, G = 1) or Om[:, g] * X, G <= 8; one pass
 *                   over X, deterministic (fixed summation order)
 *   pgl_i8_scales   scale[k] from (amax[k], sumsq[k]) for ncols = G * D columns (1 for an empty column, NaN for a non-finite one)
 *   pgl_i8_planes   residue planes [G][nplanes] of Dq * Kp signed bytes each (Dq = pgl_i8_padded_rows(D): D rounded up to the product
 *                   kernel's tile edge, 320; Kp = T rounded up to 64, at least 256), BLOCKED as [Dq / 16][Kp / 64][16][64]: the 64 time bins of K tile k of row r are at ((r / 16) (Kp / 64) + k) 1024
 *                   + (r % 16) 64; of X (Om = NULL, G = 1) or of omega_g X for the G columns of Om; scale [G][D]; buffer sizes from
 *                   pgl_i8_plane_bytes (per neuron, at the full 15 planes); t0 = index in its data set of the first bin given (0 unless X is
 *                   converted in time slices): it keys the dither of the X planes, so that slices give the integers of the whole
 *   pgl_i8_gram     residues[g][q] = (planes_x[q] planes_wx[g][q]') mod p_q, lower-triangular tiles, [G][nplanes][Dq][Dq] signed bytes
 *                   (buffer: G * pgl_i8_residue_bytes(D)); any G >= 1 -- multiples of 8 fill the per-XCD work lists (8 where a plane has many
 *                   tiles, up to PGL_I8_MAX_GROUP where it has few: the launch should be several rounds of 256 items)
 *   pgl_i8_crt      J[g] (+)= X' diag(omega_g) X, lower triangle, from the residues (scale_x [D], scale_wx [G][D])
 * The same nplanes must be used for the scales, both plane sets, the product and the reconstruction. */
int pgl_i8_max_planes(void);                     /* 15 */
int pgl_i8_padded_rows(int D);                   /* Dq */
int pgl_i8_min_planes(int T);                    /* fewest moduli with norm bits >= 50: 13 (the engine's default) */
int pgl_i8_norm_bits(int nplanes, int T);        /* floor(log2(pgl_i8_norm_limit)) */
double pgl_i8_norm_limit(int nplanes, int T);    /* norm of the integer columns */
size_t pgl_i8_plane_bytes(int D, int T);
size_t pgl_i8_residue_bytes(int D);
int pgl_i8_colstats(const double* X, long ldx, const double* Om, long ldo, int T, int D, int G, double* amax, double* sumsq, void* hip_stream);
int pgl_i8_scales(const double* amax, const double* sumsq, long ncols, int T, int nplanes, double* scale, void* hip_stream);
int pgl_i8_planes(const double* X, long ldx, const double* Om, long ldo, const double* scale, void* planes, int T, int D, int G, int nplanes,
                  long t0, void* hip_stream);
/* the same planes, byte for byte, read from the TRANSPOSED copy Xt [D][ldt] (what pgl_design_matrix / pgl_transpose also produce): its rows
 * are contiguous in time, which is the order the planes are written in -- 5-8 % faster at cfg3; what pgl_sweep uses */
int pgl_i8_planes_t(const double* Xt, long ldt, const double* Om, long ldo, const double* scale, void* planes, int T, int D, int G, int nplanes,
                    long t0, void* hip_stream);
int pgl_i8_gram(const void* planes_x, const void* planes_wx, void* residues, int T, int D, int G, int nplanes, void* hip_stream);
/* One time slice [t0, t0 + T_slice) of the same product, for data sets whose planes do not fit in memory at once (BASELINE configs[4]:
 * 86 GB of planes per neuron): planes_wx holds the slice only (pgl_i8_planes_t on the slice's rows); planes_x either the whole data set
 * (T_x = its bins; the slice is read at t0, a multiple of 64) or a slice of its own (T_x = 0).  accumulate: add to the residues of the
 * earlier slices.  T_total = bins of the whole product (it fixes the moduli the scales were chosen for).  The integer Gram is a sum over
 * time, so slices add up exactly. */
int pgl_i8_gram_slice(const void* planes_x, int T_x, int t0, const void* planes_wx, void* residues, int T_slice, int T_total, int D, int G,
                      int nplanes, int accumulate, void* hip_stream);
int pgl_i8_crt(const void* residues, const double* scale_x, const double* scale_wx, double* J, long ldj, long strideJ, int T, int D, int G,
               int nplanes, int accumulate, void* hip_stream);

/* ---- collapsed adjacency resampling (pyglm/regression.py:282-320 + 343-378) ------------------------------------ */
typedef struct {
    double* M; long ldj; long strideM;   /* sweep tableau per neuron, (D+2)^2, lower triangle; starts as a copy of J_post */
    int nb, N, B;
    const int* perm;                     /* [nb][N] proposal order (npr.permutation at :286) */
    const double* u;                     /* [nb][N] uniform of proposal step k (sample_discrete_from_log at :315) */
    const double* rho;                   /* [nb][N] */
    const double* c0;                    /* [nb][N] prior term of block m: 1/2 log|J_w| - 1/2 mu_w' J_w mu_w */
    int* a;                              /* [nb][N] in/out */
    const int* skip;                     /* [nb] or NULL: 1 = deterministic sparsity (regression.py:153-155, 274-275): no proposals */
    int* d_idx; double* d_sign; int* d_cnt;   /* [nb][kmax], [nb][kmax], [nb]: pivot list of the pending tableau update */
    int* batch_k;                        /* [nb] scratch */
    double* G;                           /* [nb][kmax][kmax] scratch */
    double* Lws;                         /* [nb][(kmax+1)^2] scratch: sub-tableau of the current proposal window */
    double* Ut; double* Wt; long ldu;    /* [nb][kmax][ldu] scratch, ldu >= D+2, even */
    int* status;                         /* [nb] sticky flags: 1 non-PD block, 2 singular pivot, 4 non-PD posterior */
    int visit_order;                     /* 0: tableau rows in J's order.  1: rows / columns in PROPOSAL order (position k = block perm[k];
                                          * bias and potential rows last; built by pgl_flip_visit_order): d_idx are positions, and the
                                          * update after window w touches only the rows not yet proposed */
    double* logodds;                     /* [nb][N] or NULL.  out: logodds[n][k] = lps[1] - lps[0] of proposal step k (block perm[k]) as
                                          * formed at pyglm/regression.py:293-307 -- the change in log marginal likelihood (:343-378)
                                          * plus log rho - log(1 - rho); NaN where rho is exactly 0 or 1 (the reference's 0 log 0) */
} pgl_flip_t;
int pgl_flip_kmax(void);                         /* max pivots (scalar rows) per tableau update: 512 */
int pgl_flip_window_blocks(int B);               /* blocks proposed per window */
int pgl_flip_apply(const pgl_flip_t* s, void* hip_stream);              /* sweep the tableau on the listed pivots (<= 128 per call is fast) */
/* same for lists of up to max_pivots <= pgl_flip_kmax() = 512 rows per neuron that are all switched ON (the pivot block is positive
 * definite): the initial sweep on the active set; the pivot-block inverse is assembled by recursive 2 x 2 blocking from in-register
 * 128 x 128 inversions. Uses Lws. */
int pgl_flip_apply_chunk(const pgl_flip_t* s, int max_pivots, void* hip_stream);
/* same, right after pgl_flip_decide(window) (which already left G = (M_DD)^-1); with visit_order = 1 only the trailing square from
 * position (window+1)*R on is updated -- rows of proposed blocks are never read again (nothing at all after the last window) */
int pgl_flip_apply_window(const pgl_flip_t* s, int window, void* hip_stream);
/* M[z] = P_z J[z] P_z' (lower triangle; P_z from perm[z]): the sweep tableau in proposal order, instead of a plain copy of J_post */
int pgl_flip_visit_order(const pgl_flip_t* s, const double* J, long ldj_src, long strideJ, void* hip_stream);
int pgl_flip_decide(const pgl_flip_t* s, int window, void* hip_stream); /* run one window of proposals; fills the pivot list */

/* ---- weight conditional (pyglm/regression.py:323-340) ---------------------------------------------------------- */
typedef struct {
    const double* J; long ldj; long strideJ;   /* assembled posterior */
    const int* a;                              /* [nb][N] */
    int* act; long ldact; int* na;             /* [nb][ldact] scratch (ldact >= D+1), [nb] active sizes (out) */
    double* Ac; long ldc; long strideC;        /* [nb][ldc][ldc] scratch */
    double* hc;                                /* [2][nb][ldc] scratch */
    double* Tinv;                              /* [nb][64][64] scratch */
    const double* z; long ldz;                 /* [nb][ldz] standard normals; the first na[n] are consumed (randn at sample_gaussian) */
    double* W; double* b;                      /* out: [nb][N*B] (zeros where a = 0), [nb] */
    int nb, N, B;
    int* status;
} pgl_chol_t;
int pgl_active_index(const pgl_chol_t* s, void* hip_stream);                  /* fills act / na */
int pgl_sample_weights(const pgl_chol_t* s, int na_max, void* hip_stream);    /* na_max >= max_n na[n] (read back by the caller) */

/* ---- one Gibbs sweep of a shard of neurons in ONE call ------------------------------------------------------------------------
 * Replaces the loop `for n in range(N): regressions[n].resample(...)` at pyglm/models.py:169-171 with the body pyglm/regression.py:265-280
 * for the nloc postsynaptic neurons [n0, n0 + nloc) of one GPU: activation (:195-201), PG draw / kappa / log-likelihood (:491-511),
 * likelihood statistics (:225-262), prior statistics (:210-223), collapsed flips (:282-320 with :343-378), weight draw (:323-340).
 * The call only queues work on the stream -- no host synchronisation, nothing decided on the host between launches -- so it returns
 * in milliseconds; pgl_get_state (or any later synchronisation of the stream) waits for it.  All pointers are device pointers owned by
 * the caller.  Dimensions: D = N B, (Dp, ldn, ldj) from pgl_sweep_dims, kmax = pgl_flip_kmax().  Batches of nb neurons share the
 * `[nb]...` scratch; the chain state (a, W, b) lives on the device and is updated in place. */
typedef struct {
    int T, Tp;                     /* time bins; rows of the padded arrays, Tp % 16 == 0 */
    const double* X;               /* [Tp][Dp]  design matrix, column D = 1 (bias regressor), padding 0 (pgl_design_matrix) */
    const double* Xt;              /* [Dp][Tp]  its transpose */
    const double* Y;               /* [T][ldn]  counts of the local neurons */
    double* Psi;                   /* [T][ldn]  out: activations */
    double* OK;                    /* [Tp][2 ldn]  out: [Omega | Kappa]; rows >= T must be zero (and stay zero) */
    double* llpart;                /* [pgl_pg_loglik_partials(T)][nloc] scratch */
    uint64_t elem0;                /* PG stream element of the first bin: sum of T of the data sets before this one */
    int int8;                      /* 1: likelihood Gram on the integer matrix cores (sA, PA valid); 0: fp64 MFMA kernel */
    int planes;                    /* moduli of this data set's planes (0: pgl_sweep_t.planes) */
    const double* sA;              /* [D] column scales of X (pgl_i8_scales) */
    const void* PA;                /* residue planes of X (pgl_i8_planes), `planes` of them; NULL: converted per time slice into pgl_sweep_t.i8_PAs */
    const double* omega_override;  /* optional [T][nloc]: replaces the PG draws (test hook: the reference fixtures inject omega) */
    const double* xmax;            /* optional [D]: max_t |X[t][d]| (the column maxima pgl_i8_colstats gives for Om = NULL).  With it and
                                    * pgl_sweep_t.i8_norm the scales of omega_n X are taken for a whole batch of neurons at once (see there) */
} pgl_dataset_t;

#define PGL_I8_MAX_GROUP 64
#define PGL_NSTAGES 16
typedef struct {                   /* host; zero-initialise.  Per stage (pgl_stage_name): HIP-event time on the launch stream, calls, work */
    double ms[PGL_NSTAGES];
    double work[PGL_NSTAGES];
    int calls[PGL_NSTAGES];
    void* pending;                 /* events recorded by pgl_sweep and not yet folded in by pgl_stage_times_collect */
    unsigned int mask;             /* 0: time every stage; else only the stages whose bit (1 << index) is set (a timed benchmark region
                                    * keeps the dominant kernel's events and drops the other few thousand per sweep) */
} pgl_stage_times_t;

typedef struct {
    int N, B, n0, nloc, nb;        /* neurons, basis functions, first local neuron (global index), local neurons, neurons per batch */
    int obs; double xi;            /* 0 Bernoulli, 1 negative binomial (b = y + xi), 2 Gaussian */
    int visit_order;               /* 1: sweep tableau in proposal order (see pgl_flip_t) */
    int planes, i8_group;          /* integer Gram: moduli in use where a data set does not say, neurons per launch (<= PGL_I8_MAX_GROUP; 8 at large D,
                                    * more where a plane has only a few tiles: multiples of 8 fill the per-XCD work lists) */
    const pgl_dataset_t* datasets; int ndatasets;      /* host array */
    int* a; double* W; double* b;  /* chain state, in/out: [nloc][N] (0/1), [nloc][D] (zeros where a = 0), [nloc] */
    const double* rho;             /* [nloc][N] */
    const double* Jw; const double* hw; const int* label;   /* prior blocks: dense [nloc][N][B][B] / [nloc][N][B] with label = NULL, or
                                    * tables [K][B][B] / [K][B] with label [nloc][N] (see pgl_assemble_posterior) */
    const double* Jb; const double* hb;                /* [nloc] */
    const double* c0;              /* [nloc][N] (label = NULL) or table [K]: 1/2 log|J_w| - 1/2 mu_w' J_w mu_w of the block */
    const int* perm; const double* u; const double* z; /* [nloc][N], [nloc][N], [nloc][D + 1]: permutation, uniforms, normals (see pgl_flip_t, pgl_chol_t) */
    const double* inv_eta; const double* G0;           /* Gaussian model only: [nloc], [ldj][ldj] = X'X */
    double* ll;                    /* out [nloc]: log-likelihood (Gaussian: sum of squared residuals) under the state BEFORE the sweep */
    int* status;                   /* out [nloc]: sticky flags, see pgl_flip_t */
    double* logodds;               /* optional out [nloc][N], see pgl_flip_t */
    /* scratch */
    double* Wt; double* bias; double* border; int* skip; double* c0_dense;   /* [Dp][ldn], [nloc], [2 ldn][Dp], [nloc], [nloc][N] (label only) */
    double* Jbuf; double* Mtab; double* Ac;            /* [nb][ldj][ldj] each; Ac may be the same buffer as Mtab (the tableau is dead when the weight draw starts) */
    double* hc; double* Tinv; double* G; double* Lws;  /* [2][nb][ldj], [nb][64][64], [nb][kmax][kmax], [nb][(kmax + 1)^2] */
    double* Ut; double* Wt_ws;     /* [nb][kmax][ldj] each */
    int* d_idx; double* d_sign; int* d_cnt; int* batch_k;   /* [nb][kmax], [nb][kmax], [nb], [nb] */
    int* act; int* na;             /* [nb][D + 1], [nb] */
    void* i8_PB; void* i8_R; double* i8_stat;          /* integer Gram: i8_group * pgl_i8_plane_bytes, i8_group * pgl_i8_residue_bytes (each
                                    * * planes / 15 suffices), [3][i8_group][D] */
    int i8_slice;                  /* time bins per slice of the integer Gram (a multiple of 64; 0 = the whole data set at once): i8_PB then
                                    * holds i8_group slices of pgl_i8_plane_bytes(D, i8_slice) */
    void* i8_PAs;                  /* planes of one slice of X, pgl_i8_plane_bytes(D, i8_slice or T): used for data sets with int8 = 1 and PA = NULL
                                    * (their X planes are converted per slice -- per group where there is one slice -- instead of kept) */
    void* i8_Rx;                   /* optional: 3 * i8_group * Dq^2 bytes.  With it (and groups that are multiples of 8) the product kernel cuts the items of the LAST
                                    * residue plane into four K quarters, which evens out its final rounds (13 x 136 items per XCD of 32 CUs at
                                    * BASELINE configs[2]: 55.25 rounds -> 51 + 17/4); the quarters' residues are added by the CRT.  NULL: no split.
                                    * Exact integer arithmetic either way: the same J to the last bit */
    double* i8_norm;               /* optional scratch, (nb rounded up to even) * Dp + nloc doubles.  With it (and pgl_dataset_t.xmax) the norms
                                    * the integer Gram scales the columns of omega_n X from are ONE fp64 MFMA contraction of the squared operands
                                    * per batch of nb neurons, and the largest element of a column is bounded by max_t omega_nt * xmax[d]: X is
                                    * read once per batch instead of once per group of 8 neurons.  The scales can only come out SMALLER than
                                    * from the exact column statistics (a column that one element dominates loses a bit or two of its 50);
                                    * NULL: pgl_i8_colstats per group */
    int nrun;                      /* sweep only nrun local neurons (0 = all nloc), those from nfirst on: what a rank of a larger job would do,
                                    * timed on this GPU (bench.py scaling_proxy); the state of the others is left alone */
    int nfirst;                    /* first local neuron of such a partial sweep (even; 0 with nrun = 0): the shard [nfirst, nfirst + nrun) of a
                                    * G-rank job is batched from nfirst on, as the rank that owns it would batch it */
    /* hints (0 = unknown): nothing depends on them but the number of (possibly empty) launches -- in particular not a bit of the result, so
     * that a shard reproduces its rows of the whole model */
    int all_deterministic;         /* 1: the caller knows every row has rho in {0, 1} (regression.py:153-155): the flip stage is not launched */
    int init_rows_bound;           /* upper bound of 1 + B * (active blocks of any local neuron) BEFORE the sweep */
    int active_rows_bound;         /* upper bound of the same AFTER the flips (only known when all_deterministic) */
    int flip_single_pass;          /* 1: one pass over the trailing tableau per proposal window instead of one per pair of windows -- the same
                                    * multiply-adds in the same order, the same bits (tests/test_gpu_parity.py compares the two) */
    pgl_stage_times_t* times;      /* optional (host): stage timing */
} pgl_sweep_t;

int pgl_sweep_dims(int N, int B, int nloc, int* Dp, int* ldn, int* ldj);
/* Sufficient statistics of a shard's rows for the network prior: out[n] = [count, sum_m w_m (B), sum_m w_m w_m' (B x B, row-major)] over the
 * active presynaptic neurons m != n0 + n of local neuron n -- what pyglm/networks.py:132-149 needs of W[A & ~eye] (an NIW update is a function
 * of the count, the sum and the sum of outer products).  a [nloc][N] (0/1), W [nloc][N*B] as pgl_sweep leaves them; out [nloc][1 + B + B*B].
 * Every entry is accumulated over m in ascending order by one thread: a neuron's numbers do not depend on the shard it is in.  The ranks
 * exchange these 1 + B + B^2 doubles per neuron with their rows, so that no rank walks the N^2 B doubles of the gathered state. */
int pgl_row_stats(const int* a, const double* W, double* out, int N, int B, int nloc, int n0, void* hip_stream);
int pgl_sweep(const pgl_sweep_t* s, uint64_t seed, uint64_t sweep, void* hip_stream);
/* copies the shard's state to host buffers (any may be NULL) and waits for the stream: a [nloc][N], W [nloc][D], b, ll, status [nloc].
 * The read-backs of pyglm/models.py:54-64 (weights / adjacency / biases) for a non-Python binder. */
int pgl_get_state(const pgl_sweep_t* s, int* a_host, double* W_host, double* b_host, double* ll_host, int* status_host, void* hip_stream);
const char* pgl_stage_name(int i);
int pgl_stage_times_collect(pgl_stage_times_t* t);    /* waits for the recorded events and adds them to ms / calls / work */

/* ---- box calibration (diagnostic; nothing on the sampling path calls it) ------------------------------------------------------- */
/* What the matrix cores of the current device sustain right now: a register-only MFMA loop on every CU for ~`seconds` (a quarter of it
 * untimed first, so that clocks and the package power limiter settle), timed with HIP events on `stream`; WAITS for the stream.
 *   kind 0: v_mfma_i32_16x16x64_i8 on random operand bytes (what residue planes look like to the multipliers)  -> *rate_out in op/s
 *   kind 1: v_mfma_f64_16x16x4_f64                                                                              -> *rate_out in flop/s
 * (2 per multiply-add).  ms_out (optional, host): duration of the timed launch.  Allocates and frees ~1 MiB of device scratch of its own.
 * bench.py quotes the product kernel of pyglm/regression.py:251-252 against this number as well as against the nominal peak, because the
 * boxes of a pool differ by several per cent under the power limit. */
int pgl_ubench_mfma(int kind, double seconds, double* rate_out, double* ms_out, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
