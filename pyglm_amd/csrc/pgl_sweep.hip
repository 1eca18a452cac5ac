// pgl_sweep: one Gibbs sweep of a shard's regressions as ONE call -- the loop `for n in range(N): regressions[n].resample(...)` of
// pyglm/models.py:169-171 with the body pyglm/regression.py:265-280, queued on a stream without a single host synchronisation:
//
//   pack a*W -> activation (regression.py:195-201) -> PG draw, kappa, log-likelihood (:491-511) -> border sums (:253-260) ->
//   deterministic rows (:153-155, 274-275) -> per batch of neurons: omega-weighted Gram (:251-252; fp64 MFMA or exact int8 residue
//   planes), posterior assembly (:210-223, 270-271), collapsed flips on the sweep tableau (:282-320, 343-378), weight draw (:323-340).
//
// Everything the host used to decide between launches is decided on the device (the pivot lists of the initial tableau sweep are built
// from a and perm by pivot_list_kernel; the weight draw runs to the largest possible active size and stops per neuron at its own), so a
// binder in any language gets the whole sweep from include/pyglm_hip.h, and pyglm_amd/engine.py is a thin caller of this function.
#include "pgl_common.h"
#include "../../include/pyglm_hip.h"
#include <cmath>
#include <tuple>
#include <vector>

namespace {

__global__ __launch_bounds__(256) void pack_weights_kernel(const int* __restrict__ a, const double* __restrict__ W, const double* __restrict__ b,
                                                           double* __restrict__ Wt, double* __restrict__ bias, int N, int B, int nloc, int Dp, int ldn) {
    // Wt[d][n] = a[n][d / B] * W[n][d]  (k-major operand of the activation contraction); zero in the padding rows and columns
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < nloc) bias[e] = b[e];
    if (e >= (long)Dp * ldn) return;
    const int d = (int)(e / ldn), n = (int)(e % ldn), D = N * B;
    double v = 0.0;
    if (d < D && n < nloc && a[(long)n * N + d / B]) v = W[(long)n * D + d];
    Wt[e] = v;
}

// regression.py:153-155 / 274-275: a row whose rho are all within 1e-6 of 0 or 1 is not sampled: a = round(rho), skip = 1
__global__ __launch_bounds__(256) void deterministic_rows_kernel(const double* __restrict__ rho, int* __restrict__ a, int* __restrict__ skip, int N) {
    const int n = blockIdx.x;
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    int bad = 0;
    for (int m = threadIdx.x; m < N; m += 256) {
        const double r = rho[(long)n * N + m];
        if (!((r < 1e-6) || (r > 1.0 - 1e-6))) bad = 1;
    }
    if (bad) s_any = 1;
    __syncthreads();
    const int det = !s_any;
    if (det)
        for (int m = threadIdx.x; m < N; m += 256) a[(long)n * N + m] = rho[(long)n * N + m] > 0.5 ? 1 : 0;
    if (threadIdx.x == 0) skip[n] = det;
}

__global__ __launch_bounds__(256) void gather_table_kernel(const double* __restrict__ table, const int* __restrict__ label, double* __restrict__ out, long n) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < n) out[e] = table[label[e]];
}

// out[e] (+)= partial[0][e] + partial[1][e] + ... in slice order (the split contraction of the border sums)
__global__ __launch_bounds__(256) void sum_slices_kernel(const double* __restrict__ partial, long n, int S, double* __restrict__ out, int accumulate,
                                                         int ld, int ncols) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n || e % ld >= ncols) return;                 // (the slices' padding columns were never written)
    double acc = accumulate ? out[e] : 0.0;
    for (int k = 0; k < S; ++k) acc += partial[(long)k * n + e];
    out[e] = acc;
}

__global__ __launch_bounds__(256) void fill_kernel(double* __restrict__ p, long n, double v) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < n) p[e] = v;
}

const char* const STAGE_NAMES[PGL_NSTAGES] = {"activation", "pg_loglik", "border", "gram", "gram.stats", "gram.planes", "gram.int8", "gram.crt",
                                              "gram_scale", "assemble", "flips", "flips.init", "flips.decide", "flips.apply", "weights", "pack"};
enum { ST_ACT = 0, ST_PG, ST_BORDER, ST_GRAM, ST_STATS, ST_PLANES, ST_I8, ST_CRT, ST_GSCALE, ST_ASM, ST_FLIPS, ST_FINIT, ST_FDEC, ST_FAPP, ST_W, ST_PACK };

struct Pending { std::vector<std::tuple<int, hipEvent_t, hipEvent_t, double>> ev; };

struct Clock {     // HIP events around a stage, on the launch stream; inert when the caller passed no pgl_stage_times_t
    pgl_stage_times_t* t; hipStream_t st;
    struct Mark { int stage; hipEvent_t e0; double work; bool on; };
    Mark tic(int stage, double work = 0.0) {
        Mark m{stage, nullptr, work, false};
        if (!t || (t->mask && !((t->mask >> stage) & 1u))) return m;
        if (hipEventCreate(&m.e0) != hipSuccess) return m;
        (void)hipEventRecord(m.e0, st);
        m.on = true;
        return m;
    }
    void toc(const Mark& m) {
        if (!m.on) return;
        hipEvent_t e1;
        if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(m.e0); return; }
        (void)hipEventRecord(e1, st);
        if (!t->pending) t->pending = new Pending();
        static_cast<Pending*>(t->pending)->ev.emplace_back(m.stage, m.e0, e1, m.work);
    }
};

inline int r_up(long x, int m) { return (int)((x + m - 1) / m * m); }

#define RC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

}  // namespace

extern "C" {

const char* pgl_stage_name(int i) { return (i >= 0 && i < PGL_NSTAGES) ? STAGE_NAMES[i] : nullptr; }

int pgl_sweep_dims(int N, int B, int nloc, int* Dp, int* ldn, int* ldj) {
    PGL_CHECK_ARG(N > 0 && B > 0 && nloc > 0);
    const long D = (long)N * B;
    if (Dp) *Dp = r_up(D + 1, 16);
    if (ldn) *ldn = r_up(nloc, 2);
    if (ldj) *ldj = r_up(D + 2, 16);
    return PGL_OK;
}

int pgl_stage_times_collect(pgl_stage_times_t* t) {
    PGL_CHECK_ARG(t != nullptr);
    Pending* p = static_cast<Pending*>(t->pending);
    if (!p) return PGL_OK;
    int rc = PGL_OK;
    for (auto& e : p->ev) {
        float ms = 0.f;
        const hipEvent_t e0 = std::get<1>(e), e1 = std::get<2>(e);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { pgl_set_error("stage timing: event query failed"); rc = PGL_ERR_HIP; }
        else { const int s = std::get<0>(e); t->ms[s] += ms; t->calls[s] += 1; t->work[s] += std::get<3>(e); }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    delete p;
    t->pending = nullptr;
    return rc;
}

int pgl_sweep(const pgl_sweep_t* s, uint64_t seed, uint64_t sweep, void* hip_stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    PGL_CHECK_ARG(s && s->N > 0 && s->B > 0 && s->B <= 32 && s->nloc > 0 && s->nb > 0 && s->n0 >= 0 && s->ndatasets > 0 && s->datasets);
    PGL_CHECK_ARG(s->obs >= 0 && s->obs <= 2 && (s->obs != 1 || s->xi > 0));
    PGL_CHECK_ARG(s->a && s->W && s->b && s->rho && s->Jw && s->hw && s->Jb && s->hb && s->c0 && s->perm && s->u && s->z && s->ll && s->status);
    PGL_CHECK_ARG(s->Wt && s->bias && s->border && s->skip && s->Jbuf && s->Mtab && s->Ac && s->hc && s->Tinv && s->G && s->Lws && s->Ut && s->Wt_ws);
    PGL_CHECK_ARG(s->d_idx && s->d_sign && s->d_cnt && s->batch_k && s->act && s->na && (s->label == nullptr || s->c0_dense != nullptr));
    PGL_CHECK_ARG(s->obs != 2 || (s->inv_eta && s->G0));
    PGL_CHECK_ARG(s->nrun >= 0 && s->nfirst >= 0 && s->nfirst % 2 == 0 && s->nfirst + s->nrun <= s->nloc && (s->nrun > 0 || s->nfirst == 0));
    PGL_CHECK_ARG(s->i8_slice >= 0 && s->i8_slice % 64 == 0);
    const int N = s->N, B = s->B, nloc = s->nloc;
    const int nrun = s->nrun > 0 ? s->nrun : nloc;        // neurons actually swept: [nf, nf + nrun) of the shard (the array layouts stay the shard's)
    const int nf = s->nfirst;
    const int nb = s->nb < nrun ? s->nb : nrun;
    const long D = (long)N * B;
    const int Dp = r_up(D + 1, 16), ldn = r_up(nloc, 2), ldj = r_up(D + 2, 16);
    const long strideJ = (long)ldj * ldj;
    const int R = pgl_k_flip_window_blocks(B);
    const int kmax = pgl_k_flip_kmax();
    if (R < 1) { pgl_set_error("B=%d too large for the proposal window", B); return PGL_ERR_ARG; }
    bool any_i8 = false;
    for (int i = 0; i < s->ndatasets; ++i) {
        const pgl_dataset_t& d = s->datasets[i];
        PGL_CHECK_ARG(d.T > 0 && d.Tp >= d.T && d.Tp % 16 == 0 && d.X && d.Xt && d.Y && d.Psi && d.OK && d.llpart);
        PGL_CHECK_ARG(!d.int8 || (d.sA && (d.PA || s->i8_PAs) && (d.planes > 0 || s->planes > 0)));      // (no resident X planes: converted per slice -- or once per group for the whole data set -- into i8_PAs)
        any_i8 = any_i8 || d.int8;
    }
    PGL_CHECK_ARG(!any_i8 || (s->i8_PB && s->i8_R && s->i8_stat && s->i8_group >= 1 && s->i8_group <= PGL_I8_MAX_GROUP && s->obs != 2));
    Clock clk{s->times, st};

    // ---- activation of the whole shard, PG draw / kappa / log-likelihood (regression.py:195-201, 491-511)
    {
        auto m = clk.tic(ST_PACK);
        const long tot = (long)Dp * ldn;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, s->a, s->W, s->b, s->Wt, s->bias, N, B, nloc, Dp, ldn);
        PGL_CHECK_LAUNCH();
        clk.toc(m);
    }
    for (int i = 0; i < s->ndatasets; ++i) {
        const pgl_dataset_t& d = s->datasets[i];
        auto m = clk.tic(ST_ACT, 2.0 * d.T * D * nrun);
        {   // Psi[t][n] = sum_d Xt[d][t] Wt[d][n] for the neurons [nf, nf + nrun)  (pgl_activation, with the operand's column count taken from nf on)
            PglGemmArgs q{};
            q.A = d.Xt; q.lda = d.Tp; q.a_cols = d.Tp;
            q.B = s->Wt + nf; q.ldb = ldn; q.b_cols = ldn - nf;
            q.C = d.Psi + nf; q.ldc = ldn;
            q.M = d.T; q.N = nrun; q.K = Dp; q.nbatch = 1; q.alpha = 1.0; q.beta = 0.0; q.tri = 0;
            RC(pgl_launch_gemm(PGL_GEMM_PLAIN, q, st));
        }
        clk.toc(m);
        m = clk.tic(ST_PG, (double)d.T * nrun);
        if (s->obs == 2) RC(pgl_k_gaussian_stats(d.Psi + nf, ldn, s->bias + nf, d.Y + nf, ldn, s->inv_eta + nf, d.OK + nf, 2 * ldn, d.OK + ldn + nf, 2 * ldn, d.llpart,
                                                 s->ll + nf, i > 0, d.T, nrun, st));
        else RC(pgl_k_pg_loglik(d.Psi + nf, ldn, s->bias + nf, d.Y + nf, ldn, d.OK + nf, 2 * ldn, d.OK + ldn + nf, 2 * ldn, d.llpart, s->ll + nf, i > 0, d.T, nrun,
                                s->obs, s->xi, seed, sweep, (uint64_t)(s->n0 + nf), d.elem0, st));
        clk.toc(m);
        if (d.omega_override) {       // test hook: the reference fixtures inject omega
            if (hipMemcpy2DAsync(d.OK, (size_t)2 * ldn * sizeof(double), d.omega_override, (size_t)nloc * sizeof(double), (size_t)nloc * sizeof(double),
                                 (size_t)d.T, hipMemcpyDeviceToDevice, st) != hipSuccess) { pgl_set_error("omega override copy failed"); return PGL_ERR_HIP; }
        }
    }
    // ---- border sums [Omega|Kappa]' [X, 1]  (regression.py:253-260)
    for (int i = 0; i < s->ndatasets; ++i) {
        const pgl_dataset_t& d = s->datasets[i];
        auto m = clk.tic(ST_BORDER, 4.0 * d.T * (D + 1) * nrun);
        // The output is small (2 nloc x (D+1)) and the contraction long (T): with few neurons it is a handful of 128 x 256 tiles -- 6
        // workgroups at N = 128, 42 on a 128-neuron shard of cfg3 -- each walking all of T.  So T is cut into S slices, one batch of the
        // GEMM each, whose partial sums go to the (still unused) J buffer and are added up in slice order by one small kernel.
        // (a partial run -- nrun < nloc -- contracts the Omega and the Kappa columns of its neurons as two pieces)
        const int npieces = nrun < nloc ? 2 : 1;
        const int Mp = nrun < nloc ? r_up(nrun, 2) : 2 * ldn;
        // The number of slices follows from T and D ALONE (up to 64 slices of >= 256 bins; for narrow models as many as any shard's J buffer
        // is sure to hold: ldj^2 / (4 Dp) <= nloc ldj^2 / (2 ldn Dp)), not from how many neurons this shard has: a neuron's sums are then
        // added up in the same order whatever the sharding, and its whole sweep comes out the same to the last bit on 1 GPU or on 8.
        // A narrow model's J buffer holds only a few slices (D = 4: four) -- BASELINE configs[0] then walks 2 500 bins per workgroup, a third
        // of a millisecond of its 0.8 ms sweep.  Its partial sums go to the flips' pivot-block buffer G instead ([nb][kmax][kmax], idle here),
        // whose capacity for ANY shard, kmax^2 / (4 Dp), exceeds 64 slices up to D ~ 1000.
        long S = 64;
        if (S > d.Tp / 256) S = d.Tp / 256;
        double* scratch = s->Jbuf;
        long cap_any = (long)ldj * ldj / (4L * Dp), scratch_doubles = (long)nb * strideJ;      // slices any shard's buffer holds / doubles this one's has
        if (cap_any < S && (long)kmax * kmax / (4L * Dp) > cap_any) {
            scratch = s->G;
            cap_any = (long)kmax * kmax / (4L * Dp);
            scratch_doubles = (long)nb * kmax * kmax;
        }
        if (S > cap_any) S = cap_any;
        // S is now a function of T and D ALONE.  A shard swept in small batches (a user's batch << nloc at small D) may not hold S slices of the
        // WHOLE border (2 ldn rows) in its scratch: the neurons then go through in column groups of as many as do fit -- never fewer than four
        // (cap_any) --, each neuron's sums still added in the same slice order (S used to be clamped by this shard's capacity, which
        // made the last bits depend on the batch size)
        if (S >= 2) {
            const int chunk = (int)(d.Tp / 16 / S) * 16, rem = d.Tp - (int)S * chunk;
            long cols_fit = scratch_doubles / (S * Dp) / 2 * 2;                  // border rows (neuron columns of Omega or of Kappa) whose S slices fit
            const bool whole = npieces == 1 && cols_fit >= 2L * ldn;            // the usual case: [Omega | Kappa] of the whole shard as one product
            const int cols = nrun < nloc ? r_up(nrun, 2) : ldn;                 // columns of Omega (and of Kappa) to contract
            const int ngroups_cols = whole ? 2 * ldn : (int)(cols_fit < cols ? cols_fit : cols);
            for (int piece = 0; piece < (whole ? 1 : 2); ++piece) {
                const int first = whole ? 0 : nf, count = whole ? 2 * ldn : cols;
                for (int c0 = 0; c0 < count; c0 += ngroups_cols) {
                    const int mcols = ngroups_cols < count - c0 ? ngroups_cols : count - c0;
                    const double* Ap = d.OK + (long)piece * ldn + first + c0;                    // columns of Omega, then of Kappa
                    const long crow = ((long)piece * ldn + first + c0) * Dp;                     // rows of the border: omega sums, then kappa sums
                    const long partg = (long)mcols * Dp;
                    PglGemmArgs a{};
                    a.A = Ap; a.lda = 2 * ldn; a.strideA = (long)chunk * a.lda; a.a_cols = mcols;
                    a.B = d.X; a.ldb = Dp; a.strideB = (long)chunk * Dp; a.b_cols = Dp;
                    a.C = scratch; a.ldc = Dp; a.strideC = partg;
                    a.M = mcols; a.N = (int)D + 1; a.K = chunk; a.nbatch = (int)S; a.alpha = 1.0; a.beta = 0.0; a.tri = 0;
                    RC(pgl_launch_gemm(PGL_GEMM_PLAIN, a, st));
                    if (rem > 0) {           // the last rem < 16 S rows: onto the first slice's sums
                        a.A = Ap + (long)S * chunk * a.lda; a.B = d.X + (long)S * chunk * Dp; a.K = rem; a.nbatch = 1; a.beta = 1.0;
                        RC(pgl_launch_gemm(PGL_GEMM_PLAIN, a, st));
                    }
                    hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((partg + 255) / 256)), dim3(256), 0, st, scratch, partg, (int)S, s->border + crow, i > 0, Dp,
                                       (int)D + 1);
                    PGL_CHECK_LAUNCH();
                }
            }
        } else {
            for (int piece = 0; piece < npieces; ++piece) {
                const double* Ap = d.OK + (long)piece * ldn + nf;            // columns [nf, nf + Mp) of Omega, then of Kappa
                const long crow = ((long)piece * ldn + nf) * Dp;             // rows of the border: omega sums, then kappa sums
                RC(pgl_contract_tn(Ap, 2 * ldn, Mp, d.X, Dp, Dp, s->border + crow, Dp, Mp, (int)D + 1, d.Tp, 1.0, i > 0 ? 1.0 : 0.0, st));
            }
        }
        clk.toc(m);
    }
    // ---- deterministic rows, status, optional log-odds record, block-prior constants
    hipLaunchKernelGGL(deterministic_rows_kernel, dim3(nloc), dim3(256), 0, st, s->rho, s->a, s->skip, N);
    PGL_CHECK_LAUNCH();
    if (hipMemsetAsync(s->status, 0, (size_t)nloc * sizeof(int), st) != hipSuccess) { pgl_set_error("memset failed"); return PGL_ERR_HIP; }
    if (s->logodds) {
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long)nloc * N + 255) / 256)), dim3(256), 0, st, s->logodds, (long)nloc * N, (double)NAN);
        PGL_CHECK_LAUNCH();
    }
    const double* c0 = s->c0;
    if (s->label) {
        hipLaunchKernelGGL(gather_table_kernel, dim3((unsigned)(((long)nloc * N + 255) / 256)), dim3(256), 0, st, s->c0, s->label, s->c0_dense, (long)nloc * N);
        PGL_CHECK_LAUNCH();
        c0 = s->c0_dense;
    }

    for (int s0 = nf; s0 < nf + nrun; s0 += nb) {
        const int nbb = nb < nf + nrun - s0 ? nb : nf + nrun - s0;
        // ---- likelihood Gram of neurons [s0, s0 + nbb)  (regression.py:251-252)
        if (s->obs == 2) {
            auto m = clk.tic(ST_GSCALE, 8.0 * nbb * D * (D + 1) / 2);
            RC(pgl_k_scaled_gram(s->G0, ldj, s->inv_eta + s0, s->Jbuf, ldj, strideJ, (int)D, nbb, st));
            clk.toc(m);
        } else {
            for (int i = 0; i < s->ndatasets; ++i) {
                const pgl_dataset_t& d = s->datasets[i];
                if (!d.int8) {
                    auto m = clk.tic(ST_GRAM, (double)nbb * d.T * D * (D + 1));
                    // A small model (a few 128 x 128 tiles per neuron) with a long recording is a handful of workgroups each walking all of T --
                    // BASELINE configs[0]: 4 items, 1.2 of the sweep's 2 ms.  Then T is cut into up to 64 slices of >= 256 bins, one work item
                    // each, whose sums land in the flips' (idle) pivot-block buffer G ([nb][kmax][kmax]: kmax^2 / ldj^2 slices per neuron --
                    // 1024 at D = 4, 12 at D = 128, 1 from D = 350) and are added in slice order.  The number of slices follows from T and D
                    // alone -- not from the number of neurons -- so a shard gets the bits of the whole.
                    long S = d.Tp / 256;
                    if (S > (long)kmax * kmax / ((long)ldj * ldj)) S = (long)kmax * kmax / ((long)ldj * ldj);
                    if (S > 64) S = 64;
                    if (D <= 512 && S >= 2) {
                        const long ks = r_up((d.Tp + S - 1) / S, 16);
                        RC(pgl_gram_split(d.X, Dp, Dp, d.OK + s0, 2 * ldn, d.Tp, (int)D, nbb, s->Jbuf, ldj, strideJ, i > 0, ks, s->G, (long)kmax * kmax, st));
                    } else RC(pgl_weighted_gram(d.X, Dp, Dp, d.OK + s0, 2 * ldn, d.Tp, (int)D, nbb, s->Jbuf, ldj, strideJ, i > 0, st));
                    clk.toc(m);
                    continue;
                }
                const int G = s->i8_group, np = d.planes > 0 ? d.planes : s->planes;
                double* amax = s->i8_stat;
                double* ss = s->i8_stat + (long)G * D;
                double* sB = s->i8_stat + 2L * G * D;
                // The scales of omega_n X need the norm of each of its columns.  With d.xmax and s->i8_norm those come for the WHOLE batch from one
                // fp64 MFMA contraction of the squared operands, ss[n][j] = sum_t omega_nt^2 x_tj^2 -- X is read once per batch instead of once
                // per group of 8 neurons (215 ms of column-statistics passes per sweep at BASELINE configs[2]) --, T cut into slices whose partial
                // sums borrow the (idle) residue buffer and are added in slice order, like the border sums above.
                const bool batch_norms = d.xmax && s->i8_norm && s0 % 2 == 0;
                const long part = (long)r_up(s->nb, 2) * Dp;
                double* ssb = s->i8_norm;                          // [nb rounded up to even][Dp] sums of squares of this batch's columns
                double* ommax = s->i8_norm + part;                 // [nloc] largest omega of every local neuron
                if (batch_norms) {
                    auto m = clk.tic(ST_STATS, 8.0 * d.T * D);
                    RC(pgl_k_i8_colmax(d.OK + s0, 2 * ldn, d.T, nbb, ommax + s0, st));
                    const int Mp = r_up(nbb, 2);
                    const size_t r_bytes = (size_t)(G < nbb ? G : nbb) * np * pgl_k_i8_padded_rows((int)D) * pgl_k_i8_padded_rows((int)D);
                    long S = 64;                                   // (from T alone, like the border sums: the same norms -- hence scales -- whatever the shard)
                    if (S > d.Tp / 256) S = d.Tp / 256;
                    if (S > (long)(r_bytes / sizeof(double)) / part) S = (long)(r_bytes / sizeof(double)) / part;
                    PglGemmArgs q{};
                    q.A = d.OK + s0; q.lda = 2 * ldn; q.a_cols = Mp;
                    q.B = d.X; q.ldb = Dp; q.b_cols = Dp;
                    q.M = Mp; q.N = (int)D; q.alpha = 1.0; q.beta = 0.0; q.tri = 0; q.ldc = Dp;
                    if (S >= 2) {
                        const int chunk = (int)(d.Tp / 16 / S) * 16, rem = d.Tp - (int)S * chunk;
                        double* partial = reinterpret_cast<double*>(s->i8_R);
                        q.strideA = (long)chunk * q.lda; q.strideB = (long)chunk * Dp; q.C = partial; q.strideC = part; q.K = chunk; q.nbatch = (int)S;
                        RC(pgl_launch_gemm(PGL_GEMM_SQUARES, q, st));
                        if (rem > 0) {           // the last rem < 16 S rows: onto the first slice's sums
                            q.A += (long)S * chunk * q.lda; q.B += (long)S * chunk * Dp; q.K = rem; q.nbatch = 1; q.beta = 1.0;
                            RC(pgl_launch_gemm(PGL_GEMM_SQUARES, q, st));
                        }
                        hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((part + 255) / 256)), dim3(256), 0, st, partial, part, (int)S, ssb, 0, Dp, (int)D);
                        PGL_CHECK_LAUNCH();
                    } else {
                        q.C = ssb; q.K = d.Tp; q.nbatch = 1;
                        RC(pgl_launch_gemm(PGL_GEMM_SQUARES, q, st));
                    }
                    clk.toc(m);
                }
                for (int g0 = 0; g0 < nbb; g0 += G) {
                    const int gz = G < nbb - g0 ? G : nbb - g0;
                    const double* om = d.OK + s0 + g0;
                    auto m = clk.tic(ST_STATS, batch_norms ? 0.0 : 8.0 * d.T * D);
                    if (batch_norms) {
                        RC(pgl_k_i8_scales_bound(ssb + (long)g0 * Dp, Dp, ommax + s0 + g0, d.xmax, (int)D, gz, d.T, np, sB, st));
                    } else {
                        // one pass over X per 8 neurons (the statistics pass borrows the residue buffer for its per-chunk partials: the previous
                        // group's CRT has read it, this group's products have not written it yet)
                        const size_t r_bytes = (size_t)gz * np * pgl_k_i8_padded_rows((int)D) * pgl_k_i8_padded_rows((int)D);
                        for (int c0 = 0; c0 < gz; c0 += 8) {
                            const int cz = 8 < gz - c0 ? 8 : gz - c0;
                            if (pgl_k_i8_stats_scratch_doubles((int)D, cz) * sizeof(double) <= r_bytes)
                                RC(pgl_k_i8_colstats_scales(d.X, Dp, om + c0, 2 * ldn, d.T, (int)D, cz, np, reinterpret_cast<double*>(s->i8_R), sB + (long)c0 * D, st));
                            else {
                                RC(pgl_k_i8_colstats(d.X, Dp, om + c0, 2 * ldn, d.T, (int)D, cz, amax, ss, st));
                                RC(pgl_k_i8_scales(amax, ss, (long)cz * D, d.T, np, sB + (long)c0 * D, st));
                            }
                        }
                    }
                    clk.toc(m);
                    // time slices (BASELINE configs[4]: one neuron's planes are 86 GB): the integer Gram is a sum over time, so the
                    // slices' products add up in the residues; a data set without resident X planes converts those per slice too
                    const int slice = s->i8_slice > 0 && s->i8_slice < d.T ? s->i8_slice : d.T;
                    for (int t0 = 0; t0 < d.T; t0 += slice) {
                        const int ts = slice < d.T - t0 ? slice : d.T - t0;
                        m = clk.tic(ST_PLANES, (double)np * (gz + (d.PA ? 0 : 1)) * ts * D);
                        if (!d.PA) RC(pgl_k_i8_planes(d.Xt + t0, d.Tp, 1, nullptr, 0, d.sA, static_cast<int8_t*>(s->i8_PAs), ts, (int)D, 1, np, t0, st));
                        RC(pgl_k_i8_planes(d.Xt + t0, d.Tp, 1, om + (long)t0 * 2 * ldn, 2 * ldn, sB, static_cast<int8_t*>(s->i8_PB), ts, (int)D, gz, np, t0, st));   // (coalesced rows of Xt)
                        clk.toc(m);
                        m = clk.tic(ST_I8, (double)gz * ts * D * (D + 1));
                        if (d.PA) RC(pgl_k_i8_gram(static_cast<const int8_t*>(d.PA), pgl_k_i8_kp(d.T), t0 / 64, static_cast<const int8_t*>(s->i8_PB),
                                                   static_cast<int8_t*>(s->i8_R), static_cast<int8_t*>(s->i8_Rx), ts, (int)D, gz, np, t0 > 0, st));
                        else RC(pgl_k_i8_gram(static_cast<const int8_t*>(s->i8_PAs), 0, 0, static_cast<const int8_t*>(s->i8_PB), static_cast<int8_t*>(s->i8_R),
                                              static_cast<int8_t*>(s->i8_Rx), ts, (int)D, gz, np, t0 > 0, st));
                        clk.toc(m);
                    }
                    m = clk.tic(ST_CRT, (double)np * gz * D * (D + 1) / 2);
                    RC(pgl_k_i8_crt(static_cast<const int8_t*>(s->i8_R), static_cast<const int8_t*>(s->i8_Rx), d.sA, sB, s->Jbuf + (long)g0 * strideJ, ldj, strideJ, (int)D, gz, np, i > 0, st));
                    clk.toc(m);
                }
            }
        }
        // ---- posterior assembly (regression.py:210-223, 253-260, 270-271)
        {
            auto m = clk.tic(ST_ASM);
            const double* jw = s->label ? s->Jw : s->Jw + (long)s0 * N * B * B;
            const double* hw = s->label ? s->hw : s->hw + (long)s0 * N * B;
            const int* lab = s->label ? s->label + (long)s0 * N : nullptr;
            RC(pgl_k_assemble_post(s->Jbuf, ldj, strideJ, s->border + (long)s0 * Dp, s->border + (long)(ldn + s0) * Dp, Dp, jw, hw, lab, s->Jb + s0, s->hb + s0,
                                   nbb, N, B, st));
            clk.toc(m);
        }
        // ---- a small model (a tableau of at most 98 rows): flips and weight draw as ONE launch with the tableau in LDS (pgl_small.hip).  Which
        // path a model takes follows from N and B alone (and the engine-wide visit_order option), never from the shard or the batch
        if (s->visit_order && pgl_k_small_fits(N, B)) {
            auto ms = clk.tic(ST_FLIPS);
            RC(pgl_k_small_tail(s->Jbuf, ldj, strideJ, nbb, N, B, s->perm + (long)s0 * N, s->u + (long)s0 * N, s->rho + (long)s0 * N, c0 + (long)s0 * N,
                                s->a + (long)s0 * N, s->skip + s0, s->z + (long)s0 * (D + 1), D + 1, s->W + (long)s0 * D, s->b + s0, s->status + s0,
                                s->logodds ? s->logodds + (long)s0 * N : nullptr, st));
            clk.toc(ms);
            continue;
        }
        // ---- collapsed flips (regression.py:282-320)
        auto mf = clk.tic(ST_FLIPS);
        if (!s->all_deterministic) {
            PglFlipState fs{s->Mtab, ldj, strideJ, nbb, N, B, s->perm + (long)s0 * N, s->u + (long)s0 * N, s->rho + (long)s0 * N, c0 + (long)s0 * N,
                            s->a + (long)s0 * N, s->skip + s0, s->d_idx, s->d_sign, s->d_cnt, s->batch_k, s->G, s->Lws, s->Ut, s->Wt_ws, ldj, s->status + s0,
                            s->visit_order ? 1 : 0, s->logodds ? s->logodds + (long)s0 * N : nullptr};
            if (s->visit_order) RC(pgl_k_flip_permute(fs, s->Jbuf, ldj, strideJ, st));
            else if (hipMemcpyAsync(s->Mtab, s->Jbuf, (size_t)nbb * strideJ * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) {
                pgl_set_error("tableau copy failed"); return PGL_ERR_HIP;
            }
            // initial sweep on S0 = {bias} U {active blocks}, in chunks of 512 pivots (rank-512 passes); the lists come from the device
            RC(pgl_k_flip_pivot_list(fs, s->act, D + 1, s->na, st));
            const int ck = kmax;
            long rows = s->init_rows_bound > 0 && s->init_rows_bound <= D + 1 ? s->init_rows_bound : D + 1;
            for (int c = 0; (long)c * ck < rows; ++c) {
                // (the extent of a chunk's launches from D alone: with the hint in it, the blocking of the pivot-block inverse -- and with it the
                // rounding of the log-odds, in the 13th digit -- followed the most active neuron of the batch, i.e. the sharding)
                const long left = (D + 1) - (long)c * ck;
                auto m = clk.tic(ST_FINIT);
                RC(pgl_k_flip_pivot_chunk(fs, s->act, D + 1, s->na, c, ck, st));
                RC(pgl_k_flip_apply(fs, 0, (int)(left < ck ? left : ck), -1, st));
                clk.toc(m);
            }
            const int nwin = (N + R - 1) / R;
            // windows in pairs: one pass over the trailing tableau for two windows' panels (pgl_k_flip_apply_pair); its four counters per
            // neuron live in the pivot-list buffer, which is free between the initial sweep and the weight draw
            const bool pair = s->visit_order && !s->flip_single_pass;
            for (int w = 0; w < nwin; ++w) {
                auto m = clk.tic(ST_FDEC);
                RC(pgl_k_flip_decide(fs, w, st));
                clk.toc(m);
                m = clk.tic(ST_FAPP);
                if (!pair || (w % 2 == 0 && w + 1 >= nwin)) RC(pgl_k_flip_apply(fs, 1, 0, w, st));
                else RC(pgl_k_flip_apply_pair(fs, w % 2, w, s->act, st));
                clk.toc(m);
            }
        }
        clk.toc(mf);
        // ---- weights (regression.py:323-340); runs to the largest possible active size, every neuron stops at its own
        auto mw = clk.tic(ST_W);
        PglCholState cs{s->Jbuf, ldj, strideJ, s->a + (long)s0 * N, s->act, D + 1, s->na, s->Ac, ldj, strideJ, s->hc, s->Tinv, s->z + (long)s0 * (D + 1), D + 1,
                        s->W + (long)s0 * D, s->b + s0, nbb, N, B, s->status + s0};
        RC(pgl_k_chol_index(cs, st));
        long na_bound = s->active_rows_bound > 0 && s->active_rows_bound <= D + 1 ? s->active_rows_bound : D + 1;
        RC(pgl_k_chol_sample(cs, (int)na_bound, st));
        clk.toc(mw);
    }
    return PGL_OK;
}

int pgl_get_state(const pgl_sweep_t* s, int* a_host, double* W_host, double* b_host, double* ll_host, int* status_host, void* hip_stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    PGL_CHECK_ARG(s && s->nloc > 0);
    const long D = (long)s->N * s->B;
    auto cp = [&](void* dst, const void* src, size_t bytes) { return dst == nullptr || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st) == hipSuccess; };
    bool ok = cp(a_host, s->a, (size_t)s->nloc * s->N * sizeof(int)) && cp(W_host, s->W, (size_t)s->nloc * D * sizeof(double)) &&
              cp(b_host, s->b, (size_t)s->nloc * sizeof(double)) && cp(ll_host, s->ll, (size_t)s->nloc * sizeof(double)) &&
              cp(status_host, s->status, (size_t)s->nloc * sizeof(int));
    if (!ok || hipStreamSynchronize(st) != hipSuccess) { pgl_set_error("pgl_get_state: copy failed: %s", hipGetErrorString(hipGetLastError())); return PGL_ERR_HIP; }
    return PGL_OK;
}

}  // extern "C"
