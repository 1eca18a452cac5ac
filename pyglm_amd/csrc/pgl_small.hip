// Collapsed flips + weight draw of a SMALL model as ONE launch: one workgroup per neuron, the whole sweep tableau in LDS.
//
// pyglm/regression.py:282-320 (`_collapsed_resample_a`: N sequential Bernoulli proposals, each a difference of two marginal likelihoods,
// :343-378) and :323-340 (`_resample_W`: [W_active; b] ~ N(J^-1 h, J^-1)).  The general path (pgl_flips.hip, pgl_chol.hip) is built for
// tableaus of thousands of rows -- pivot chunks of 512, proposal windows, rank-k passes over HBM -- and spends ~25 launches per batch on a
// model whose tableau is a few hundred bytes: at BASELINE configs[0] (N = 4, B = 1: a 6 x 6 tableau) those launches were 0.2 of the 0.65 ms
// of a sweep.  Here, for D + 2 <= PGL_SMALL_MAX rows, a neuron's tableau M = sweep([[J, h], [h', 0]], S) lives in LDS for its whole life:
//   * initial sweep on S0 = {bias} U {active blocks}, pivot by pivot (Gauss-Jordan: a_pp -> -1/d, a_ip -> a_ip/d, a_ij -= a_ip a_pj / d);
//   * the proposals in the order perm: block m's log-odds from its B x B diagonal block and its entries of the h column, exactly as
//     decide_kernel forms them (inactive: -1/2 log|S_m| + 1/2 r' S_m^-1 r + c0; active: +1/2 log|P_mm| + 1/2 mu' P_mm^-1 mu + c0; + log rho -
//     log(1 - rho) = lps[1] - lps[0] of :293-307), the draw as `sample_discrete_from_log` on the same uniform, a flip as B forward (switch
//     on) or reverse (switch off) sweeps;
//   * the weight draw on the final active set: J_SS gathered from the posterior (global), Cholesky, w = L^-T (L^-1 h + z)  (= J^-1 h + L^-T z,
//     the reference's sample_gaussian(J=, h=)), scattered into W and b.
// Decisions must equal the oracle's bit for bit and the weights to 1e-7 like the general path's (tests/test_gpu_parity.py runs every small
// case through this kernel); its last bits differ from the general path's, so WHICH path a model takes follows from D alone -- never from
// the shard or the batch.
#include "pgl_common.h"

namespace {

struct SmallArgs {
    const double* J; long ldj; long strideJ;       // assembled posterior per neuron: lower triangle, row D = bias, row D + 1 = h
    int N, B;
    const int* perm; const double* u; const double* rho; const double* c0;   // [nb][N] each (u indexed by proposal step)
    int* a; const int* skip;                      // [nb][N] in/out, [nb]
    const double* z; long ldz;                    // [nb][ldz] normals, the first na used
    double* W; double* b;                         // [nb][D], [nb]
    int* status; double* logodds;                 // [nb]; optional [nb][N] (by proposal step)
};

constexpr int SM_T = 256;

// forward (dir = +1) or reverse (dir = -1) sweep of the symmetric tableau A (Md x Md, leading dimension ld) on pivot p; cp: Md doubles of scratch
__device__ __forceinline__ bool sweep_pivot(double* A, int ld, int Md, int p, int dir, double* cp, int tid, bool want_positive) {
    for (int i = tid; i < Md; i += SM_T) cp[i] = A[i * ld + p];
    __syncthreads();
    const double d = cp[p];
    const bool ok = want_positive ? (d > 0.0) : (d < 0.0);
    const double inv = 1.0 / d;
    const double sgn = dir > 0 ? 1.0 : -1.0;
    for (int e = tid; e < Md * Md; e += SM_T) {
        const int i = e / Md, j = e - i * Md;
        double v;
        if (i == p && j == p) v = -inv;
        else if (i == p) v = sgn * cp[j] * inv;
        else if (j == p) v = sgn * cp[i] * inv;
        else v = A[i * ld + j] - cp[i] * cp[j] * inv;
        A[i * ld + j] = v;
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(SM_T) void small_tail_kernel(SmallArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int N = g.N, B = g.B, D = N * B, Md = D + 2, ld = Md | 1, hcol = D + 1;
    double* A = sm;                         // [Md][ld]
    double* cp = sm + (size_t)Md * ld;      // [Md]
    double* vec = cp + Md;                  // [Md]: h / y / w of the weight draw
    double* fac = vec + Md;                 // [B][B]
    int* act = reinterpret_cast<int*>(fac + B * B);    // [Md] active row list of the weight draw
    __shared__ int s_v, s_bad, s_na;
    const double* Jn = g.J + (long)n * g.strideJ;
    int* an = g.a + (long)n * N;
    if (tid == 0) s_bad = 0;
    // ---- tableau [[J, h], [h', 0]] from the lower triangle
    for (int e = tid; e < Md * Md; e += SM_T) {
        const int i = e / Md, j = e - i * Md;
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        A[i * ld + j] = (hi == hcol && lo == hcol) ? 0.0 : Jn[(long)hi * g.ldj + lo];
    }
    __syncthreads();
    if (!g.skip[n]) {
        // ---- initial sweep on {bias} U active blocks
        int bad = 0;
        for (int p = 0; p <= D; ++p) {
            const bool on = p == D || an[p / B] != 0;          // (block-uniform)
            if (on && !sweep_pivot(A, ld, Md, p, +1, cp, tid, true)) bad = 1;
        }
        if (tid == 0 && bad) s_bad |= 2;
        // ---- proposals, in order
        const int* perm = g.perm + (long)n * N;
        for (int k = 0; k < N; ++k) {
            const int m = perm[k], p0 = m * B;
            const int am = an[m];
            if (tid == 0) {
                const double sgn = am ? -1.0 : 1.0;
                double* Cb = fac;                  // B x B lower factor of +-(diagonal block), row-major
                bool ok = true;
                double logdet = 0.0;
                for (int i = 0; i < B; ++i)
                    for (int j = 0; j <= i; ++j) {
                        double s = sgn * A[(p0 + i) * ld + p0 + j];
                        for (int x = 0; x < j; ++x) s -= Cb[i * B + x] * Cb[j * B + x];
                        if (i == j) { if (!(s > 0.0)) { ok = false; s = 1.0; } Cb[i * B + i] = sqrt(s); logdet += log(s); }
                        else Cb[i * B + j] = s / Cb[j * B + j];
                    }
                double quad = 0.0;
                double* vb = vec;
                for (int i = 0; i < B; ++i) {
                    double s = A[(p0 + i) * ld + hcol];
                    for (int x = 0; x < i; ++x) s -= Cb[i * B + x] * vb[x];
                    vb[i] = s / Cb[i * B + i];
                    quad += vb[i] * vb[i];
                }
                const double dml = (am ? 0.5 * logdet : -0.5 * logdet) + 0.5 * quad + g.c0[(long)n * N + m];
                const double rho = g.rho[(long)n * N + m];
                int v;
                double lo = __builtin_nan("");
                if (rho == 0.0 || rho == 1.0) v = 0;        // reference :298/:307: 0 * log(0) = NaN reaches sample_discrete_from_log, which then returns 0
                else {
                    const double d = dml + log(rho) - log(1.0 - rho);
                    lo = d;
                    const double mx = d > 0.0 ? d : 0.0;
                    const double e0 = exp(-mx), e1 = exp(d - mx);
                    v = (g.u[(long)n * N + k] * (e0 + e1) > e0) ? 1 : 0;
                }
                if (!ok) s_bad |= 1;
                if (g.logodds) g.logodds[(long)n * N + k] = lo;
                s_v = v;
            }
            __syncthreads();
            const int v = s_v;
            if (v != am) {
                int bad2 = 0;
                for (int i = 0; i < B; ++i)
                    if (!sweep_pivot(A, ld, Md, p0 + i, v ? +1 : -1, cp, tid, v != 0)) bad2 = 1;
                if (tid == 0) { an[m] = v; if (bad2) s_bad |= 2; }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- weight draw on the final active set (regression.py:323-340): rows of the active blocks in ascending order, the bias last
    if (tid == 0) {
        int na = 0;
        for (int m = 0; m < N; ++m)
            if (an[m]) for (int i = 0; i < B; ++i) act[na++] = m * B + i;
        act[na++] = D;
        s_na = na;
    }
    __syncthreads();
    const int na = s_na;
    for (int e = tid; e < na * na; e += SM_T) {
        const int i = e / na, j = e - i * na;
        if (j <= i) A[i * ld + j] = Jn[(long)act[i] * g.ldj + act[j]];       // (act ascending: act[i] >= act[j])
    }
    for (int i = tid; i < na; i += SM_T) vec[i] = Jn[(long)hcol * g.ldj + act[i]];
    for (int i = tid; i < D; i += SM_T) g.W[(long)n * D + i] = 0.0;
    __syncthreads();
    // right-looking Cholesky (lower), column by column
    int cbad = 0;
    for (int k = 0; k < na; ++k) {
        double d = A[k * ld + k];
        if (!(d > 0.0)) { cbad = 1; d = 1.0; }
        const double r = sqrt(d);
        __syncthreads();
        for (int i = k + tid; i < na; i += SM_T) A[i * ld + k] = (i == k) ? r : A[i * ld + k] / r;
        __syncthreads();
        const int rem = na - k - 1;
        for (int e = tid; e < rem * rem; e += SM_T) {
            const int i = k + 1 + e / rem, j = k + 1 + e % rem;
            if (j <= i) A[i * ld + j] -= A[i * ld + k] * A[j * ld + k];
        }
        __syncthreads();
    }
    if (tid == 0) {
        // y = L^-1 h;  w = L^-T (y + z)
        const double* zn = g.z + (long)n * g.ldz;
        for (int i = 0; i < na; ++i) {
            double s = vec[i];
            for (int x = 0; x < i; ++x) s -= A[i * ld + x] * vec[x];
            vec[i] = s / A[i * ld + i];
        }
        for (int i = 0; i < na; ++i) vec[i] += zn[i];
        for (int i = na - 1; i >= 0; --i) {
            double s = vec[i];
            for (int x = i + 1; x < na; ++x) s -= A[x * ld + i] * vec[x];
            vec[i] = s / A[i * ld + i];
        }
        if (cbad) s_bad |= 4;
    }
    __syncthreads();
    for (int i = tid; i < na - 1; i += SM_T) g.W[(long)n * D + act[i]] = vec[i];
    if (tid == 0) {
        g.b[n] = vec[na - 1];
        if (s_bad) atomicOr(&g.status[n], s_bad);
    }
}

}  // namespace

int pgl_k_small_max_rows(void) { return 98; }       // D + 2 <= 98 (and B <= 16): the tableau, two vectors, a block factor and the index list fit 82 KiB of LDS
bool pgl_k_small_fits(int N, int B) { return (long)N * B + 2 <= pgl_k_small_max_rows() && B <= 16; }

int pgl_k_small_tail(const double* J, long ldj, long strideJ, int nb, int N, int B, const int* perm, const double* u, const double* rho, const double* c0,
                     int* a, const int* skip, const double* z, long ldz, double* W, double* b, int* status, double* logodds, hipStream_t st) {
    const int Md = N * B + 2;
    if (!pgl_k_small_fits(N, B)) { pgl_set_error("small tail: %d rows, B = %d (max %d rows, B <= 16)", Md, B, pgl_k_small_max_rows()); return PGL_ERR_ARG; }
    const size_t lds = ((size_t)Md * (Md | 1) + 2 * (size_t)Md + (size_t)B * B) * sizeof(double) + (size_t)Md * sizeof(int) + 16;
    static PglPerDeviceSize have;
    if (int rc = pgl_grow_dynamic_lds(reinterpret_cast<const void*>(small_tail_kernel), lds, have)) return rc;
    SmallArgs g{J, ldj, strideJ, N, B, perm, u, rho, c0, a, skip, z, ldz, W, b, status, logodds};
    hipLaunchKernelGGL(small_tail_kernel, dim3(nb), dim3(SM_T), lds, st, g);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}
