// Gaussian conditional of the active weights for a batch of neurons (gfx950).
//
// Reference: pyglm/regression.py:323-340 (_resample_W): Jp = J_post[ix_(a,a)], hp = h_post[a],
// [W_active; b] = sample_gaussian(J=Jp, h=hp)  ==  L = chol(Jp);  x = L^-T z + Jp^-1 hp   (pybasicbayes, published form).
// Steps here: (1) compact the active sub-block (upper triangle) into Ac; (2) blocked right-looking Cholesky in the
// UPPER form Ac = U'U (U = L'), so every panel U[q0:q0+64, :] is k-major and feeds the fp64 MFMA contraction directly:
// the 64 x 64 diagonal factor and its inverse are formed in LDS, the row-panel solve U12 = U11^-T A12 and the rank-64 /
// rank-256 updates of the trailing matrix run on pgl_gemm.hip; (3) U'w = h, U (mu + x) = w + z;  out = mu + x.
// The forward solve costs nothing of its own: h rides along as column na of the compact block (the system is factored as if it were
// (na + 1)-dimensional, the last "pivot" never taken), so every panel solve and trailing update carries it and the column ends up
// holding w.  The backward solve is one small launch per 64-row panel, from the last: every workgroup solves the panel's diagonal block
// for itself and then subtracts the panel's contribution from its own rows above -- the rows are independent, so a neuron's solve
// spreads over as many workgroups as it has rows / 256, and U streams through once.  (One workgroup per neuron doing all of it took
// 34 ms per batch at cfg3 and over a second per batch at the 32 769-dim systems of configs[4].)
#include "pgl_common.h"

namespace {

constexpr int NBC = 64;

struct CholArgs {
    const double* J; long ldj; long strideJ;   // assembled posterior (lower triangle valid), M = D+2, h in row D+1
    const int* a;                              // [nb][N]
    int* act; long ldact;                      // [nb][ldact] active scalar rows (ascending blocks, bias last)
    int* na;                                   // [nb]
    double* Ac; long ldc; long strideC;        // [nb][ldc][ldc] compact active block -> U in place (upper)
    double* hc;                                // [nb][ldc]  h_active -> w -> mu
    double* Tinv;                              // [nb][64][64] k-major inverse of the current diagonal factor (U11^-1)
    const double* z; long ldz;                 // [nb][ldz]
    double* W;                                 // [nb][N*B] out
    double* b;                                 // [nb] out
    int N, B;
    int* status;
};

__global__ __launch_bounds__(256) void active_index_kernel(CholArgs g) {
    const int n = blockIdx.x;
    if (threadIdx.x != 0) return;
    int cnt = 0;
    int* act = g.act + (long)n * g.ldact;
    for (int m = 0; m < g.N; ++m)
        if (g.a[(long)n * g.N + m])
            for (int b = 0; b < g.B; ++b) act[cnt++] = m * g.B + b;
    act[cnt++] = g.N * g.B;
    g.na[n] = cnt;
}

// Ac[i][j] = J_sym[act[i]][act[j]] for j >= i (upper triangle only: nothing downstream reads below the diagonal of Ac), hc[i] = h[act[i]].
// J stores its lower triangle, so entry (act[i], act[j]), j >= i, lives in ROW act[j]: a 64 x 64 tile is read along i (the columns act[i]
// of a stored row ascend, in runs of B) and turned in LDS so that Ac is written along j.  (One thread per element with the row index in
// blockIdx.y read the stored rows column-wise and launched 18 million workgroups per batch at cfg3: 16 ms; this is 3.)
__global__ __launch_bounds__(256) void gather_active_kernel(CholArgs g) {
    const int n = blockIdx.z;
    const int na = g.na[n];
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    if (i0 >= na || j0 >= na || j0 + 63 < i0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double t[64][65];
    __shared__ int ai[64], aj[64];
    const int* act = g.act + (long)n * g.ldact;
    if (threadIdx.x < 64) ai[threadIdx.x] = i0 + threadIdx.x < na ? act[i0 + threadIdx.x] : -1;
    else if (threadIdx.x < 128) aj[threadIdx.x - 64] = j0 + threadIdx.x - 64 < na ? act[j0 + threadIdx.x - 64] : -1;
    __syncthreads();
    const double* J = g.J + (long)n * g.strideJ;
    const int gi = ai[lane];
    for (int jj = wave; jj < 64; jj += 4) {
        const int gj = aj[jj];
        double v = 0.0;
        if (gi >= 0 && gj >= 0) v = gj >= gi ? J[(long)gj * g.ldj + gi] : J[(long)gi * g.ldj + gj];
        t[jj][lane] = v;
    }
    __syncthreads();
    double* Ac = g.Ac + (long)n * g.strideC;
    const int j = j0 + lane;
    for (int ii = wave; ii < 64; ii += 4) {
        const int i = i0 + ii;
        if (i < na && j < na && j >= i) Ac[(long)i * g.ldc + j] = t[lane][ii];
    }
    // h_active rides along as column na (rows < na); the corner (na, na) is never a pivot
    if (i0 == j0 && threadIdx.x < 64 && gi >= 0) Ac[(long)(i0 + lane) * g.ldc + na] = J[(long)(g.N * g.B + 1) * g.ldj + gi];
    if (i0 == 0 && j0 == 0 && threadIdx.x == 0) Ac[(long)na * g.ldc + na] = 0.0;
}

// factor the 64 x 64 diagonal block at q0: A11 = U11' U11 (U11 written to the upper triangle in place) and Tinv = the inverse of the lower
// factor L = U11', k-major, for the row-panel solve U12 = L^-1 A12 on the MFMA contraction.
// ONE WAVE per neuron, the block in registers: lane i holds row i of the lower triangle (64 doubles).  A step of the right-looking
// factorisation reads what the other rows contribute with v_readlane (the row index is the loop counter of a fully unrolled loop, so
// every register index is static): no LDS, no barrier.  The inverse is the forward substitution L X = I run the same way, row k
// broadcast lane by lane as it becomes final.  (The first version kept the block in LDS with three workgroup barriers per column and
// one thread per column of the inverse: 180 us per launch, 81 launches per batch at cfg3, 513 at configs[4].)
template <int P>
__device__ __forceinline__ double lane_bcast(double v) {      // value of lane P, in every lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), P), hi = __builtin_amdgcn_readlane(__double2hiint(v), P);
    return __hiloint2double(hi, lo);
}
template <int P, int J>
__device__ __forceinline__ void potrf_row_update(double (&a)[NBC], double lip) {
    if constexpr (J < NBC) {
        a[J] -= lip * lane_bcast<J>(a[P]);                    // A[i][J] -= L[i][P] L[J][P]   (entries above the diagonal: unused garbage)
        potrf_row_update<P, J + 1>(a, lip);
    }
}
template <int P>
__device__ __forceinline__ void potrf_steps(double (&a)[NBC], int lane, int& bad) {
    if constexpr (P < NBC) {
        double d = lane_bcast<P>(a[P]);
        if (!(d > 0.0)) { bad = 1; d = 1.0; }
        const double r = sqrt(d), inv = 1.0 / r;
        a[P] = lane == P ? r : a[P] * inv;                    // column P of L (rows below the diagonal scaled; rows above: unused)
        potrf_row_update<P, P + 1>(a, a[P]);
        potrf_steps<P + 1>(a, lane, bad);
    }
}
template <int K, int J>
__device__ __forceinline__ void inv_row_update(double (&x)[NBC], double lik, bool below) {
    if constexpr (J <= K) {
        // (the broadcast is taken by ALL lanes and the update applied by a select: a v_readlane under a divergent branch reads a lane
        // the compiler considers inactive -- undefined in its model, and wrong in practice)
        const double xk = lane_bcast<K>(x[J]);
        x[J] = below ? x[J] - lik * xk : x[J];                // X[i][J] -= L[i][K] X[K][J]   for the rows i > K
        inv_row_update<K, J + 1>(x, lik, below);
    }
}
template <int K>
__device__ __forceinline__ void inv_steps(const double (&a)[NBC], double (&x)[NBC], int lane) {
    if constexpr (K < NBC) {
        // row K of X is final once scaled by 1 / L[K][K]; rows below it take its contribution
        const double rk = 1.0 / lane_bcast<K>(a[K]);
#pragma unroll
        for (int j = 0; j <= K; ++j) x[j] = lane == K ? x[j] * rk : x[j];
        inv_row_update<K, 0>(x, a[K], lane > K);
        inv_steps<K + 1>(a, x, lane);
    }
}
// The factorisation and the inversion as ONE chain of steps: step P of the inverse needs column P of L, which is final after step P of the
// factorisation, and touches other registers than step P + 1 of the factorisation does -- two dependent chains of cross-lane broadcasts that the
// scheduler can now interleave (each alone leaves the SIMD waiting on a v_readlane most of the time).  Same operations on the same values
// as potrf_steps followed by inv_steps: the same bits.
template <int P>
__device__ __forceinline__ void potrf_inv_steps(double (&a)[NBC], double (&x)[NBC], int lane, int& bad) {
    if constexpr (P < NBC) {
        double d = lane_bcast<P>(a[P]);
        if (!(d > 0.0)) { bad = 1; d = 1.0; }
        const double r = sqrt(d), inv = 1.0 / r;
        a[P] = lane == P ? r : a[P] * inv;
        potrf_row_update<P, P + 1>(a, a[P]);
#pragma unroll
        for (int j = 0; j <= P; ++j) x[j] = lane == P ? x[j] * inv : x[j];
        inv_row_update<P, 0>(x, a[P], lane > P);
        potrf_inv_steps<P + 1>(a, x, lane, bad);
    }
}
template <int K>
__device__ __forceinline__ void hcol_steps(const double (&a)[NBC], double& v, int lane) {
    if constexpr (K < NBC) {
        const double wk = lane_bcast<K>(v) / lane_bcast<K>(a[K]);
        if (lane == K) v = wk; else if (lane > K) v -= a[K] * wk;
        hcol_steps<K + 1>(a, v, lane);
    }
}

template <bool MERGED>
__global__ __launch_bounds__(64) void potrf_diag_kernel(CholArgs g, int q0) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int na = g.na[n];
    if (q0 >= na) return;
    const int nb = min(NBC, na - q0);
    double* Ag = g.Ac + (long)n * g.strideC + (long)q0 * g.ldc + q0;
    // row `lane` of the lower triangle = column `lane` of the stored upper one; the identity beyond a short last block
    double a[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) a[j] = (lane < nb && j <= lane) ? Ag[(long)j * g.ldc + lane] : (j == lane ? 1.0 : 0.0);
    int bad = 0;
    // X = L^-1 (lower): row `lane` in registers, starting from the identity; built step by step beside the factor
    double x[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) x[j] = j == lane ? 1.0 : 0.0;
    if constexpr (MERGED) potrf_inv_steps<0>(a, x, lane, bad);
    else { potrf_steps<0>(a, lane, bad); inv_steps<0>(a, x, lane); }
#pragma unroll
    for (int j = 0; j < NBC; ++j)
        if (lane < nb && j <= lane) Ag[(long)j * g.ldc + lane] = a[j];          // U11[j][lane] = L[lane][j]
    if (lane == 0 && bad) atomicOr(&g.status[n], 4);
    // the neuron's last block, not a full one: column na (the h column) lies inside it, where no panel solve reaches -- L w = h here
    if (q0 + nb == na && nb < NBC) {
        double* hcol = g.Ac + (long)n * g.strideC + (long)q0 * g.ldc + na;
        double v = lane < nb ? hcol[(long)lane * g.ldc] : 0.0;
        hcol_steps<0>(a, v, lane);
        if (lane < nb) hcol[(long)lane * g.ldc] = v;
    }
    double* Tn = g.Tinv + (long)n * NBC * NBC;
#pragma unroll
    for (int k = 0; k < NBC; ++k) Tn[k * NBC + lane] = (lane >= k && lane < nb && k < nb) ? x[k] : 0.0;     // Tinv[k][m] = (L^-1)[m][k]
}

// r = w + z: w from column na of the factored block, z the standard normals (U (mu + x) = w + z gives the draw in one solve)
__global__ __launch_bounds__(256) void backsolve_init_kernel(CholArgs g) {
    const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int na = g.na[n];
    if (i >= na) return;
    g.hc[(long)n * g.ldc + i] = g.Ac[(long)n * g.strideC + (long)i * g.ldc + na] + g.z[(long)n * g.ldz + i];
}

// panel pi (rows p0 = 64 pi .. p0 + pw) of U v = r, from the last panel: v_p = U_pp^-1 r_p by every workgroup for itself (one wave, in
// LDS), then r[row] -= U[row][p0 .. p0 + pw) . v_p for the workgroup's rows above the panel.  Solved values go to the second plane of hc
// (nobody writes the panel's own rows of r in this launch, so the redundant solves all read the same numbers).
constexpr int BS_ROWS = 256;
__global__ __launch_bounds__(256) void backsolve_panel_kernel(CholArgs g, int pi, long plane) {
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = g.na[n];
    const int p0 = pi * NBC;
    if (p0 >= na) return;
    const int pw = min(NBC, na - p0);
    const int r0 = blockIdx.x * BS_ROWS;
    if (r0 >= p0 && blockIdx.x > 0) return;                 // no rows of this workgroup above the panel (workgroup 0 still publishes v_p)
    const double* U = g.Ac + (long)n * g.strideC;
    const long ld = g.ldc;
    double* r = g.hc + (long)n * g.ldc;
    double* v = r + plane;
    __shared__ double Ud[NBC][NBC + 1];
    __shared__ double vp[NBC];
    for (int e = tid; e < NBC * NBC; e += 256) {
        const int i = e >> 6, c = e & 63;
        Ud[i][c] = (i < pw && c < pw && c >= i) ? U[(long)(p0 + i) * ld + p0 + c] : (i == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (wave == 0) {
        double x = lane < pw ? r[p0 + lane] : 0.0;
        for (int i = pw - 1; i >= 0; --i) {
            const double xi = __shfl(x, i) / Ud[i][i];
            if (lane == i) x = xi; else if (lane < i) x -= Ud[lane][i] * xi;
        }
        vp[lane] = lane < pw ? x : 0.0;
        if (blockIdx.x == 0 && lane < pw) v[p0 + lane] = x;
    }
    __syncthreads();
    // rows above the panel: 16 lanes per row (4 consecutive doubles each), 4 rows per wave-instruction, 16 rows per wave and trip
    const int rl = lane >> 4, cl = (lane & 15) * 4;
    const double v0 = vp[cl], v1 = vp[cl + 1], v2 = vp[cl + 2], v3 = vp[cl + 3];
    const int rend = min(p0, r0 + BS_ROWS);
    for (int rb = r0 + wave * 16; rb < rend; rb += 64) {
        double acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rb + q * 4 + rl;
            acc[q] = 0.0;
            if (row < rend) {
                const double* up = U + (long)row * ld + p0 + cl;
                if (cl + 3 < pw) { const d2_t a = *reinterpret_cast<const d2_t*>(up), b = *reinterpret_cast<const d2_t*>(up + 2); acc[q] = (a[0] * v0 + a[1] * v1) + (b[0] * v2 + b[1] * v3); }
                else { for (int c = 0; c < 4; ++c) if (cl + c < pw) acc[q] += up[c] * vp[cl + c]; }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double a = acc[q];
            a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
            const int row = rb + q * 4 + rl;
            if ((lane & 15) == 0 && row < rend) r[row] -= a;
        }
    }
}

// scatter (after W has been zeroed): mu + x -- one vector, the solve of w + z -- for the active blocks, the bias last
__global__ __launch_bounds__(256) void scatter_active_kernel(CholArgs g, long plane) {
    const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int na = g.na[n];
    if (i >= na) return;
    const double val = g.hc[(long)n * g.ldc + plane + i];
    if (i == na - 1) g.b[n] = val; else g.W[(long)n * g.N * g.B + g.act[(long)n * g.ldact + i]] = val;
}

}  // namespace


static CholArgs mk(const PglCholState& s) {
    return CholArgs{s.J, s.ldj, s.strideJ, s.a, s.act, s.ldact, s.na, s.Ac, s.ldc, s.strideC, s.hc, s.Tinv, s.z, s.ldz, s.W, s.b, s.N, s.B, s.status};
}

int pgl_k_chol_index(const PglCholState& s, hipStream_t st) {
    hipLaunchKernelGGL(active_index_kernel, dim3(s.nb), dim3(64), 0, st, mk(s));
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

// na_max: host-side upper bound of the active sizes (every neuron stops at its own na[n])
int pgl_k_chol_sample(const PglCholState& s, int na_max, hipStream_t st) {
    CholArgs g = mk(s);
    if (na_max <= 0) return PGL_OK;
    hipLaunchKernelGGL(gather_active_kernel, dim3((na_max + 63) / 64, (na_max + 63) / 64, s.nb), dim3(256), 0, st, g);
    PGL_CHECK_LAUNCH();
    // the products see na + 1 columns (the h column rides along): extents from nd = na_max + 1, per-neuron sizes as na[n] - (c0 - 1)
    const int nd = na_max + 1;
    auto trailing = [&](int krow0, int K, int c0, int mfix) -> int {
        // C[c0.., c0..] -= P' P with P = rows [krow0, krow0+K) of Ac, columns from c0 (k-major panel)
        PglGemmArgs t{};
        const double* P = s.Ac + (long)krow0 * s.ldc + c0;
        const int rem = nd - c0;
        t.A = P; t.lda = s.ldc; t.strideA = s.strideC;
        t.B = P; t.ldb = s.ldc; t.strideB = s.strideC;
        t.C = s.Ac + (long)c0 * s.ldc + c0; t.ldc = s.ldc; t.strideC = s.strideC;
        t.N = rem; t.K = K; t.a_cols = rem + (rem & 1); t.b_cols = t.a_cols; t.nbatch = s.nb; t.nz_total = 0;
        t.alpha = -1.0; t.beta = 1.0; t.batch_k = nullptr; t.batch_dim = s.na; t.dim_off = c0 - 1; t.W = nullptr; t.ldw = 0;
        // strip: mfix rows only -- and never rows past a neuron's own remainder (they would land in the padding or, beyond ldc, in the next
        // neuron's block)
        if (mfix > 0) { t.M = mfix; t.tri = 0; t.dim_mode = 3; return pgl_launch_gemm(PGL_GEMM_PLAIN, t, st); }
        t.M = rem; t.tri = 2; t.dim_mode = 0; t.pipe = 1;
        return pgl_launch_gemm(PGL_GEMM_TRI1, t, st);
    };
    auto panel_solve = [&](int q0) -> int {
        // rows [q0, q0+64), columns from q0+64:  A12 <- L^-1 A12  (in place: a tile reads its 64 x 256 block completely before storing it)
        const int c0 = q0 + NBC, rem = nd - c0;
        PglGemmArgs t{};
        t.A = s.Tinv; t.lda = NBC; t.strideA = (long)NBC * NBC; t.a_cols = NBC;
        t.B = s.Ac + (long)q0 * s.ldc + c0; t.ldb = s.ldc; t.strideB = s.strideC; t.b_cols = rem + (rem & 1);
        t.C = s.Ac + (long)q0 * s.ldc + c0; t.ldc = s.ldc; t.strideC = s.strideC;
        t.M = NBC; t.N = rem; t.K = NBC; t.nbatch = s.nb; t.alpha = 1.0; t.beta = 0.0; t.tri = 0;
        t.batch_dim = s.na; t.dim_off = c0 - 1; t.dim_mode = 1;
        return pgl_launch_gemm(PGL_GEMM_PLAIN, t, st);
    };
    // super-panels of SP 64-row sub-panels: before sub-panel i is factored its 64-row strip takes the updates of sub-panels 0..i-1 (one
    // rank-64i strip update), and the trailing matrix is updated ONCE per super-panel with rank 64 SP.  The trailing passes stream the
    // whole remaining matrix (HBM-bound at rank 128): SP = 4 halves them again, SP = 6 is the measured optimum.
    // (full 5121-dim systems x 256, ms: 2: 308, 4: 282, 6: 274, 8: 274; 32 769-dim systems x 4, ms per 8: 6: 1887, 8: 1862)
    static const int SP_env = [] { const int v = pgl_ab_int("PGL_CHOL_SP", 0); return v >= 1 && v <= 8 ? v : 0; }();
    const int SP = SP_env ? SP_env : (s.ldact > 8192 ? 8 : 6);         // (from the model's size, not from the hint na_max: a hint must not move a bit of the result)
    static const bool merged = pgl_ab_int("PGL_CHOL_MERGED", 1) != 0;      // factor and inverse of a diagonal block as one interleaved chain (-2.4 % of the stage)
    bool done = false;
    for (int q0 = 0; q0 < na_max && !done; q0 += SP * NBC) {
        for (int i = 0; i < SP; ++i) {
            const int qi = q0 + i * NBC;
            if (qi >= na_max) { done = true; break; }
            if (i > 0) {
                const int rc = trailing(q0, i * NBC, qi, NBC);          // strip: rows of sub-panel i, columns from its diagonal block
                if (rc) return rc;
            }
            if (merged) hipLaunchKernelGGL(potrf_diag_kernel<true>, dim3(s.nb), dim3(64), 0, st, g, qi);
            else hipLaunchKernelGGL(potrf_diag_kernel<false>, dim3(s.nb), dim3(64), 0, st, g, qi);
            PGL_CHECK_LAUNCH();
            if (nd - qi - NBC <= 0) { done = true; break; }
            const int rc = panel_solve(qi);
            if (rc) return rc;
        }
        if (done || nd - q0 - SP * NBC <= 0) break;
        const int rc = trailing(q0, SP * NBC, q0 + SP * NBC, 0);        // rank-256 update of everything right of / below the super-panel
        if (rc) return rc;
    }
    // U (mu + x) = w + z, panel by panel from the last; then the scatter
    const long plane = (long)s.nb * s.ldc;
    // (the whole backward solve as ONE launch -- a workgroup per 256 rows, panels handed on through flags in device memory -- was built and
    // measured in round 5: same bits, 118.2 against 112.7 ms per batch for the weight stage at 3 670 active rows; the waiting workgroups hold
    // the CUs the chain's head needs.  profiles/r05_chol_merged_chain_ab.txt)
    hipLaunchKernelGGL(backsolve_init_kernel, dim3((na_max + 255) / 256, s.nb), dim3(256), 0, st, g);
    PGL_CHECK_LAUNCH();
    for (int pi = (na_max + NBC - 1) / NBC - 1; pi >= 0; --pi) {
        const int rows_above = pi * NBC;
        const int nblk = rows_above > 0 ? (rows_above + BS_ROWS - 1) / BS_ROWS : 1;
        hipLaunchKernelGGL(backsolve_panel_kernel, dim3(nblk, s.nb), dim3(256), 0, st, g, pi, plane);
        PGL_CHECK_LAUNCH();
    }
    if (hipMemsetAsync(s.W, 0, (size_t)s.nb * s.N * s.B * sizeof(double), st) != hipSuccess) { pgl_set_error("memset failed"); return PGL_ERR_HIP; }
    hipLaunchKernelGGL(scatter_active_kernel, dim3((na_max + 255) / 256, s.nb), dim3(256), 0, st, g, plane);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}
