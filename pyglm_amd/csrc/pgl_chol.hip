// Gaussian conditional of the active weights for a batch of neurons (gfx950).
//
// Reference: pyglm/regression.py:323-340 (_resample_W): Jp = J_post[ix_(a,a)], hp = h_post[a],
// [W_active; b] = sample_gaussian(J=Jp, h=hp)  ==  L = chol(Jp);  x = L^-T z + Jp^-1 hp   (pybasicbayes, published form).
// Steps here: (1) compact the active sub-block (upper triangle) into Ac; (2) blocked right-looking Cholesky in the
// UPPER form Ac = U'U (U = L'), so every panel U[q0:q0+64, :] is k-major and feeds the fp64 MFMA contraction directly:
// the 64 x 64 diagonal factor and its inverse are formed in LDS, the row-panel solve U12 = U11^-T A12 and the rank-64 /
// rank-128 updates of the trailing matrix run on pgl_gemm.hip; (3) U'w = h, U mu = w, U x = z;  out = mu + x.
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
    if (i0 == j0 && threadIdx.x < 64 && gi >= 0) g.hc[(long)n * g.ldc + i0 + lane] = J[(long)(g.N * g.B + 1) * g.ldj + gi];
}

// factor the 64x64 diagonal block at q0: A11 = U11' U11, U11 written to the upper triangle in place
__global__ __launch_bounds__(256) void potrf_diag_kernel(CholArgs g, int q0) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const int na = g.na[n];
    if (q0 >= na) return;
    const int nb = min(NBC, na - q0);
    __shared__ double A[NBC][NBC + 1];
    __shared__ int s_bad;
    double* Ag = g.Ac + (long)n * g.strideC + (long)q0 * g.ldc + q0;
    for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, j = e % nb; A[i][j] = Ag[(long)(i <= j ? i : j) * g.ldc + (i <= j ? j : i)]; }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    // right-looking lower Cholesky on the symmetric block (L = U11'), column by column
    for (int p = 0; p < nb; ++p) {
        if (tid == 0) { double d = A[p][p]; if (!(d > 0.0)) { s_bad = 1; d = 1.0; } A[p][p] = sqrt(d); }
        __syncthreads();
        const double dinv = 1.0 / A[p][p];
        for (int i = p + 1 + tid; i < nb; i += 256) A[i][p] *= dinv;
        __syncthreads();
        const int rem = nb - p - 1;
        for (int e = tid; e < rem * rem; e += 256) {
            const int i = p + 1 + e / rem, j = p + 1 + e % rem;
            if (j <= i) A[i][j] -= A[i][p] * A[j][p];
        }
        __syncthreads();
    }
    for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, j = e % nb; if (i <= j) Ag[(long)i * g.ldc + j] = A[j][i]; }
    if (tid == 0 && s_bad) atomicOr(&g.status[n], 4);
    // inverse of the lower factor, one column per thread (forward substitution on e_j); the row-panel solve
    // U12 = U11^-T A12 = L^-1 A12 then runs on the MFMA contraction with Tinv[k][m] = (L^-1)[m][k] as its k-major operand
    __shared__ double Li[NBC][NBC + 1];
    if (tid < NBC) {
        const int j = tid;
        for (int i = 0; i < NBC; ++i) {
            double sacc = (i == j) ? 1.0 : 0.0;
            if (i < nb && j < nb) {
                for (int k = j; k < i; ++k) sacc -= A[i][k] * Li[k][j];
                Li[i][j] = (i >= j) ? sacc / A[i][i] : 0.0;
            } else {
                Li[i][j] = 0.0;
            }
        }
    }
    __syncthreads();
    double* Tn = g.Tinv + (long)n * NBC * NBC;
    for (int e = tid; e < NBC * NBC; e += 256) { const int k = e / NBC, m = e % NBC; Tn[e] = Li[m][k]; }
}

// one workgroup per neuron:  U'w = h, then U mu = w and U x = z; scatter mu + x.  Blocked by 64 rows: the 64 x 64 diagonal block is solved
// by one wave out of LDS (the unknowns live one per lane, a step is a shuffle and one multiply-add), the rest of the 64-row panel is
// streamed once -- forward as a column-parallel update of the remaining right-hand side, backward as row dot products over 64 x 64
// tiles staged through LDS, one tile per wave -- with two workgroup barriers per panel instead of two per row.
__global__ __launch_bounds__(256) void solve_sample_kernel(CholArgs g) {
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = g.na[n];
    const double* U = g.Ac + (long)n * g.strideC;
    const long ld = g.ldc;
    double* h = g.hc + (long)n * g.ldc;            // becomes w, then mu
    double* xz = h + (long)gridDim.x * g.ldc;      // second plane of hc: z -> x
    const double* z = g.z + (long)n * g.ldz;
    __shared__ double Ud[NBC][NBC + 1];            // diagonal block
    __shared__ double wp[2][NBC];                  // the panel's solved unknowns (two right-hand sides backward)
    __shared__ double tile[4][NBC / 2][NBC + 1];   // backward: 32 x 64 half tiles of the panel, one per wave at a time
    __shared__ double part[4][2][NBC];
    for (int i = tid; i < na; i += 256) xz[i] = z[i];
    __syncthreads();
    // ---- forward  U'w = h: panel p0: solve the diagonal block (w_j = h_j / U_jj; h_c -= U_jc w_j), then h[c] -= sum_j U[p0+j][c] w_j for c beyond
    for (int p0 = 0; p0 < na; p0 += NBC) {
        const int pw = min(NBC, na - p0);
        for (int e = tid; e < NBC * NBC; e += 256) {
            const int r = e >> 6, c = e & 63;
            Ud[r][c] = (r < pw && c < pw && c >= r) ? U[(long)(p0 + r) * ld + p0 + c] : (r == c ? 1.0 : 0.0);
        }
        __syncthreads();
        if (wave == 0) {
            double v = lane < pw ? h[p0 + lane] : 0.0;
            for (int j = 0; j < pw; ++j) {
                const double wj = __shfl(v, j) / Ud[j][j];
                if (lane == j) v = wj; else if (lane > j) v -= Ud[j][lane] * wj;
            }
            wp[0][lane] = v;
            if (lane < pw) h[p0 + lane] = v;
        }
        __syncthreads();
        // 32 independent row loads per column in flight (the kernel has four waves per CU: latency is hidden by depth, not by occupancy);
        // rows past the end of a short last panel are clamped and meet wp = 0
        for (int c = p0 + pw + tid; c < na; c += 256) {
            double acc = 0.0;
#pragma unroll
            for (int jb = 0; jb < NBC; jb += 32) {
                double u[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) u[j] = U[(long)min(p0 + jb + j, na - 1) * ld + c];
#pragma unroll
                for (int j = 0; j < 32; ++j) acc += u[j] * wp[0][jb + j];
            }
            h[c] -= acc;
        }
        __syncthreads();
    }
    // ---- backward  U mu = w, U x = z (two right-hand sides): panels from the last; s_r = sum_{c beyond the panel} U[r][c] v[c] by tiles
    const int np = (na + NBC - 1) / NBC;
    for (int pi = np - 1; pi >= 0; --pi) {
        const int p0 = pi * NBC, pw = min(NBC, na - p0);
        double sa1 = 0.0, sa2 = 0.0, sb1 = 0.0, sb2 = 0.0;   // lane r < 32 of every wave: partial sums of rows p0 + r and p0 + 32 + r
        for (int c0 = p0 + NBC + wave * NBC; c0 < na; c0 += 4 * NBC) {
            const int cw = min(NBC, na - c0);
            const double v1 = lane < cw ? h[c0 + lane] : 0.0, v2 = lane < cw ? xz[c0 + lane] : 0.0;
#pragma unroll
            for (int rh = 0; rh < NBC; rh += NBC / 2) {
                for (int r = 0; r < NBC / 2; ++r)
                    tile[wave][r][lane] = (rh + r < pw && lane < cw) ? U[(long)(p0 + rh + r) * ld + c0 + lane] : 0.0;   // coalesced rows
                __builtin_amdgcn_wave_barrier();        // one wave = one instruction stream: its LDS writes are visible to its reads
                double t1 = 0.0, t2 = 0.0;
#pragma unroll 8
                for (int c = 0; c < NBC; ++c) {
                    const double u = tile[wave][lane & 31][c];
                    t1 += u * __shfl(v1, c);
                    t2 += u * __shfl(v2, c);
                }
                if (rh == 0) { sa1 += t1; sa2 += t2; } else { sb1 += t1; sb2 += t2; }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane < 32) {
            part[wave][0][lane] = sa1; part[wave][1][lane] = sa2;
            part[wave][0][32 + lane] = sb1; part[wave][1][32 + lane] = sb2;
        }
        for (int e = tid; e < NBC * NBC; e += 256) {
            const int r = e >> 6, c = e & 63;
            Ud[r][c] = (r < pw && c < pw && c >= r) ? U[(long)(p0 + r) * ld + p0 + c] : (r == c ? 1.0 : 0.0);
        }
        __syncthreads();
        if (wave < 2) {                            // wave 0: mu, wave 1: x
            double* vec = wave == 0 ? h : xz;
            double v = lane < pw ? vec[p0 + lane] - ((part[0][wave][lane] + part[1][wave][lane]) + (part[2][wave][lane] + part[3][wave][lane])) : 0.0;
            for (int i = pw - 1; i >= 0; --i) {
                const double xi = __shfl(v, i) / Ud[i][i];
                if (lane == i) v = xi; else if (lane < i) v -= Ud[lane][i] * xi;
            }
            if (lane < pw) vec[p0 + lane] = v;
        }
        __syncthreads();
    }
    // scatter: zeros for inactive blocks
    const int D = g.N * g.B;
    double* W = g.W + (long)n * D;
    for (int i = tid; i < D; i += 256) W[i] = 0.0;
    __syncthreads();
    const int* act = g.act + (long)n * g.ldact;
    for (int i = tid; i < na; i += 256) {
        const double v = h[i] + xz[i];
        if (i == na - 1) g.b[n] = v; else W[act[i]] = v;
    }
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

// na_max: host-side upper bound of the active sizes (read back from s.na by the caller)
int pgl_k_chol_sample(const PglCholState& s, int na_max, hipStream_t st) {
    CholArgs g = mk(s);
    if (na_max <= 0) return PGL_OK;
    hipLaunchKernelGGL(gather_active_kernel, dim3((na_max + 63) / 64, (na_max + 63) / 64, s.nb), dim3(256), 0, st, g);
    PGL_CHECK_LAUNCH();
    auto trailing = [&](int krow0, int K, int c0, int mfix) -> int {
        // C[c0.., c0..] -= P' P with P = rows [krow0, krow0+K) of Ac, columns from c0 (k-major panel)
        PglGemmArgs t{};
        const double* P = s.Ac + (long)krow0 * s.ldc + c0;
        const int rem = na_max - c0;
        t.A = P; t.lda = s.ldc; t.strideA = s.strideC;
        t.B = P; t.ldb = s.ldc; t.strideB = s.strideC;
        t.C = s.Ac + (long)c0 * s.ldc + c0; t.ldc = s.ldc; t.strideC = s.strideC;
        t.N = rem; t.K = K; t.a_cols = rem + (rem & 1); t.b_cols = t.a_cols; t.nbatch = s.nb; t.nz_total = 0;
        t.alpha = -1.0; t.beta = 1.0; t.batch_k = nullptr; t.batch_dim = s.na; t.dim_off = c0; t.W = nullptr; t.ldw = 0;
        // strip: mfix rows only -- and never rows past a neuron's own remainder (they would land in the padding or, beyond ldc, in the next
        // neuron's block)
        if (mfix > 0) { t.M = mfix; t.tri = 0; t.dim_mode = 3; return pgl_launch_gemm(PGL_GEMM_PLAIN, t, st); }
        t.M = rem; t.tri = 2; t.dim_mode = 0; t.pipe = 1;
        return pgl_launch_gemm(PGL_GEMM_TRI1, t, st);
    };
    auto panel_solve = [&](int q0) -> int {
        // rows [q0, q0+64), columns from q0+64:  A12 <- L^-1 A12  (in place: a tile reads its 64 x 256 block completely before storing it)
        const int c0 = q0 + NBC, rem = na_max - c0;
        PglGemmArgs t{};
        t.A = s.Tinv; t.lda = NBC; t.strideA = (long)NBC * NBC; t.a_cols = NBC;
        t.B = s.Ac + (long)q0 * s.ldc + c0; t.ldb = s.ldc; t.strideB = s.strideC; t.b_cols = rem + (rem & 1);
        t.C = s.Ac + (long)q0 * s.ldc + c0; t.ldc = s.ldc; t.strideC = s.strideC;
        t.M = NBC; t.N = rem; t.K = NBC; t.nbatch = s.nb; t.alpha = 1.0; t.beta = 0.0; t.tri = 0;
        t.batch_dim = s.na; t.dim_off = c0; t.dim_mode = 1;
        return pgl_launch_gemm(PGL_GEMM_PLAIN, t, st);
    };
    // super-panels of SP 64-row sub-panels: before sub-panel i is factored its 64-row strip takes the updates of sub-panels 0..i-1 (one
    // rank-64i strip update), and the trailing matrix is updated ONCE per super-panel with rank 64 SP.  The trailing passes stream the
    // whole remaining matrix (HBM-bound at rank 128): SP = 4 halves them again.
    constexpr int SP = 4;
    bool done = false;
    for (int q0 = 0; q0 < na_max && !done; q0 += SP * NBC) {
        for (int i = 0; i < SP; ++i) {
            const int qi = q0 + i * NBC;
            if (qi >= na_max) { done = true; break; }
            if (i > 0) {
                const int rc = trailing(q0, i * NBC, qi, NBC);          // strip: rows of sub-panel i, columns from its diagonal block
                if (rc) return rc;
            }
            hipLaunchKernelGGL(potrf_diag_kernel, dim3(s.nb), dim3(256), 0, st, g, qi);
            PGL_CHECK_LAUNCH();
            if (na_max - qi - NBC <= 0) { done = true; break; }
            const int rc = panel_solve(qi);
            if (rc) return rc;
        }
        if (done || na_max - q0 - SP * NBC <= 0) break;
        const int rc = trailing(q0, SP * NBC, q0 + SP * NBC, 0);        // rank-256 update of everything right of / below the super-panel
        if (rc) return rc;
    }
    hipLaunchKernelGGL(solve_sample_kernel, dim3(s.nb), dim3(256), 0, st, g);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}
