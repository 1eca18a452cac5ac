// Collapsed spike-and-slab resampling of the adjacency indicators a[m] for a batch of neurons (gfx950).
//
// Reference: pyglm/regression.py:282-320 (_collapsed_resample_a) + :343-378 (_marginal_likelihood): for every
// presynaptic m in a random order the reference extracts the active sub-block of (J_prior, J_post), takes two
// dense Choleskys and two dpotrs -- O((sum(a) B)^3) per proposal, N+1 times per neuron.
//
// Here the same log-odds come from a symmetric SWEEP TABLEAU of the posterior system, kept per neuron:
//     A = [[J_post, h_post], [h_post', 0]]   (size D+2, lower triangle stored),   M = sweep(A, S)
// where S = {bias} U {active blocks}.  With P = J_SS^-1:
//     inactive m:  M_mm = Schur complement  Sigma_m = J_mm - J_mS P J_Sm,   M_mh = r_m = h_m - J_mS P h_S
//     active   m:  M_mm = -P_mm,                                            M_mh = mu_m = (P h_S)_m
// so the change in log marginal likelihood for switching block m on is
//     -1/2 log|Sigma_m| + 1/2 r' Sigma^-1 r + c0[m]      (inactive)      [c0 = prior term, 1/2 log|J_w| - 1/2 mu_w' J_w mu_w]
//     +1/2 log|P_mm|    + 1/2 mu' P_mm^-1 mu + c0[m]     (active)
// -- O(B^3) per proposal.  Because J_prior is block diagonal (regression.py:218) its Cholesky terms reduce to c0.
// A proposal window of R blocks only ever touches the (R B + 1)^2 sub-tableau on those blocks and h, so a window is
// gathered into LDS, its R proposals run there (accepted flips = local block sweeps), and the NET set Delta of changed
// blocks is applied to the full tableau as one symmetric rank-|Delta|B update:
//     G = (M_DD)^-1;   M_RR -= M_RD G M_DR;   M_RD = M_RD G Sg;   M_DD = -Sg G Sg      (Sg = diag(+1 forward, -1 reverse))
// (the non-pivot formula is the same for forward and reverse sweeps; derivation in DESIGN.md).  The rank-k update
// runs on the fp64 MFMA kernel (pgl_gemm.hip, lower-triangular tiles only).
#include "pgl_common.h"

namespace {

constexpr int KMAX = 512;   // max pivots (scalar rows) per tableau update: a chunk of the initial sweep (rank-512 passes over the tableau)
constexpr int KWIN = 320;   // max scalar rows of the blocks of one proposal window (its sub-tableau lives in an L2-resident global scratch).
                            // A flip costs the window's sub-tableau squared, a window one pass over the trailing tableau; measured at cfg3 (256
                            // neurons, proposals + window updates, ms per batch; PGL_FLIP_WINDOW): with ~13 % of the blocks flipping 240: 110,
                            // 320: 98, 400: 93, 480: 96; with 30 %: 320: 149, 400: 154; with 50 %: 240: 190, 320: 205 -- the chain flips 7-35 %

struct FlipArgs {
    double* M; long ldj; long strideM;           // tableau per neuron
    int N, B, R;                                 // R = blocks per window
    const int* perm;                             // [nb][N]
    const double* u;                             // [nb][N]
    const double* rho;                           // [nb][N]
    const double* c0;                            // [nb][N]
    int* a;                                      // [nb][N] in/out
    const int* skip;                             // [nb] or null
    int* d_idx;                                  // [nb][KMAX] scalar row index of each pivot
    double* d_sign;                              // [nb][KMAX]
    int* d_cnt;                                  // [nb]
    int* batch_k;                                // [nb]  padded K for the MFMA update (0 = nothing to do)
    double* G;                                   // [nb][KMAX][KMAX]
    double* Lws;                                 // [nb][(KMAX+1)^2] window sub-tableau scratch
    double* Ut; double* Wt; long ldu;            // [nb][KMAX][ldu]
    int* status;                                 // [nb] sticky error flags (1 = non-PD block met)
    int permuted;                                // 1: tableau rows/columns are in VISIT order (position k holds block perm[k]; bias, h last)
    int c_begin;                                 // first column the pivot-row gather has to produce (trailing-only window updates)
    double* logodds;                             // [nb][N] or null: lps[1] - lps[0] per proposal step (parity checks)
    const int* row_off;                          // [nb] or null: the gathered pivot rows go to Ut rows row_off[n].. (second panel of a window pair)
};

__device__ __forceinline__ double tab_get(const double* M, long ld, int i, int j) { return i >= j ? M[(long)i * ld + j] : M[(long)j * ld + i]; }

// ------------------------------------------------------------------ proposals of one window, in LDS
// BT = B at compile time for the common block sizes (0 = any B): the loops over a block unroll and the rank-B update of a flip runs on
// register tiles.
template <int BT>
__global__ __launch_bounds__(1024) void decide_kernel(FlipArgs g, int window) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int N = g.N, B = BT ? BT : g.B, D = N * B;
    const int k0 = window * g.R;
    const int nblk = min(g.R, N - k0);
    if (nblk <= 0 || (g.skip && g.skip[n])) { if (tid == 0) { g.d_cnt[n] = 0; g.batch_k[n] = 0; } return; }
    const int nl = nblk * B + 1, ldl = nl;
    double* L = g.Lws + (size_t)n * (KMAX + 1) * (KMAX + 1);   // [nl][ldl], global (stays in this CU's L1 / the XCD's L2)
    double* Tm = lds;                        // [nl][B]
    double* Prow = Tm + (size_t)nl * B;      // [B][nl]  pivot rows of the block being swept
    double* Cinv = Prow + (size_t)nl * B;    // [B][B]
    double* Cbs = Cinv + B * B;              // [R][B][B] Cholesky factor of every proposal's block
    double* vbs = Cbs + (size_t)g.R * B * B; // [R][B]
    __shared__ int s_flip, s_sign, s_first, s_nfl;
    __shared__ int s_fl[KMAX];              // rows of the blocks that have flipped so far (ascending)
    if (tid == 0) s_nfl = 0;
    __shared__ int s_flipped[KMAX];

    const double* M = g.M + (long)n * g.strideM;
    const int* perm = g.perm + (long)n * N;
    // the window's sub-tableau is symmetric: only its lower triangle (i >= j) is kept and updated -- half the read-modify-writes per flip
    for (int e = tid; e < nl * nl; e += nthr) {
        const int i = e / nl, j = e % nl;
        if (j > i) continue;
        const int gi = (i < nl - 1) ? (g.permuted ? k0 * B + i : perm[k0 + i / B] * B + i % B) : D + 1;
        const int gj = (j < nl - 1) ? (g.permuted ? k0 * B + j : perm[k0 + j / B] * B + j % B) : D + 1;
        L[i * ldl + j] = tab_get(M, g.ldj, gi, gj);
    }
    __syncthreads();
    auto Ls = [&](int i, int j) { return i >= j ? L[i * ldl + j] : L[j * ldl + i]; };

    const int hcol = nl - 1;
    // The proposals are sequential -- a flip changes the sub-tableau every later proposal reads -- but most do not flip (7-35 % do along
    // the chain).  So all proposals not yet decided are EVALUATED at once, one per thread, from the current sub-tableau; the decisions up to
    // and including the first flip are committed (they saw exactly the tableau the serial order would have shown them), that flip is
    // applied by the whole workgroup, and the rest is evaluated again.  A window costs (flips + 1) evaluation rounds instead of one
    // serial evaluation per proposal (64 of them, a dozen dependent L2 reads each); the decisions are the same function of the same numbers.
    int k = 0;
    while (k < nblk) {
        if (tid == 0) s_first = nblk;
        __syncthreads();
        const int kk = k + tid;
        int my_v = 0, my_am = 0, my_ok = 1;
        double my_lo = __builtin_nan("");
        if (kk < nblk) {
            const int m = perm[k0 + kk], p0 = kk * B;
            double* Cb = Cbs + (size_t)kk * B * B;
            double* vb = vbs + (size_t)kk * B;
            const int am = g.a[(long)n * N + m];
            const double sgn = am ? -1.0 : 1.0;
            bool ok = true;
            double logdet = 0.0;
            for (int i = 0; i < B; ++i) {        // Cholesky of Q = sgn * L[p,p] (lower, in Cb)
                for (int j = 0; j <= i; ++j) {
                    double s = sgn * L[(p0 + i) * ldl + p0 + j];
                    for (int x = 0; x < j; ++x) s -= Cb[i * B + x] * Cb[j * B + x];
                    if (i == j) { if (!(s > 0.0)) { ok = false; s = 1.0; } Cb[i * B + i] = sqrt(s); logdet += log(s); }
                    else Cb[i * B + j] = s / Cb[j * B + j];
                }
            }
            double quad = 0.0;
            for (int i = 0; i < B; ++i) {        // y = Lc^-1 v ;  quad = y'y = v' Q^-1 v
                double s = L[hcol * ldl + p0 + i];          // (h is the last row: entry (p0 + i, h) of the symmetric sub-tableau)
                for (int x = 0; x < i; ++x) s -= Cb[i * B + x] * vb[x];
                vb[i] = s / Cb[i * B + i];
                quad += vb[i] * vb[i];
            }
            const double dml = (am ? 0.5 * logdet : -0.5 * logdet) + 0.5 * quad + g.c0[(long)n * N + m];
            const double rho = g.rho[(long)n * N + m];
            int v;
            if (rho == 0.0 || rho == 1.0) {
                v = 0;   // reference :298/:307: 0*log(0) = NaN reaches sample_discrete_from_log, which then returns 0
            } else {
                const double d = dml + log(rho) - log(1.0 - rho);       // lps[1] - lps[0]
                my_lo = d;
                const double mx = d > 0.0 ? d : 0.0;
                const double e0 = exp(-mx), e1 = exp(d - mx);           // exp(lps - max)
                const double uu = g.u[(long)n * N + k0 + kk];
                v = (uu * (e0 + e1) > e0) ? 1 : 0;                      // cum = [e0, e0+e1]; count(r > cum)
            }
            my_v = v; my_am = am; my_ok = ok;
            if (v != am) atomicMin(&s_first, kk);
        }
        __syncthreads();
        const int kf = s_first;                  // first proposal of this round that flips (nblk: none)
        if (kk < nblk && kk <= kf) {             // commit: these saw the sub-tableau the serial order shows them
            const int m = perm[k0 + kk];
            if (!my_ok) atomicOr(&g.status[n], 1);
            if (g.logodds) g.logodds[(long)n * N + k0 + kk] = my_lo;
            s_flipped[kk] = (my_v != my_am) ? (my_v ? 1 : -1) : 0;
            g.a[(long)n * N + m] = my_v;
            if (kk == kf) s_sign = my_v ? 1 : -1;   // forward sweep when switching on
        }
        if (tid == 0) s_flip = kf < nblk;
        const int p0 = (kf < nblk ? kf : 0) * B;
        const double* Cb = Cbs + (size_t)(kf < nblk ? kf : 0) * B * B;
        k = kf + 1;
        __syncthreads();
        const int do_flip = s_flip, flip_sign = s_sign;   // block-uniform; re-read only after the barrier below
        __syncthreads();
        if (do_flip) {
            // Cinv = (L[p,p])^-1 = sgn * Q^-1, column x solved by thread x from the Cholesky factor
            if (tid < B) {
                const int x = tid;
                double col[32];
                for (int i = 0; i < B; ++i) {            // forward Lc y = e_x
                    double s = (i == x) ? 1.0 : 0.0;
                    for (int j = 0; j < i; ++j) s -= Cb[i * B + j] * col[j];
                    col[i] = s / Cb[i * B + i];
                }
                for (int i = B - 1; i >= 0; --i) {       // backward Lc' z = y
                    double s = col[i];
                    for (int j = i + 1; j < B; ++j) s -= Cb[j * B + i] * col[j];
                    col[i] = s / Cb[i * B + i];
                }
                const double sg = (flip_sign > 0) ? 1.0 : -1.0;   // forward: block was inactive -> L[p,p] = +Q
                for (int i = 0; i < B; ++i) Cinv[i * B + x] = sg * col[i];
            }
            __syncthreads();
            for (int e = tid; e < nl * B; e += nthr) {    // Tm = L[:,p] Cinv
                const int i = e / B, x = e % B;
                double s = 0.0;
                for (int y = 0; y < B; ++y) s += Ls(i, p0 + y) * Cinv[y * B + x];
                Tm[e] = s;
            }
            __syncthreads();
            for (int e = tid; e < nl * B; e += nthr) {    // pivot rows -> LDS
                const int x = e / nl, j = e % nl;
                Prow[e] = Ls(p0 + x, j);
            }
            __syncthreads();
            // non-pivot entries: the sub-tableau lives in global memory (L2), so the read-modify-writes are issued in batches of
            // 8 independent loads per thread instead of one dependent load/store at a time
            // (lower triangle only, folded into a rectangle so that an element's (i, j) costs one division as before: row r of the
            // rectangle is row nl-1-r of the triangle followed by row r-1 (nl odd) or r (nl even))
            if constexpr (BT > 0) {
                // 4 x 4 register tiles of the lower triangle (two halves of 2 rows): the B pivot-row values of the tile's 4 columns and the
                // B multipliers of its rows are read from LDS once per tile -- 2.5 LDS reads and no integer division per entry, where the
                // entry-per-thread loop below spends 10 and two divisions (it was ~60 % of a flip).  Same expression per entry: same bits.
                // Only LIVE rows and columns are updated: those of the proposals still to come (and h), and those of the blocks that have
                // flipped (the window's G is read from them at the end).  The rows of a block that was proposed and did not flip are never
                // read again -- after k of 64 proposals with 15 % flips that is half the triangle.  The sub-tableau does not fit the L2 of
                // an XCD (32 workgroups x 413 KB), so a flip is bound by the traffic to the memory-side cache: fewer bytes, not fewer flops.
                // Compacted index a -> row: the flipped rows so far (s_fl, ascending), then the rows from the next proposal's on.
                const int nfl = s_nfl, r_live = p0 + BT, nc = nfl + (nl - r_live);
                auto row_of = [&](int a) { return a < nfl ? s_fl[a] : r_live + (a - nfl); };
                const int nt = (nc + 3) / 4, ntile = nt * (nt + 1) / 2;
                for (int t = tid; t < ntile; t += nthr) {
                    int ti = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
                    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
                    while (ti * (ti + 1) / 2 > t) --ti;
                    const int tj = t - ti * (ti + 1) / 2, i0 = ti * 4, j0 = tj * 4;
                    int jc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) jc[c] = j0 + c < nc ? row_of(j0 + c) : -1;
                    double pr[BT][4];
#pragma unroll
                    for (int x = 0; x < BT; ++x)
#pragma unroll
                        for (int c = 0; c < 4; ++c) pr[x][c] = jc[c] >= 0 ? Prow[x * nl + jc[c]] : 0.0;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        double tm[2][BT], old_[2][4];
                        int ir[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int a = i0 + half * 2 + r;
                            ir[r] = a < nc ? row_of(a) : -1;
#pragma unroll
                            for (int x = 0; x < BT; ++x) tm[r][x] = ir[r] >= 0 ? Tm[ir[r] * BT + x] : 0.0;
#pragma unroll
                            for (int c = 0; c < 4; ++c) old_[r][c] = (ir[r] >= 0 && j0 + c <= a) ? L[ir[r] * ldl + jc[c]] : 0.0;
                        }
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int a = i0 + half * 2 + r;
                            if (ir[r] < 0) continue;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (j0 + c > a) continue;
                                double sum = 0.0;
#pragma unroll
                                for (int x = 0; x < BT; ++x) sum += tm[r][x] * pr[x][c];
                                L[ir[r] * ldl + jc[c]] = old_[r][c] - sum;
                            }
                        }
                    }
                }
            } else {
            const bool odd = nl & 1;
            const int wdt = odd ? nl : nl + 1, ntri = wdt * (odd ? (nl + 1) / 2 : nl / 2);
            for (int e0 = tid; e0 < ntri; e0 += nthr * 8) {
                double old_[8];
                int ii[8], jj[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = e0 + q * nthr;
                    const int r = e / wdt, c = e - r * wdt, first = nl - r;
                    const int i = c < first ? nl - 1 - r : (odd ? r - 1 : r);
                    ii[q] = i; jj[q] = c < first ? c : c - first;
                    old_[q] = (e < ntri) ? L[i * ldl + jj[q]] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = e0 + q * nthr;
                    if (e >= ntri) continue;
                    const int i = ii[q], j = jj[q];
                    if ((i >= p0 && i < p0 + B) || (j >= p0 && j < p0 + B)) continue;
                    double sum = 0.0;
                    for (int x = 0; x < B; ++x) sum += Tm[i * B + x] * Prow[x * nl + j];
                    L[i * ldl + j] = old_[q] - sum;
                }
            }
            }
            __syncthreads();
            const double sg = (flip_sign > 0) ? 1.0 : -1.0;
            for (int e = tid; e < nl * B; e += nthr) {    // pivot rows / columns
                const int i = e / B, x = e % B;
                if (i >= p0 && i < p0 + B) { if (p0 + x <= i) L[i * ldl + p0 + x] = -Cinv[(i - p0) * B + x]; }
                else { const double val = sg * Tm[e]; if (i > p0 + x) L[i * ldl + p0 + x] = val; else L[(p0 + x) * ldl + i] = val; }
            }
            if (tid < B) s_fl[s_nfl + tid] = p0 + tid;      // (s_nfl was last read before the barrier above)
            __syncthreads();
            if (tid == 0) s_nfl += B;
        }
    }
    if (tid == 0) {
        int cnt = 0;
        for (int k = 0; k < nblk; ++k)
            if (s_flipped[k]) {
                const int m = g.permuted ? k0 + k : perm[k0 + k];
                for (int b = 0; b < B; ++b) { g.d_idx[(long)n * KMAX + cnt] = m * B + b; g.d_sign[(long)n * KMAX + cnt] = (double)s_flipped[k]; ++cnt; }
            }
        g.d_cnt[n] = cnt;
        g.batch_k[n] = (cnt + 15) & ~15;
    }
    __syncthreads();
    // G = (M_DD)^-1 of the PRE-window tableau comes for free: after the local sweeps the flipped rows hold
    // M'_DD = -Sg G Sg, hence G[q][r] = -s_q s_r L[loc q][loc r]
    {
        __shared__ int s_loc[KMAX];
        __shared__ int s_cnt;
        if (tid == 0) {
            int cnt = 0;
            for (int k = 0; k < nblk; ++k)
                if (s_flipped[k])
                    for (int b = 0; b < B; ++b) s_loc[cnt++] = k * B + b;
            s_cnt = cnt;
        }
        __syncthreads();
        const int cnt = s_cnt;
        double* Gn = g.G + (long)n * KMAX * KMAX;
        const int kp = (cnt + 15) & ~15;
        for (int e = tid; e < kp * KMAX; e += nthr) {
            const int q = e / KMAX, r = e % KMAX;
            double v = 0.0;
            if (q < cnt && r < cnt) {
                const double sq = (double)s_flipped[s_loc[q] / B], sr = (double)s_flipped[s_loc[r] / B];
                v = -sq * sr * Ls(s_loc[q], s_loc[r]);
            }
            Gn[e] = v;
        }
    }
}

// ------------------------------------------------------------------ G = (M_DD)^-1 by in-order symmetric sweeps (every pivot block is definite)
__global__ __launch_bounds__(256) void invert_kernel(FlipArgs g) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int k = g.d_cnt[n];
    double* Gn = g.G + (long)n * KMAX * KMAX;
    if (k <= 0) { if (tid == 0) g.batch_k[n] = 0; return; }
    // chunks of up to 128 pivots (what the initial sweep uses) are inverted in LDS; larger lists fall back to the slow
    // in-place global path (proposal windows never come here: decide_kernel emits their G directly)
    const bool in_lds = k <= 128;
    const int ld = in_lds ? k + 1 : KMAX;
    double* colp = lds;              // [KMAX]
    double* A = in_lds ? lds + KMAX : Gn;
    const double* M = g.M + (long)n * g.strideM;
    const int* idx = g.d_idx + (long)n * KMAX;
    for (int e = tid; e < k * k; e += 256) {
        const int i = e / k, j = e % k;
        A[i * ld + j] = tab_get(M, g.ldj, idx[i], idx[j]);
    }
    __syncthreads();
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    for (int p = 0; p < k; ++p) {
        for (int i = tid; i < k; i += 256) colp[i] = A[i * ld + p];
        __syncthreads();
        const double d = colp[p];
        if (tid == 0 && !(fabs(d) > 0.0)) s_bad = 1;
        const double inv = 1.0 / d;
        for (int e = tid; e < k * k; e += 256) {
            const int i = e / k, j = e % k;
            double v;
            if (i == p && j == p) v = -inv;
            else if (i == p) v = colp[j] * inv;
            else if (j == p) v = colp[i] * inv;
            else v = A[i * ld + j] - colp[i] * colp[j] * inv;
            A[i * ld + j] = v;
        }
        __syncthreads();
    }
    // all-forward sweep of the whole block gives -A^-1
    if (in_lds) {
        for (int e = tid; e < KMAX * KMAX; e += 256) {
            const int i = e / KMAX, j = e % KMAX;
            Gn[e] = (i < k && j < k) ? -A[i * ld + j] : 0.0;
        }
    } else {
        __syncthreads();
        for (int e = tid; e < KMAX * KMAX; e += 256) {
            const int i = e / KMAX, j = e % KMAX;
            Gn[e] = (i < k && j < k) ? -Gn[e] : 0.0;
        }
    }
    if (tid == 0) { g.batch_k[n] = (k + 15) & ~15; if (s_bad) atomicOr(&g.status[n], 2); }
}

// ------------------------------------------------------------------ pivot lists of 129..256 rows: G by 2 x 2 blocks of 128
// (the initial sweep on the active set is cheaper per pivot with rank-256 updates: the tableau is read and written once per 256
// pivots and the update becomes MFMA-bound; the 256 x 256 inverse is assembled from two in-LDS 128 x 128 inversions and five
// 128^3 products on the MFMA kernel)   with T = G_A M_AB, S = M_BB - M_BA T:
//     G_BB = S^-1,   G_BA = -S^-1 T',   G_AA = G_A + T S^-1 T' = G_A - G_BA' T'
constexpr int KB2 = 128;            // block size
// Akk[i][j] = M[idx[i]][idx[j]] for i, j < k, identity elsewhere in the frame x frame corner (so the fixed-size block products and
// inversions need no per-neuron sizes)
__global__ __launch_bounds__(256) void gather_kk_kernel(FlipArgs g, int frame) {
    const int n = blockIdx.y, tid = threadIdx.x;
    const int k = g.d_cnt[n];
    double* A = g.Lws + (size_t)n * KMAX * KMAX;
    const double* M = g.M + (long)n * g.strideM;
    const int* idx = g.d_idx + (long)n * KMAX;
    const int e = blockIdx.x * 256 + tid;
    if (e >= frame * frame) return;
    const int i = e / frame, j = e % frame;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < k && j < k) v = tab_get(M, g.ldj, idx[i], idx[j]);
    A[i * KMAX + j] = v;
}

// dst[o+i][o+j] = (src[o.., o..])^-1 for the 128 x 128 block at offset o, by 128 in-order symmetric sweeps (the block is definite).
// The matrix lives in REGISTERS: thread (tr, tc) = (tid / 128, tid % 128) owns column tc, rows tr, tr+2, ... (64 doubles); each step
// publishes the pivot column through LDS (two owner threads write it, everyone reads it back as wave-wide broadcasts) and costs one
// FMA per element.  (The first version kept the matrix in LDS and paid four dependent LDS accesses per element: 1.27 ms per block.)
__global__ __launch_bounds__(256) void invert128_kernel(FlipArgs g, const double* src_base, double* dst_base, int o) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const int k = g.d_cnt[n];
    constexpr int kk = KB2;
    __shared__ double colp[2][kk];            // double-buffered pivot column
    __shared__ int s_bad;
    const double* src = src_base + (size_t)n * KMAX * KMAX;
    double* dst = dst_base + (size_t)n * KMAX * KMAX;
    if (k <= o) {                             // nothing real in this block (the frame holds the identity there)
        for (int e = tid; e < kk * kk; e += 256) dst[(o + e / kk) * KMAX + o + e % kk] = (e / kk == e % kk) ? 1.0 : 0.0;
        return;
    }
    const int tr = tid >> 7, tc = tid & (kk - 1);
    double a[kk / 2];
#pragma unroll
    for (int r = 0; r < kk / 2; ++r) a[r] = src[(o + tr + 2 * r) * KMAX + o + tc];
    if (tid == 0) s_bad = 0;
    if (tc == 0) {
#pragma unroll
        for (int r = 0; r < kk / 2; ++r) colp[0][tr + 2 * r] = a[r];
    }
    __syncthreads();
    for (int p = 0; p < kk; ++p) {
        const double* cp = colp[p & 1];
        const double d = cp[p];
        if (tid == 0 && !(d > 0.0)) s_bad = 1;
        const double inv = 1.0 / d;
        const bool pivcol = tc == p;
        // one fma and one select per element: the pivot column's thread runs the same update with cj = -inv on a zero column
        // (0 - ci (-inv) = ci inv, and -inv in the pivot row)
        const double cj = pivcol ? -inv : cp[tc] * inv;
#pragma unroll
        for (int r = 0; r < kk / 2; ++r) {
            const int i = tr + 2 * r;
            const double ci = cp[i];
            const double v = (pivcol ? 0.0 : a[r]) - ci * cj;
            a[r] = (i == p) ? cj : v;
        }
        if (tc == p + 1) {                    // the next pivot column, already updated
#pragma unroll
            for (int r = 0; r < kk / 2; ++r) colp[(p + 1) & 1][tr + 2 * r] = a[r];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < kk / 2; ++r) dst[(o + tr + 2 * r) * KMAX + o + tc] = -a[r];
    if (tid == 0 && s_bad) atomicOr(&g.status[n], 2);
}

// one level of the block inverse is done: G's off-diagonal blocks of the 2h x 2h square at offset o come from the scratch frame
// (G_BA = Akk[BA], G_AB = G_BA')
__global__ __launch_bounds__(256) void offdiag_g_kernel(FlipArgs g, int o, int h) {
    const int n = blockIdx.y;
    double* Gn = g.G + (size_t)n * KMAX * KMAX;
    const double* A = g.Lws + (size_t)n * KMAX * KMAX;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= h * h) return;
    const int i = e / h, j = e % h;                          // element (i, j) of the BA block
    const double v = A[(o + h + i) * KMAX + o + j];
    Gn[(o + h + i) * KMAX + o + j] = v;
    Gn[(o + j) * KMAX + o + h + i] = v;
}

// zeros outside the k x k corner (the frame holds the identity there), padded K for the panel products
__global__ __launch_bounds__(256) void mask_g_kernel(FlipArgs g) {
    const int n = blockIdx.y, tid = threadIdx.x;
    const int k = g.d_cnt[n];
    double* Gn = g.G + (size_t)n * KMAX * KMAX;
    const int e = blockIdx.x * 256 + tid;
    if (e == 0) g.batch_k[n] = (k + 15) & ~15;
    if (e >= KMAX * KMAX) return;
    const int i = e / KMAX, j = e % KMAX;
    if (i >= k || j >= k) Gn[e] = 0.0;
}

// ------------------------------------------------------------------ Ut[q][c] = M[idx[q], c]  (old panel, k-major), zero rows up to the padded K
// The tableau stores its lower triangle, so row idx[q] of the symmetric matrix is a stored row up to the diagonal and a stored
// COLUMN beyond it.  One workgroup moves a 64 (columns c) x 64 (pivots q) tile.  Where the whole tile lies left of the diagonal the
// stored rows are copied lane-per-column; elsewhere lanes run along q -- row c read at the columns idx[q..q+63], which sit in a few
// contiguous runs -- and the tile is turned in LDS so that the writes to Ut are again lane-per-column.
constexpr int GT = 64;
__global__ __launch_bounds__(256) void gather_panel_kernel(FlipArgs g) {
    const int n = blockIdx.z;
    const int k = g.d_cnt[n];
    if (k <= 0) return;
    const int kp = (k + 15) & ~15;
    const int q0 = blockIdx.y * GT, c0 = g.c_begin + blockIdx.x * GT;
    if (q0 >= kp) return;
    const int Md = g.N * g.B + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* M = g.M + (long)n * g.strideM;
    const int* idx = g.d_idx + (long)n * KMAX;
    double* Ut = g.Ut + (long)n * KMAX * g.ldu + (g.row_off ? (long)g.row_off[n] * g.ldu : 0);
    __shared__ int s_idx[GT];
    __shared__ int s_min;
    __shared__ double tile[GT][GT + 1];
    if (threadIdx.x < GT) s_idx[threadIdx.x] = (q0 + threadIdx.x < k) ? idx[q0 + threadIdx.x] : -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int mn = 0x7fffffff;
        for (int i = 0; i < GT; ++i) if (s_idx[i] >= 0 && s_idx[i] < mn) mn = s_idx[i];
        s_min = mn;
    }
    __syncthreads();
    const int c = c0 + lane;
    if (c0 + GT - 1 <= s_min) {                       // stored rows only
        if (c >= g.ldu) return;
        for (int qi = wave; qi < GT && q0 + qi < kp; qi += 4) {
            const int gq = s_idx[qi];
            Ut[(long)(q0 + qi) * g.ldu + c] = (gq >= 0 && c < Md) ? M[(long)gq * g.ldj + c] : 0.0;
        }
        return;
    }
    const int gq = s_idx[lane];
    for (int ci = wave; ci < GT; ci += 4) {
        const int cc = c0 + ci;
        tile[ci][lane] = (gq >= 0 && cc < Md) ? tab_get(M, g.ldj, gq, cc) : 0.0;
    }
    __syncthreads();
    if (c >= g.ldu) return;
    for (int qi = wave; qi < GT && q0 + qi < kp; qi += 4) Ut[(long)(q0 + qi) * g.ldu + c] = tile[lane][qi];
}

// ------------------------------------------------------------------ pivot rows/columns after the rank-k update
// M_sym[idx[q], c] = sg[q] Wt[q][c] for non-pivot c;  = -sg[q] G[q][r] sg[r] where c = idx[r] (written once, from the side gq >= c).
// Same 64 x 64 tiling and the same two access patterns as the gather, in the opposite direction.
__global__ __launch_bounds__(256) void fixup_kernel(FlipArgs g) {
    const int n = blockIdx.z;
    const int k = g.d_cnt[n];
    const int q0 = blockIdx.y * GT, c0 = blockIdx.x * GT;
    if (k <= 0 || q0 >= k) return;
    const int Md = g.N * g.B + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* M = g.M + (long)n * g.strideM;
    const int* idx = g.d_idx + (long)n * KMAX;
    const double* sgn = g.d_sign + (long)n * KMAX;
    const double* Wt = g.Wt + (long)n * KMAX * g.ldu;
    const double* Gn = g.G + (long)n * KMAX * KMAX;
    __shared__ int s_idx[GT];
    __shared__ double s_sg[GT];
    __shared__ int s_piv[GT];       // position in the pivot list of column c0 + i, or -1
    __shared__ int s_min;
    __shared__ double tile[GT][GT + 1];
    if (threadIdx.x < GT) {
        const bool ok = q0 + threadIdx.x < k;
        s_idx[threadIdx.x] = ok ? idx[q0 + threadIdx.x] : -1;
        s_sg[threadIdx.x] = ok ? sgn[q0 + threadIdx.x] : 0.0;
        s_piv[threadIdx.x] = -1;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < k; q += 256) {
        const int cc = idx[q] - c0;
        if (cc >= 0 && cc < GT) s_piv[cc] = q;
    }
    if (threadIdx.x == 0) {
        int mn = 0x7fffffff;
        for (int i = 0; i < GT; ++i) if (s_idx[i] >= 0 && s_idx[i] < mn) mn = s_idx[i];
        s_min = mn;
    }
    __syncthreads();
    const int c = c0 + lane;
    if (c0 + GT - 1 <= s_min) {                       // every pivot row of the tile is a stored row here
        if (c >= Md) return;
        const int r = s_piv[lane];
        const double sr = r >= 0 ? sgn[r] : 0.0;
        for (int qi = wave; qi < GT; qi += 4) {
            const int gq = s_idx[qi];
            if (gq < 0) break;
            const double v = r < 0 ? s_sg[qi] * Wt[(long)(q0 + qi) * g.ldu + c] : -s_sg[qi] * Gn[(q0 + qi) * KMAX + r] * sr;
            M[(long)gq * g.ldj + c] = v;
        }
        return;
    }
    for (int qi = wave; qi < GT; qi += 4)
        tile[qi][lane] = (s_idx[qi] >= 0 && c < Md) ? s_sg[qi] * Wt[(long)(q0 + qi) * g.ldu + c] : 0.0;
    __syncthreads();
    const int gq = s_idx[lane];
    if (gq < 0) return;
    for (int ci = wave; ci < GT; ci += 4) {
        const int cc = c0 + ci;
        if (cc >= Md) break;
        const int r = s_piv[ci];
        double v;
        if (r < 0) v = tile[lane][ci];
        else { if (gq < cc) continue; v = -s_sg[lane] * Gn[(q0 + lane) * KMAX + r] * sgn[r]; }
        if (gq >= cc) M[(long)gq * g.ldj + cc] = v; else M[(long)cc * g.ldj + gq] = v;
    }
}

// ------------------------------------------------------------------ the pivot rows / columns from the rank-k update itself (initial sweep)
// fixup_kernel rewrites the pivot rows and columns after the update -- 512 x (D + 2) entries per neuron and chunk, the column part as
// scattered 8-byte stores.  The update can produce them itself if the pivot COLUMNS of its two operands are patched first: with
//     Wt'[t][idx_q] = delta_tq - s_q G[t][q]          Ut'[t][idx_r] = M_DD[t][r] - s_r delta_tr      (all other columns as they are)
// the product M - Wt'' Ut' gives, because G M_DD = 1 and Wt = G Ut,
//     row idx_q, other columns c:      M[idx_q][c] - Ut[q][c] + s_q Wt[q][c]            =  s_q Wt[q][c]            (Ut[q][c] IS M[idx_q][c])
//     other rows i, column idx_r:      M[i][idx_r] - (Wt' M_DD)[i][r] + s_r Wt[r][i]    =  s_r Wt[r][i]
//     idx_q, idx_r:                    (s_q + s_r) delta_qr - s_q s_r G[q][r]           =  -s_q G[q][r] s_r  +  2 s_q delta_qr
// i.e. exactly what fixup_kernel writes, but for 2 s_q on the diagonal of the pivot block, which diag_fix_kernel takes off again (an exact
// operation).  A patch is k x k entries per operand instead of k x (D + 2): -4.9 % of the initial sweep at BASELINE configs[2], same decisions,
// same log-likelihood to 15 digits (profiles/r05_flips_pivot_patch_ab.txt).
// NOT THE DEFAULT (an experiment of the -DPGL_AB build, PGL_FLIP_PATCH=1): the row form cancels exactly (Ut[q][c] IS M[idx_q][c]), but the
// column form leaves ((1 - M_DD G) Ut)[r][i] behind -- the residual of G as an inverse, which the direct write of s Wt never sees.  On the
// 16 000-dimensional active systems of configs[4] the final tableau's residual |J_SS M_SS v + v| / |v| went from < 1e-8 to 2.4e-8 (3e-7 on the
// bias row, whose column of Ut is ten times larger): tests/test_gpu_fullsize.py::test_cfg5_shape_sweep_with_flips.  Accuracy was kept.
__global__ __launch_bounds__(256) void patch_pivot_cols_kernel(FlipArgs g) {
    const int n = blockIdx.y;
    const int k = g.d_cnt[n];
    if (k <= 0) return;
    const int kp = (k + 15) & ~15;
    const int* idx = g.d_idx + (long)n * KMAX;
    const double* sgn = g.d_sign + (long)n * KMAX;
    const double* Gn = g.G + (long)n * KMAX * KMAX;
    double* Wt = g.Wt + (long)n * KMAX * g.ldu;
    double* Ut = g.Ut + (long)n * KMAX * g.ldu;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < kp * k; e += gridDim.x * 256) {
        const int t = e / k, q = e - t * k;                     // q fastest: the B columns of a block are neighbours in a row of Wt
        const double sq = sgn[q];
        Wt[(long)t * g.ldu + idx[q]] = (t == q ? 1.0 : 0.0) - sq * Gn[(long)t * KMAX + q];
        if (t == q) Ut[(long)t * g.ldu + idx[q]] -= sq;
    }
}
__global__ __launch_bounds__(256) void diag_fix_kernel(FlipArgs g) {
    const int n = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.d_cnt[n]) return;
    const int i = g.d_idx[(long)n * KMAX + q];
    g.M[(long)n * g.strideM + (long)i * g.ldj + i] -= 2.0 * g.d_sign[(long)n * KMAX + q];
}

// ------------------------------------------------------------------ tableau in visit order
// M[n] = P J[n] P' (lower triangle), P = the neuron's proposal order: position k holds block perm[k]; the bias row D and the
// potential row D+1 stay last.  Once a window of positions has been proposed its rows are never read again, so the rank-k update after
// window w only has to touch the trailing square from position (w+1) R on (pgl_k_flip_apply, window form) -- a third of the
// full-tableau work summed over the windows -- and a window's sub-tableau is a contiguous diagonal square.
// One workgroup builds one DESTINATION block row pi (the B rows of position pi, columns up to and including its diagonal block): a thread owns
// a destination column, so every store is part of a coalesced row segment (the first version walked the SOURCE rows and scattered 40-byte
// pieces -- partial-line writes, 37 ms per batch of 256 at BASELINE configs[2] against ~11 ms for a plain copy).  The reads gather: block
// (pi, pj) is source block (mi, mj) = (perm[pi], perm[pj]) where mi >= mj -- B-double pieces of the B stored rows of block row mi, which this
// workgroup ends up reading almost entirely -- and the transpose of (mj, mi) otherwise: B contiguous doubles of a row of block row mj per
// thread.  blockIdx.y == N / N + 1: the bias and potential rows (columns gathered by position).
// BT = B at compile time for the common block sizes (the B values of a block then live in registers and their loads are all in flight
// together; with a run-time B they went through scratch one at a time: 36 ms per batch), 0 = any B.
template <int BT>
__global__ __launch_bounds__(256) void permute_tableau_kernel(FlipArgs g, const double* __restrict__ Jsrc, long lds_, long strideJ) {
    const int n = blockIdx.z, N = g.N, B = BT ? BT : g.B, D = N * B;
    const int* __restrict__ perm = g.perm + (long)n * N;
    const double* __restrict__ J = Jsrc + (long)n * strideJ;
    double* __restrict__ M = g.M + (long)n * g.strideM;
    const int pi = blockIdx.y;
    if (pi < N) {
        const int mi = perm[pi], ncol = (pi + 1) * B;
        for (int c = threadIdx.x; c < ncol; c += 256) {
            const int pj = c / B, y = c - pj * B, mj = perm[pj];
            double v[BT ? BT : 32];
            if (mi > mj) {
#pragma unroll
                for (int x = 0; x < B; ++x) v[x] = J[(long)(mi * B + x) * lds_ + mj * B + y];
            } else if (mi < mj) {
                const double* __restrict__ src = J + (long)(mj * B + y) * lds_ + mi * B;       // row of the mirrored block: B contiguous doubles
#pragma unroll
                for (int x = 0; x < B; ++x) v[x] = src[x];
            } else {
#pragma unroll
                for (int x = 0; x < B; ++x) v[x] = tab_get(J, lds_, mi * B + x, mi * B + y);   // diagonal block: mirrored above its diagonal
            }
#pragma unroll
            for (int x = 0; x < B; ++x) M[(long)(pi * B + x) * g.ldj + c] = v[x];
        }
    } else {
        const int row = D + (pi - N);            // D: bias row, D + 1: potential row
        for (int c = threadIdx.x; c <= row; c += 256) {
            const int sc = c < D ? perm[c / B] * B + c % B : c;
            M[(long)row * g.ldj + c] = J[(long)row * lds_ + sc];
        }
    }
}

// ------------------------------------------------------------------ pivot lists of the initial sweep, built on the device
// list[n] = (bias row, then the B rows of every active block in tableau order): the set S0 = {bias} U {active blocks} the tableau is
// first swept on, as row indices of the tableau (visit-order tableau: positions k with a[perm[k]] on, bias at D).  One workgroup per
// neuron; the active blocks are compacted with a ballot / prefix count per 256 positions, in order -- so the list is exactly what the
// host used to build with np.nonzero, and the chain does not depend on who builds it.
__global__ __launch_bounds__(256) void pivot_list_kernel(FlipArgs g, int* __restrict__ list, long ldl, int* __restrict__ count) {
    const int n = blockIdx.x, tid = threadIdx.x, N = g.N, B = g.B, D = N * B;
    int* L = list + (long)n * ldl;
    if (g.skip && g.skip[n]) { if (tid == 0) count[n] = 0; return; }
    const int* a = g.a + (long)n * N;
    const int* perm = g.perm + (long)n * N;
    __shared__ int s_wave[4];
    __shared__ int s_base;
    if (tid == 0) { s_base = 0; L[0] = D; }
    __syncthreads();
    for (int k0 = 0; k0 < N; k0 += 256) {
        const int k = k0 + tid;
        const int on = k < N ? (a[g.permuted ? perm[k] : k] != 0) : 0;
        const unsigned long long bal = __ballot(on);
        const int lane = tid & 63, wave = tid >> 6;
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (on) {
            const int p = off + before;
            for (int b = 0; b < B; ++b) L[1 + p * B + b] = k * B + b;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) count[n] = 1 + s_base * B;
}

// chunk c of every neuron's list -> the pending pivot list (d_idx, d_sign = +1, d_cnt)
__global__ __launch_bounds__(256) void pivot_chunk_kernel(FlipArgs g, const int* __restrict__ list, long ldl, const int* __restrict__ count, int chunk,
                                                          int per_chunk) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const int begin = chunk * per_chunk;
    int cnt = count[n] - begin;
    cnt = cnt < 0 ? 0 : cnt > per_chunk ? per_chunk : cnt;
    for (int j = tid; j < KMAX; j += 256) {
        g.d_idx[(long)n * KMAX + j] = j < cnt ? list[(long)n * ldl + begin + j] : 0;
        g.d_sign[(long)n * KMAX + j] = 1.0;
    }
    if (tid == 0) g.d_cnt[n] = cnt;
}

}  // namespace

size_t pgl_k_flip_lds_decide(int B, int R) {
    const int nl = R * B + 1;
    return (2 * (size_t)nl * B + (size_t)B * B + (size_t)R * (B * B + B)) * sizeof(double);
}


int pgl_k_flip_window_blocks(int B) {
    static const int kwin = [] { const int v = pgl_ab_int("PGL_FLIP_WINDOW", KWIN); return v >= 16 && v <= KMAX - 2 ? v : KWIN; }();
    int r = kwin / B;                                   // blocks per window: at most KMAX pivots ...
    while (r > 1 && pgl_k_flip_lds_decide(B, r) > 150 * 1024) --r;   // ... and the LDS scratch of decide_kernel must fit
    return r < 1 ? 0 : r;
}

// apply the pivot list currently in (d_idx, d_sign, d_cnt) to every neuron's tableau
// window >= 0 (visit-order tableau only): the pivots lie in proposal window `window`, whose rows -- like those of all earlier windows --
// are dead afterwards: only the trailing square is updated, the pivot rows and columns are not rewritten.
int pgl_k_flip_apply(const PglFlipState& s, int have_G, int max_pivots, int window, hipStream_t st) {
    const int R_ = pgl_k_flip_window_blocks(s.B);
    const int Md_ = s.N * s.B + 2;
    int r0 = 0;                      // first live row / column
    if (window >= 0 && s.permuted) {
        if ((long)(window + 1) * R_ >= s.N) return PGL_OK;          // last window: nothing is read afterwards
        r0 = ((window + 1) * R_ * s.B) & ~1;
    }
    FlipArgs g{s.M, s.ldj, s.strideM, s.N, s.B, R_, s.perm, s.u, s.rho, s.c0, s.a, s.skip, s.d_idx, s.d_sign, s.d_cnt,
               s.batch_k, s.G, s.Lws, s.Ut, s.Wt, s.ldu, s.status, s.permuted, r0, s.logodds};
    const size_t lds_inv = ((size_t)KMAX + 128 * 129) * sizeof(double);
    static PglPerDevice once;
    if (int rc = pgl_set_dynamic_lds(reinterpret_cast<const void*>(invert_kernel), lds_inv, once)) return rc;
    const int Md = Md_;
    if (!have_G && max_pivots > KB2) {
        if (max_pivots > KMAX) { pgl_set_error("flip_apply: %d pivots per call (max %d)", max_pivots, KMAX); return PGL_ERR_ARG; }
        // G = (M_PP)^-1 on a frame of 256 or 512 rows (identity beyond a neuron's own pivots), by recursive 2 x 2 blocking down to the
        // in-register 128 x 128 inversion:  T = G_A M_AB,  S = M_BB - M_BA T,  G_BB = S^-1,  G_BA = -S^-1 T',  G_AA = G_A - G_BA' T'
        const int frame = max_pivots > 2 * KB2 ? 4 * KB2 : 2 * KB2;
        const long sq = (long)KMAX * KMAX;
        double* Akk = s.Lws;
        hipLaunchKernelGGL(gather_kk_kernel, dim3(frame * frame / 256, s.nb), dim3(256), 0, st, g, frame);
        PGL_CHECK_LAUNCH();
        auto block_gemm = [&](const double* A, const double* Bm, double* C, double alpha, double beta, int h) {
            PglGemmArgs q{};
            q.A = A; q.lda = KMAX; q.strideA = sq; q.B = Bm; q.ldb = KMAX; q.strideB = sq; q.C = C; q.ldc = KMAX; q.strideC = sq;
            q.M = h; q.N = h; q.K = h; q.a_cols = h; q.b_cols = h; q.nbatch = s.nb; q.alpha = alpha; q.beta = beta; q.tri = 0;
            return pgl_launch_gemm(PGL_GEMM_PLAIN, q, st);
        };
        struct Rec {
            static int inv(const decltype(block_gemm)& gemm, const PglFlipState& s, const FlipArgs& g, double* Akk, int o, int n, hipStream_t st) {
                if (n == KB2) {
                    hipLaunchKernelGGL(invert128_kernel, dim3(s.nb), dim3(256), 0, st, g, (const double*)Akk, s.G, o);
                    PGL_CHECK_LAUNCH();
                    return PGL_OK;
                }
                const int h = n / 2;
                int rc = inv(gemm, s, g, Akk, o, h, st);                                               if (rc) return rc;   // G_AA = M_AA^-1
                double* G_AA = s.G + (long)o * KMAX + o;             double* T = s.G + (long)o * KMAX + o + h;              // T in G's upper-right block
                double* Tt = s.G + (long)(o + h) * KMAX + o;         double* G_BB = s.G + (long)(o + h) * KMAX + o + h;
                const double* M_AB = Akk + (long)o * KMAX + o + h;   double* S = Akk + (long)(o + h) * KMAX + o + h;
                double* G_BA = Akk + (long)(o + h) * KMAX + o;
                rc = gemm(G_AA, M_AB, T, 1.0, 0.0, h);                                                  if (rc) return rc;   // T  = G_A M_AB
                rc = gemm(M_AB, G_AA, Tt, 1.0, 0.0, h);                                                 if (rc) return rc;   // T' = M_BA G_A
                rc = gemm(M_AB, T, S, -1.0, 1.0, h);                                                    if (rc) return rc;   // S  = M_BB - M_BA T
                rc = inv(gemm, s, g, Akk, o + h, h, st);                                                if (rc) return rc;   // G_BB = S^-1
                rc = gemm(G_BB, Tt, G_BA, -1.0, 0.0, h);                                                if (rc) return rc;   // G_BA = -S^-1 T'
                rc = gemm(G_BA, Tt, G_AA, -1.0, 1.0, h);                                                if (rc) return rc;   // G_AA = G_A - G_BA' T'
                hipLaunchKernelGGL(offdiag_g_kernel, dim3((h * h + 255) / 256, s.nb), dim3(256), 0, st, g, o, h);
                PGL_CHECK_LAUNCH();
                return PGL_OK;
            }
        };
        int rc = Rec::inv(block_gemm, s, g, Akk, 0, frame, st);
        if (rc) return rc;
        hipLaunchKernelGGL(mask_g_kernel, dim3((KMAX * KMAX + 255) / 256, s.nb), dim3(256), 0, st, g);
        PGL_CHECK_LAUNCH();
    } else if (!have_G) {
        hipLaunchKernelGGL(invert_kernel, dim3(s.nb), dim3(256), lds_inv, st, g);
        PGL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(gather_panel_kernel, dim3((unsigned)((s.ldu - r0 + GT - 1) / GT), KMAX / GT, s.nb), dim3(256), 0, st, g);
    PGL_CHECK_LAUNCH();
    // Wt = G Ut   (K x ldu), then  M -= Wt' Ut  on lower-triangular tiles (columns / rows from r0 on)
    const int ncol = (int)s.ldu - r0;
    PglGemmArgs w{};
    w.A = s.G; w.lda = KMAX; w.strideA = (long)KMAX * KMAX;
    w.B = s.Ut + r0; w.ldb = s.ldu; w.strideB = (long)KMAX * s.ldu;
    w.C = s.Wt + r0; w.ldc = s.ldu; w.strideC = (long)KMAX * s.ldu;
    w.M = KMAX; w.N = ncol; w.K = KMAX; w.a_cols = KMAX; w.b_cols = ncol; w.nbatch = s.nb; w.nz_total = 0;
    w.alpha = 1.0; w.beta = 0.0; w.tri = 0; w.batch_k = s.batch_k; w.batch_dim = s.batch_k; w.dim_off = 0; w.dim_mode = 2; w.W = nullptr; w.ldw = 0;   // only the first K rows of W are non-zero / used
    w.pipe = 1;
    int rc = pgl_launch_gemm(PGL_GEMM_PLAIN, w, st);
    if (rc) return rc;
    // full-tableau form (initial sweep): the update writes the pivot rows and columns itself from patched operand columns
    static const bool patch_ab = pgl_ab_int("PGL_FLIP_PATCH", 0) != 0;       // (an experiment of the -DPGL_AB build: see patch_pivot_cols_kernel)
    const bool patch = r0 == 0 && patch_ab;
    if (patch) {
        hipLaunchKernelGGL(patch_pivot_cols_kernel, dim3(64, s.nb), dim3(256), 0, st, g);
        PGL_CHECK_LAUNCH();
    }
    PglGemmArgs t{};
    t.A = s.Wt + r0; t.lda = s.ldu; t.strideA = (long)KMAX * s.ldu;
    t.B = s.Ut + r0; t.ldb = s.ldu; t.strideB = (long)KMAX * s.ldu;
    t.C = s.M + (long)r0 * s.ldj + r0; t.ldc = s.ldj; t.strideC = s.strideM;
    t.M = Md - r0; t.N = Md - r0; t.K = KMAX; t.a_cols = ncol; t.b_cols = ncol; t.nbatch = s.nb; t.nz_total = 0;
    t.alpha = -1.0; t.beta = 1.0; t.tri = 1; t.batch_k = s.batch_k; t.batch_dim = nullptr; t.dim_off = 0; t.W = nullptr; t.ldw = 0;
    t.pipe = 1;
    rc = pgl_launch_gemm(PGL_GEMM_TRI1, t, st);
    if (rc) return rc;
    if (patch) {
        hipLaunchKernelGGL(diag_fix_kernel, dim3(KMAX / 256, s.nb), dim3(256), 0, st, g);
        PGL_CHECK_LAUNCH();
    } else if (r0 == 0) {            // (trailing form: the pivot rows / columns are dead, nothing to rewrite)
        hipLaunchKernelGGL(fixup_kernel, dim3((Md + GT - 1) / GT, KMAX / GT, s.nb), dim3(256), 0, st, g);
        PGL_CHECK_LAUNCH();
    }
    return PGL_OK;
}

// ---- two proposal windows per pass over the trailing tableau (visit-order tableau)
// The update after a window is bound by the read-modify-write of the trailing tableau (rank ~50 at the chain's flip rates), so the panels of
// windows w and w + 1 are STACKED and applied in one pass: after window w only the column strip of window w + 1 is brought up to date (a
// skinny product: that is all the proposals of w + 1 and the gather of their pivot rows read), the pivot rows of w + 1 go behind those of w in
// the panel buffers (per-neuron row offset = the first panel's padded length), and one product of rank k_w + k_{w+1} updates the rest.  Every
// entry sees the same multiply-adds in the same order as with one pass per window: same bits.  A neuron whose first panel would not leave
// room for a full second one (k_w > KMAX - window rows) takes the ordinary pass after window w (its batch_k is zero in the pair's products).
// ws: 4 x nb ints of scratch (off, kfull, kstrip, kcomb).
__global__ __launch_bounds__(256) void pair_plan_first_kernel(const int* __restrict__ batch_k, int nb, int room, int* __restrict__ ws) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= nb) return;
    const int k1 = batch_k[n];
    const bool paired = k1 + room <= KMAX;
    ws[n] = paired ? k1 : 0;                 // row offset of the second panel
    ws[nb + n] = paired ? 0 : k1;            // rank of the ordinary pass (unpaired neurons)
    ws[2 * nb + n] = paired ? k1 : 0;        // rank of the strip update
}
__global__ __launch_bounds__(256) void pair_plan_second_kernel(const int* __restrict__ batch_k, int nb, int* __restrict__ ws) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < nb) ws[3 * nb + n] = ws[n] + batch_k[n];
}

int pgl_k_flip_apply_pair(const PglFlipState& s, int phase, int window, int* ws, hipStream_t st) {
    const int R_ = pgl_k_flip_window_blocks(s.B);
    const int Md = s.N * s.B + 2;
    if (!s.permuted || (long)(window + 1) * R_ >= s.N) return PGL_OK;          // last window: nothing is read afterwards
    const int r0 = ((window + 1) * R_ * s.B) & ~1;                                // first live row / column after `window`
    FlipArgs g{s.M, s.ldj, s.strideM, s.N, s.B, R_, s.perm, s.u, s.rho, s.c0, s.a, s.skip, s.d_idx, s.d_sign, s.d_cnt,
               s.batch_k, s.G, s.Lws, s.Ut, s.Wt, s.ldu, s.status, s.permuted, r0, s.logodds, phase ? ws : nullptr};
    hipLaunchKernelGGL(gather_panel_kernel, dim3((unsigned)((s.ldu - r0 + GT - 1) / GT), KMAX / GT, s.nb), dim3(256), 0, st, g);
    PGL_CHECK_LAUNCH();
    const int ncol = (int)s.ldu - r0;
    PglGemmArgs w{};
    w.A = s.G; w.lda = KMAX; w.strideA = (long)KMAX * KMAX;
    w.B = s.Ut + r0; w.ldb = s.ldu; w.strideB = (long)KMAX * s.ldu;
    w.C = s.Wt + r0; w.ldc = s.ldu; w.strideC = (long)KMAX * s.ldu;
    w.M = KMAX; w.N = ncol; w.K = KMAX; w.a_cols = KMAX; w.b_cols = ncol; w.nbatch = s.nb;
    w.alpha = 1.0; w.beta = 0.0; w.tri = 0; w.batch_k = s.batch_k; w.batch_dim = s.batch_k; w.dim_off = 0; w.dim_mode = 2;
    w.batch_row_off = phase ? ws : nullptr;
    w.pipe = 1;
    if (int rc = pgl_launch_gemm(PGL_GEMM_PLAIN, w, st)) return rc;
    PglGemmArgs t{};
    t.A = s.Wt + r0; t.lda = s.ldu; t.strideA = (long)KMAX * s.ldu;
    t.B = s.Ut + r0; t.ldb = s.ldu; t.strideB = (long)KMAX * s.ldu;
    t.C = s.M + (long)r0 * s.ldj + r0; t.ldc = s.ldj; t.strideC = s.strideM;
    t.M = Md - r0; t.N = Md - r0; t.K = KMAX; t.a_cols = ncol; t.b_cols = ncol; t.nbatch = s.nb;
    t.alpha = -1.0; t.beta = 1.0; t.tri = 1; t.pipe = 1;
    if (phase == 0) {
        const int room = (R_ * s.B + 15) & ~15;
        hipLaunchKernelGGL(pair_plan_first_kernel, dim3((s.nb + 255) / 256), dim3(256), 0, st, s.batch_k, s.nb, room, ws);
        PGL_CHECK_LAUNCH();
        t.batch_k = ws + s.nb;                        // the ordinary pass, for the neurons that cannot pair
        if (int rc = pgl_launch_gemm(PGL_GEMM_TRI1, t, st)) return rc;
        // the column strip of window + 1 (its rows down to the last): all that window's proposals and pivot-row gather read
        const int nblk1 = min(R_, s.N - (window + 1) * R_);
        PglGemmArgs q = t;
        q.N = (window + 1) * R_ * s.B + nblk1 * s.B - r0; q.tri = 0; q.pipe = 0; q.batch_k = ws + 2 * s.nb;
        return pgl_launch_gemm(PGL_GEMM_PLAIN, q, st);
    }
    hipLaunchKernelGGL(pair_plan_second_kernel, dim3((s.nb + 255) / 256), dim3(256), 0, st, s.batch_k, s.nb, ws);
    PGL_CHECK_LAUNCH();
    t.batch_k = ws + 3 * s.nb;                        // both panels, stacked
    return pgl_launch_gemm(PGL_GEMM_TRI1, t, st);
}

int pgl_k_flip_permute(const PglFlipState& s, const double* J, long ldjs, long strideJ, hipStream_t st) {
    FlipArgs g{s.M, s.ldj, s.strideM, s.N, s.B, 0, s.perm, s.u, s.rho, s.c0, s.a, s.skip, s.d_idx, s.d_sign, s.d_cnt,
               s.batch_k, s.G, s.Lws, s.Ut, s.Wt, s.ldu, s.status, 1, 0, s.logodds};
    const dim3 grid(1, s.N + 2, s.nb);
#define PGL_PERMUTE(b_) case b_: hipLaunchKernelGGL(permute_tableau_kernel<b_>, grid, dim3(256), 0, st, g, J, ldjs, strideJ); break;
    switch (s.B) {
        PGL_PERMUTE(1) PGL_PERMUTE(2) PGL_PERMUTE(3) PGL_PERMUTE(4) PGL_PERMUTE(5) PGL_PERMUTE(6) PGL_PERMUTE(7) PGL_PERMUTE(8)
        default: hipLaunchKernelGGL(permute_tableau_kernel<0>, grid, dim3(256), 0, st, g, J, ldjs, strideJ); break;
    }
#undef PGL_PERMUTE
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_flip_decide(const PglFlipState& s, int window, hipStream_t st) {
    const int R = pgl_k_flip_window_blocks(s.B);
    if (R < 1) { pgl_set_error("B=%d exceeds the window capacity %d", s.B, KWIN); return PGL_ERR_ARG; }
    FlipArgs g{s.M, s.ldj, s.strideM, s.N, s.B, R, s.perm, s.u, s.rho, s.c0, s.a, s.skip, s.d_idx, s.d_sign, s.d_cnt,
               s.batch_k, s.G, s.Lws, s.Ut, s.Wt, s.ldu, s.status, s.permuted, 0, s.logodds};
    const size_t lds = pgl_k_flip_lds_decide(s.B, R);
    // (one LDS high-water mark for all instantiations: the request is raised on each of them whenever any launch needs more)
#define PGL_DECIDE(b_)                                                                                                      \
    case b_: {                                                                                                              \
        static PglPerDeviceSize lds_set_b;                                                                                  \
        if (int rc = pgl_grow_dynamic_lds(reinterpret_cast<const void*>(decide_kernel<b_>), lds, lds_set_b)) return rc;     \
        hipLaunchKernelGGL(decide_kernel<b_>, dim3(s.nb), dim3(1024), lds, st, g, window);                                  \
    } break;
    switch (s.B <= 8 ? s.B : 0) {
        PGL_DECIDE(0) PGL_DECIDE(1) PGL_DECIDE(2) PGL_DECIDE(3) PGL_DECIDE(4) PGL_DECIDE(5) PGL_DECIDE(6) PGL_DECIDE(7) PGL_DECIDE(8)
        default: break;
    }
#undef PGL_DECIDE
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_flip_pivot_list(const PglFlipState& s, int* list, long ldl, int* count, hipStream_t st) {
    FlipArgs g{s.M, s.ldj, s.strideM, s.N, s.B, 0, s.perm, s.u, s.rho, s.c0, s.a, s.skip, s.d_idx, s.d_sign, s.d_cnt,
               s.batch_k, s.G, s.Lws, s.Ut, s.Wt, s.ldu, s.status, s.permuted, 0, s.logodds};
    hipLaunchKernelGGL(pivot_list_kernel, dim3(s.nb), dim3(256), 0, st, g, list, ldl, count);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_flip_pivot_chunk(const PglFlipState& s, const int* list, long ldl, const int* count, int chunk, int per_chunk, hipStream_t st) {
    if (per_chunk < 1 || per_chunk > KMAX) { pgl_set_error("pivot chunk of %d rows (max %d)", per_chunk, KMAX); return PGL_ERR_ARG; }
    FlipArgs g{s.M, s.ldj, s.strideM, s.N, s.B, 0, s.perm, s.u, s.rho, s.c0, s.a, s.skip, s.d_idx, s.d_sign, s.d_cnt,
               s.batch_k, s.G, s.Lws, s.Ut, s.Wt, s.ldu, s.status, s.permuted, 0, s.logodds};
    hipLaunchKernelGGL(pivot_chunk_kernel, dim3(s.nb), dim3(256), 0, st, g, list, ldl, count, chunk, per_chunk);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_flip_kmax(void) { return KMAX; }
