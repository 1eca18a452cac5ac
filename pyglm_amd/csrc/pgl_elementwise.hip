// Streaming (HBM-bound) kernels of the Gibbs sweep for gfx950: Polya-gamma draws, kappa and the spike-train
// log-likelihood in one pass over Psi; causal basis convolution that builds the design matrix in HBM;
// assembly of the posterior precision from the Gram tiles, the border sums and the block-diagonal prior.
#include "pgl_common.h"
#include "pgl_rng.h"

namespace {

// ------------------------------------------------------------------ stream-parity hook
__global__ void philox_words_kernel(uint64_t seed, uint32_t purpose, uint32_t j, uint64_t elem0, uint64_t stream, uint32_t* out, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o0, o1, o2, o3;
    pgl_philox4x32_10(j | (purpose << 24), (uint32_t)(elem0 + i), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed,
                      (uint32_t)(seed >> 32), o0, o1, o2, o3);
    reinterpret_cast<uint4*>(out)[i] = make_uint4(o0, o1, o2, o3);
}

// ------------------------------------------------------------------ pgdrawvpar replacement (regression.py:504-507)
__global__ __launch_bounds__(256) void pg_draw_kernel(const double* __restrict__ b, const double* __restrict__ z, double* __restrict__ out,
                                                      size_t len, uint64_t seed, uint64_t stream, uint64_t elem0) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= len) return;
    const double bi = b ? b[i] : 1.0;
    out[i] = pgl_pg_draw(bi, z[i], seed, stream, elem0 + i);
}

// ------------------------------------------------------------------ psi -> (omega, kappa, log-lik)   regression.py:491-511
// Psi is T x nloc (time-major, from the activation GEMM, bias not yet added).  Lane = neuron column, wave = row:
// a wave reads/writes 64 consecutive doubles of one time bin.  Each block covers ROWS time bins of one 64-neuron
// column group and leaves one log-likelihood partial per neuron; a second pass adds partials in a fixed order
// (deterministic; no atomics).
constexpr int PGLL_ROWS = 64;

struct PgLlArgs {
    double* Psi; long ldpsi;            // in: X.w   out: psi = X.w + bias   [T][ldpsi]
    const double* bias;                 // [nloc]
    const double* Y; long ldy;          // spikes/counts of the local neurons: Y[t*ldy + n]
    double* Omega; long ldo;            // out [T][ldo]   (may be null: log-likelihood only)
    double* Kappa; long ldk;            // out [T][ldk]   (may be null)
    double* llpart;                     // [nblk_t][nloc]
    int T, nloc;
    int obs;                            // 0 Bernoulli (a=y,b=1,c=1)  1 negative binomial (a=y, b=y+xi, c=C(y+xi-1,y))
                                        // 2 Gaussian (regression.py:380-446): omega = 1/eta, kappa = y/eta, "ll" = sum of squared residuals
    double xi;
    const double* inv_eta;              // [nloc] 1/eta per neuron (obs == 2 only)
    uint64_t seed, sweep, neuron0, elem0;
};

// one time bin's term of the log-likelihood (regression.py:491-494); one function for both kernels below, so that they round alike
__device__ __forceinline__ double pg_ll_term(double logc, double a, double b, double psi) { return logc + a * psi - b * log1p(exp(psi)); }

__global__ __launch_bounds__(256) void pg_loglik_kernel(PgLlArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + lane;
    const int t0 = blockIdx.x * PGLL_ROWS;
    __shared__ double red[4][64];
    double ll = 0.0;
    if (n < g.nloc) {
        const double bn = g.bias ? g.bias[n] : 0.0;
        const uint64_t stream = ((uint64_t)g.sweep << 32) | (uint64_t)(uint32_t)(g.neuron0 + n);
        for (int r = wave; r < PGLL_ROWS; r += 4) {
            const int t = t0 + r;
            if (t >= g.T) break;
            const double psi = g.Psi[(long)t * g.ldpsi + n] + bn;
            g.Psi[(long)t * g.ldpsi + n] = psi;
            const double y = g.Y[(long)t * g.ldy + n];
            if (g.obs == 2) {
                const double ie = g.inv_eta[n], r = y - psi;
                ll += r * r;
                if (g.Kappa) g.Kappa[(long)t * g.ldk + n] = y * ie;
                if (g.Omega) g.Omega[(long)t * g.ldo + n] = ie;
                continue;
            }
            double a = y, b = 1.0, logc = 0.0;
            if (g.obs == 1) { b = y + g.xi; logc = lgamma(y + g.xi) - lgamma(y + 1.0) - lgamma(g.xi); }
            ll += pg_ll_term(logc, a, b, psi);
            if (g.Kappa) g.Kappa[(long)t * g.ldk + n] = a - 0.5 * b;
            if (g.Omega) g.Omega[(long)t * g.ldo + n] = pgl_pg_draw(b, psi, g.seed, stream, g.elem0 + (uint64_t)t);
        }
    }
    red[wave][lane] = ll;
    __syncthreads();
    if (wave == 0 && n < g.nloc) g.llpart[(long)blockIdx.x * g.nloc + n] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// The same for a NARROW shard (fewer than 64 local neurons: a small model, BASELINE configs[0]): with a lane per neuron most lanes would idle
// and every busy one walk 16 time bins one after the other (0.37 of that sweep's 1.0 ms of GPU time at N = 4).  Here the block's
// PGLL_ROWS x nloc cells are dealt to its 256 threads, each cell's log-likelihood term goes to LDS, and then thread (wave, neuron) adds ITS rows'
// terms in the order the kernel above adds them -- the same numbers in the same order: the same log-likelihood to the last bit, whatever the shard.
__global__ __launch_bounds__(256) void pg_loglik_narrow_kernel(PgLlArgs g) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = g.nloc;                                    // < 64
    const int t0 = blockIdx.x * PGLL_ROWS;
    __shared__ double term[PGLL_ROWS][64];
    __shared__ double red[4][64];
    for (int c = tid; c < PGLL_ROWS * nl; c += 256) {
        const int r = c / nl, n = c - r * nl, t = t0 + r;
        double v = 0.0;
        if (t < g.T) {
            const double bn = g.bias ? g.bias[n] : 0.0;
            const uint64_t stream = ((uint64_t)g.sweep << 32) | (uint64_t)(uint32_t)(g.neuron0 + n);
            const double psi = g.Psi[(long)t * g.ldpsi + n] + bn;
            g.Psi[(long)t * g.ldpsi + n] = psi;
            const double y = g.Y[(long)t * g.ldy + n];
            double a = y, b = 1.0, logc = 0.0;
            if (g.obs == 1) { b = y + g.xi; logc = lgamma(y + g.xi) - lgamma(y + 1.0) - lgamma(g.xi); }
            v = pg_ll_term(logc, a, b, psi);
            if (g.Kappa) g.Kappa[(long)t * g.ldk + n] = a - 0.5 * b;
            if (g.Omega) g.Omega[(long)t * g.ldo + n] = pgl_pg_draw(b, psi, g.seed, stream, g.elem0 + (uint64_t)t);
        }
        term[r][n] = v;
    }
    __syncthreads();
    double ll = 0.0;
    if (lane < nl)
        for (int r = wave; r < PGLL_ROWS; r += 4) {
            if (t0 + r >= g.T) break;
            ll += term[r][lane];
        }
    red[wave][lane] = ll;
    __syncthreads();
    if (wave == 0 && lane < nl) g.llpart[(long)blockIdx.x * nl + lane] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

__global__ void colsum_partials_kernel(const double* __restrict__ part, int nblk, int ncol, double* __restrict__ out, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= ncol) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(long)b * ncol + n];
    out[n] = accumulate ? out[n] + s : s;
}

// ------------------------------------------------------------------ design matrix (utils/basis.py:5-34)
// X[t][n*B+b] = sum_{l=0}^{R-1} basis[l][b] * S[t-1-l][n]   (strictly causal: the reference prepends a zero row, :18),
// clipped at 0 when clip != 0 (:30-32).  Also writes the ones column at index D (bias regressor for the border sums)
// and, when Xt != null, the transposed copy Xt[d][t] used by the activation GEMM.
struct ConvArgs {
    const double* S; long lds;      // [T][lds]
    const double* basis;            // [R][B]
    double* X; long ldx;            // [Tp][ldx]
    double* Xt; long ldxt;          // [Dp][ldxt] or null
    int T, N, B, R, clip;
};

__global__ __launch_bounds__(256) void basis_conv_kernel(ConvArgs g) {
    extern __shared__ double sb[];   // basis [R][B]
    for (int i = threadIdx.x; i < g.R * g.B; i += blockDim.x) sb[i] = g.basis[i];
    __syncthreads();
    const int D = g.N * g.B;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;   // over T x N
    if (idx >= (long)g.T * g.N) return;
    const int t = (int)(idx / g.N), n = (int)(idx % g.N);
    double acc[32];
    for (int b = 0; b < g.B; ++b) acc[b] = 0.0;
    const int lmax = min(g.R, t);
    for (int l = 0; l < lmax; ++l) {
        const double s = g.S[(long)(t - 1 - l) * g.lds + n];
        if (s != 0.0)
            for (int b = 0; b < g.B; ++b) acc[b] += sb[l * g.B + b] * s;
    }
    for (int b = 0; b < g.B; ++b) {
        double v = acc[b];
        if (g.clip && v < 0.0) v = 0.0;
        g.X[(long)t * g.ldx + n * g.B + b] = v;
        if (g.Xt) g.Xt[(long)(n * g.B + b) * g.ldxt + t] = v;
    }
    if (n == 0) {
        g.X[(long)t * g.ldx + D] = 1.0;
        if (g.Xt) g.Xt[(long)D * g.ldxt + t] = 1.0;
    }
}

// tiled transpose  dst[c][r] = src[r][c]
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ src, long lds_, double* __restrict__ dst, long ldd, int rows, int cols) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = by + i, c = bx + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(long)r * lds_ + c] : 0.0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = bx + i, r = by + tx;
        if (c < cols && r < rows) dst[(long)c * ldd + r] = tile[tx][i];
    }
}

// ------------------------------------------------------------------ posterior assembly (regression.py:210-223, 253-260, 270-271)
// J holds, per local neuron, the Gram tiles X'OX in rows/cols [0, D) (lower triangle valid).  This adds the
// block-diagonal prior precision, writes the bias row D  = [X'omega + 0, sum(omega) + J_b], and the potential row
// D+1 = [X'kappa + h_w, sum(kappa) + h_b, 0]; M = D+2.  One block per (neuron, 256-column chunk).
struct PostArgs {
    double* J; long ldj; long strideJ;
    const double* bo; const double* bk; long ldb;   // [nb][ldb] each: sum_t omega X~ and sum_t kappa X~ (col D = plain sums)
    const double* Jw;                 // [nloc][N][B][B], or (label != null) a table [K][B][B] of distinct blocks
    const double* hw;                 // [nloc][N][B],    or a table [K][B]
    const int* label;                 // [nloc][N] index of block (n, m) in the tables, or null
    const double* Jb;                 // [nloc]
    const double* hb;                 // [nloc]
    int nloc, N, B;
};

__global__ __launch_bounds__(256) void assemble_post_kernel(PostArgs g) {
    const int n = blockIdx.y;
    const int D = g.N * g.B;
    double* J = g.J + (long)n * g.strideJ;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c <= D) {
        const double xs = g.bo[(long)n * g.ldb + c];
        const double hk = g.bk[(long)n * g.ldb + c];
        if (c < D) {
            J[(long)D * g.ldj + c] = xs;
            const int m = c / g.B, bi = c % g.B;
            const long blk = g.label ? (long)g.label[(long)n * g.N + m] : (long)n * g.N + m;
            J[(long)(D + 1) * g.ldj + c] = hk + g.hw[blk * g.B + bi];
            const double* jw = g.Jw + blk * g.B * g.B;
            // lower-triangular part of the diagonal block (row r = m*B+bi, columns m*B .. r)
            for (int bj = 0; bj <= bi; ++bj) J[(long)c * g.ldj + m * g.B + bj] += jw[bi * g.B + bj];
        } else {
            J[(long)D * g.ldj + D] = xs + g.Jb[n];
            J[(long)(D + 1) * g.ldj + D] = hk + g.hb[n];
            J[(long)(D + 1) * g.ldj + D + 1] = 0.0;
        }
    }
}

// ------------------------------------------------------------------ Gaussian observations: J_lkhd[n] = (1/eta_n) X'X
// omega is the constant 1/eta_n (regression.py:421-423), so the Gram X'X is formed once per dataset and every sweep only
// scales it per neuron.  Each thread reads one pair of G0 and streams it to the nb neuron slots (coalesced 16-B stores);
// the lower triangle is what the later stages read, the pass covers whole rows up to the diagonal pair.
struct ScaleArgs {
    const double* G0; long ldg;
    const double* inv_eta;
    double* J; long ldj; long strideJ;
    int D, nb;
};

__global__ __launch_bounds__(256) void scaled_gram_kernel(ScaleArgs g) {
    const int r = blockIdx.y;
    const int c = 2 * (blockIdx.x * 256 + threadIdx.x);
    if (c > r || c >= g.D) return;
    const double2 v = *reinterpret_cast<const double2*>(g.G0 + (long)r * g.ldg + c);
    double* dst = g.J + (long)r * g.ldj + c;
    for (int n = 0; n < g.nb; ++n) {
        const double ie = g.inv_eta[n];
        *reinterpret_cast<double2*>(dst + (long)n * g.strideJ) = make_double2(v.x * ie, v.y * ie);
    }
}

}  // namespace

// ------------------------------------------------------------------ host launchers (called from pgl_api.hip)
int pgl_k_philox_words(uint64_t seed, uint32_t purpose, uint32_t j, uint64_t elem0, uint64_t stream, uint32_t* out, size_t n, hipStream_t st) {
    if (n == 0) return PGL_OK;
    hipLaunchKernelGGL(philox_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, purpose, j, elem0, stream, out, n);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_pg_draw(const double* b, const double* z, double* out, size_t len, uint64_t seed, uint64_t stream, uint64_t elem0, hipStream_t st) {
    if (len == 0) return PGL_OK;
    hipLaunchKernelGGL(pg_draw_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, b, z, out, len, seed, stream, elem0);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

// ------------------------------------------------------------------ sufficient statistics of a shard's rows for the network prior
// pyglm/networks.py:132-149 hands W[A & ~eye] (all active off-diagonal weight vectors of the population) to an NIW Gaussian, whose update needs
// only their count, sum and sum of outer products.  Per postsynaptic neuron n (one workgroup): out[n] = [count, sum_m w_m (B), sum_m w_m w_m' (B x B)]
// over the ACTIVE presynaptic m != n0 + n -- 1 + B + B^2 doubles that ride along in the row a rank contributes to the per-sweep all_gather, so
// that no rank walks the N^2 B doubles of the gathered state.  Every entry is accumulated over m = 0, 1, 2, ... by ONE thread: the same
// additions in the same order whatever the shard.
constexpr int RS_CHUNK = 64;
__global__ __launch_bounds__(256) void row_stats_kernel(const int* __restrict__ a, const double* __restrict__ W, double* __restrict__ out, int N, int B, int n0) {
    const int n = blockIdx.x, tid = threadIdx.x, self = n0 + n;
    const int ne = 1 + B + B * B;
    __shared__ double w[RS_CHUNK * 32];
    __shared__ int act[RS_CHUNK];
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};             // entries tid, tid + 256, ... (ne <= 1 + 32 + 1024)
    const double* Wn = W + (long)n * N * B;
    for (int m0 = 0; m0 < N; m0 += RS_CHUNK) {
        const int cnt = RS_CHUNK < N - m0 ? RS_CHUNK : N - m0;
        for (int i = tid; i < cnt * B; i += 256) w[i] = Wn[(long)m0 * B + i];
        if (tid < cnt) act[tid] = a[(long)n * N + m0 + tid] != 0 && (m0 + tid) != self;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int e = tid + 256 * k;
            if (e >= ne) break;
            const int i1 = e <= B ? e - 1 : (e - 1 - B) / B, i2 = e <= B ? -1 : (e - 1 - B) % B;
            double s = acc[k];
            for (int i = 0; i < cnt; ++i) {
                if (!act[i]) continue;
                s += e == 0 ? 1.0 : i2 < 0 ? w[i * B + i1] : w[i * B + i1] * w[i * B + i2];
            }
            acc[k] = s;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int e = tid + 256 * k;
        if (e < ne) out[(long)n * ne + e] = acc[k];
    }
}

int pgl_k_row_stats(const int* a, const double* W, double* out, int N, int B, int nloc, int n0, hipStream_t st) {
    hipLaunchKernelGGL(row_stats_kernel, dim3(nloc), dim3(256), 0, st, a, W, out, N, B, n0);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_pg_loglik(double* Psi, long ldpsi, const double* bias, const double* Y, long ldy, double* Omega, long ldo, double* Kappa, long ldk,
                    double* llpart, double* ll_out, int accumulate, int T, int nloc, int obs, double xi, uint64_t seed, uint64_t sweep,
                    uint64_t neuron0, uint64_t elem0, hipStream_t st) {
    PgLlArgs a{Psi, ldpsi, bias, Y, ldy, Omega, ldo, Kappa, ldk, llpart, T, nloc, obs, xi, nullptr, seed, sweep, neuron0, elem0};
    const int nblk = (T + PGLL_ROWS - 1) / PGLL_ROWS;
    if (nloc < 64 && obs != 2) hipLaunchKernelGGL(pg_loglik_narrow_kernel, dim3(nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pg_loglik_kernel, dim3(nblk, (nloc + 63) / 64), dim3(256), 0, st, a);
    PGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((nloc + 255) / 256), dim3(256), 0, st, llpart, nblk, nloc, ll_out, accumulate);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_gaussian_stats(double* Psi, long ldpsi, const double* bias, const double* Y, long ldy, const double* inv_eta, double* Omega, long ldo,
                         double* Kappa, long ldk, double* part, double* sse_out, int accumulate, int T, int nloc, hipStream_t st) {
    PgLlArgs a{Psi, ldpsi, bias, Y, ldy, Omega, ldo, Kappa, ldk, part, T, nloc, 2, 1.0, inv_eta, 0, 0, 0, 0};
    const int nblk = (T + PGLL_ROWS - 1) / PGLL_ROWS;
    hipLaunchKernelGGL(pg_loglik_kernel, dim3(nblk, (nloc + 63) / 64), dim3(256), 0, st, a);
    PGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((nloc + 255) / 256), dim3(256), 0, st, part, nblk, nloc, sse_out, accumulate);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_scaled_gram(const double* G0, long ldg, const double* inv_eta, double* J, long ldj, long strideJ, int D, int nb, hipStream_t st) {
    ScaleArgs a{G0, ldg, inv_eta, J, ldj, strideJ, D, nb};
    hipLaunchKernelGGL(scaled_gram_kernel, dim3((D / 2 + 255) / 256 + 1, D), dim3(256), 0, st, a);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_pg_loglik_nblk(int T) { return (T + PGLL_ROWS - 1) / PGLL_ROWS; }

int pgl_k_basis_conv(const double* S, long lds, const double* basis, double* X, long ldx, double* Xt, long ldxt, int T, int N, int B, int R,
                     int clip, hipStream_t st) {
    if (B > 32) { pgl_set_error("basis_conv: B=%d > 32 unsupported", B); return PGL_ERR_ARG; }
    ConvArgs a{S, lds, basis, X, ldx, Xt, ldxt, T, N, B, R, clip};
    const long total = (long)T * N;
    hipLaunchKernelGGL(basis_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), (size_t)R * B * sizeof(double), st, a);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_transpose(const double* src, long lds_, double* dst, long ldd, int rows, int cols, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_assemble_post(double* J, long ldj, long strideJ, const double* bo, const double* bk, long ldb, const double* Jw, const double* hw,
                        const int* label, const double* Jb, const double* hb, int nloc, int N, int B, hipStream_t st) {
    PostArgs a{J, ldj, strideJ, bo, bk, ldb, Jw, hw, label, Jb, hb, nloc, N, B};
    const int D = N * B;
    hipLaunchKernelGGL(assemble_post_kernel, dim3((D + 1 + 255) / 256, nloc), dim3(256), 0, st, a);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}
