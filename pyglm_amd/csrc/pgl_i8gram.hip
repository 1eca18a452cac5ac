// The omega-weighted Gram  J_n = X' diag(omega_n) X  by exact integer arithmetic on the int8 MFMA (the engine's choice at large shapes,
// the fp64 kernel of pgl_gemm.hip otherwise).  The fp64 operands are scaled COLUMN BY COLUMN to integers,
//     A[t][i] = round(x_ti sA_i),     B_n[t][j] = rint((omega_nt x_tj) sB_nj),
// with power-of-two scales chosen from each column's Euclidean norm and largest element (i8_colstats_kernel, i8_scales_kernel):
//     |A_i|_2, |B_nj|_2 in (limit(K, T) / 2, limit(K, T)]   unless the largest element would reach 2^50 (then that bound decides),
// so that by Cauchy-Schwarz every entry of the integer Gram S = A'B_n obeys |S_ij| <= |A_i||B_nj| < prod(p)/2 for the K <= 15 pairwise
// coprime moduli p <= 256 in use (limit = sqrt(prod(p)/2) less the rounding slack: 2^46.9 / 2^50.8 / 2^54.6 / 2^58.4 for K = 12..15).  S is computed modulo each p -- one int8 GEMM per
// modulus on residues that fit a signed byte, int32 accumulation (re-reduced mod p every 128 000 time bins) -- and reconstructed exactly by
// the Chinese remainder theorem;  J_ij = S_ij / (sA_i sB_nj).  The only approximation is the rounding of the operands to integers: with
// independent roundings the error of J_ij has standard deviation sqrt((|A_i|^-2 + |B_nj|^-2) / 12) |a_i||b_nj|, i.e. 2.1e-16 .. 4.2e-16
// |a_i||b_j| at K = 13 (a column whose norm is one outlier element is held at 2^49..2^50 by the element bound at every K) -- for ANY
// dynamic range: the precision is pinned to the column norms, not to the column maxima.  Roundings of REPEATED values are not independent
// (a design matrix of filtered spikes takes few distinct values per column, and every occurrence of a value rounds the same way: with
// plain round-to-nearest the error was ~5x that model on BASELINE configs[2]'s data and 1e-14 |a_i||b_j| at the worst entry on configs[4]'s,
// whose narrow first basis function leaves ~1000 distinct values in a column of 200 000).  The columns of X -- the operand that repeats;
// omega_nt x_tj does not -- are therefore rounded with a DITHER: round(v) = floor(v) + [frac(v) + u(t, i) >= 1], u a hash of the time bin and
// the column in [0, 1) (pgl_i8_dither): unbiased, independent from bin to bin, exact for integers (frac = 0 never rounds up), the same
// integers whatever the slicing or the launch geometry.  Emulated in NumPy on configs[4]-shaped data: worst entry 1.2e-14 -> 1.2e-15, rms
// 4.1e-15 -> 3.1e-16; configs[2]-shaped: 2.2e-15 -> 1.3e-15, 6.0e-16 -> 4.2e-16 (tests/test_gpu_i8gram.py).
//
//   i8_colstats_kernel max_t |v| and sum_t v^2 per column of X (once per data set) and of omega_g X (per neuron and sweep), deterministic
//   i8_scales_kernel   the scale of every column from those statistics
//   i8_planes_t_kernel fp64 (the TRANSPOSED design matrix: a column's bins are contiguous) -> K residue planes in BLOCKED layout
//                      [row / 16][K tile of 64 bins][row % 16][64 B]: PA for X (once per data set, or per time slice where they cannot stay
//                      resident), PB[g] for omega_g X (per neuron, per sweep); a lane keeps its 16 bins in registers for all neurons of a
//                      group; a residue is three fp64 instructions (no integer division); i8_planes_kernel: the same from the t-major X
//   i8_gram_kernel     R[g][q] = (PA[q] PB[g][q]') mod p_q on lower 320 x 320 tiles (v_mfma_i32_16x16x64_i8; 4 waves = 2 x 2, ONE wave per
//                      SIMD, wave tile 160 x 160 = 100 accumulators of 16 x 16 split over AGPRs and VGPRs by inline-asm register classes;
//                      K tiles DMA-staged into 3 LDS stages of 40 KiB, one contiguous KiB per request, 16-byte chunks XOR-swizzled:
//                      conflict-free ds_read_b128); persistent workgroups, per-XCD work lists in a clustered tile order (L2 sharing);
//                      int32 sums re-reduced every 128 000 bins; time slices of a data set add up in R (accumulate).  The 256 x 256-tile
//                      kernel of round 1 (two waves per SIMD) is kept behind PGL_I8_TILE=256
//   i8_crt_kernel      K residues -> mixed-radix digits (Garner) -> fp64 by Horner -> unscaled (exponent arithmetic) into the lower triangle of J
#include "pgl_common.h"
#include <cmath>
#include <type_traits>

namespace {

constexpr int NP = 15;
// compile-time table: every use below sits in a fully unrolled loop, so reductions mod p become multiply-shift sequences.
// 256 comes first: its residue is the low byte, and as the least significant mixed-radix digit its two representations of 128 are harmless
struct ModTable {
    int p[NP] = {256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197};
    int w[NP][NP] = {};               // w[k][q] = (p_0 ... p_{k-1}) mod p_q, symmetric representative (k < q)
    int pinv[NP] = {};                // pinv[q] = (p_0 ... p_{q-1})^-1 mod p_q
    constexpr ModTable() {
        for (int q = 1; q < NP; ++q) {
            int prod = 1;
            for (int k = 0; k < q; ++k) {
                w[k][q] = prod > p[q] / 2 ? prod - p[q] : prod;
                prod = (prod * (p[k] % p[q])) % p[q];
            }
            int x = 1;
            while ((prod * x) % p[q] != 1) ++x;
            pinv[q] = x;
        }
    }
};
constexpr ModTable MT{};
__constant__ int c_mod[NP] = {256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197};   // run-time indexed (Gram epilogue)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int ELEM_BITS = 50;          // |scaled element| < 2^50: the residue trick below needs |v| + 1.5 2^52 to stay below 2^53

// ------------------------------------------------------------------ column statistics of V = X (Om == null) or omega_g X
// amax[g][c] = max_t |v_tc|, ss[g][c] = sum_t v_tc^2 for the G <= 8 weight columns of a group in ONE pass over X.  A workgroup owns 16
// columns for all T (16 x 64 threads: column = tid % 16, time lane = tid / 16), so the sums are formed in a fixed order -- no atomics:
// the scales, and with them every bit of J, do not depend on launch timing.  A NaN / inf in a column makes its sum of squares
// non-finite; i8_scales_kernel turns that into a NaN scale (the Gram of that column is then NaN, as on the fp64 kernel).
constexpr int CS_COLS = 16, CS_LANES = 64, CS_G = 8, CS_ROWS = 4;     // a pass stages CS_LANES * CS_ROWS = 256 time bins of omega
template <int G, bool WEIGHTED>
__global__ __launch_bounds__(CS_COLS * CS_LANES) void i8_colstats_kernel(const double* __restrict__ X, long ldx, const double* __restrict__ Om,
                                                                        long ldo, int T, int D, double* __restrict__ amax,
                                                                        double* __restrict__ ss) {
    constexpr int CH = CS_LANES * CS_ROWS;
    __shared__ double red[2][CS_LANES][CS_COLS + 1];
    __shared__ double oms[WEIGHTED ? CH : 1][WEIGHTED ? G : 1];
    const int tid = threadIdx.x, cl = tid % CS_COLS, tl = tid / CS_COLS, c = blockIdx.x * CS_COLS + cl;
    const bool live = c < D;
    // gridDim.y > 1: the time axis in gridDim.y chunks (multiples of CH bins), chunk y's statistics at amax / ss + y G D (folded in chunk order
    // by i8_scales_kernel): 320 column blocks alone are one and a quarter workgroups per CU, each walking all of T -- 2.8 TB/s
    const int tchunk = ((T + (int)gridDim.y - 1) / (int)gridDim.y + CH - 1) / CH * CH;
    const int tbeg = (int)blockIdx.y * tchunk;
    T = T < tbeg + tchunk ? T : tbeg + tchunk;
    amax += (long)blockIdx.y * G * D;
    ss += (long)blockIdx.y * G * D;
    double m[G], q[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { m[g] = 0.0; q[g] = 0.0; }
    for (int t0 = tbeg; t0 < T; t0 += CH) {
        if (WEIGHTED) {
            __syncthreads();
            for (int e = tid; e < CH * G; e += CS_COLS * CS_LANES) {
                const int t = t0 + e / G;
                oms[e / G][e % G] = t < T ? Om[(long)t * ldo + e % G] : 0.0;
            }
            __syncthreads();
        }
        double x[CS_ROWS];
#pragma unroll
        for (int r = 0; r < CS_ROWS; ++r) {              // CS_ROWS independent loads in flight per thread
            const int t = t0 + tl + r * CS_LANES;
            x[r] = (live && t < T) ? X[(long)t * ldx + c] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < CS_ROWS; ++r)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double v = WEIGHTED ? x[r] * oms[tl + r * CS_LANES][g] : x[r];       // the same product the planes kernel rounds
                m[g] = fmax(m[g], fabs(v));               // (a NaN is dropped here and caught through the sum of squares)
                q[g] = fma(v, v, q[g]);
            }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        __syncthreads();
        red[0][tl][cl] = m[g];
        red[1][tl][cl] = q[g];
        __syncthreads();
        if (tl == 0 && live) {                   // one thread per column folds the 64 time lanes in order
            double mm = 0.0, qq = 0.0;
            for (int k = 0; k < CS_LANES; ++k) {
                mm = fmax(mm, red[0][k][cl]);
                qq += red[1][k][cl];
            }
            amax[(long)g * D + c] = mm;
            ss[(long)g * D + c] = qq;
        }
    }
}

// scale[k] = the largest POWER OF TWO with  scale |v|_2 <= limit  and  scale max|v| < 2^ELEM_BITS: the integer column gets a norm in
// (limit/2, limit] (`limit` = pgl_k_i8_norm_limit) unless its largest element would reach 2^50.  Powers of two on purpose: the scaling is
// then exact, and so is the whole product for data with few significant bits -- spike counts, an identity basis -- whereas an arbitrary scale
// rounds every occurrence of a repeated value the same way and those errors add up coherently (measured: 5x the random-rounding model
// on basis-filtered spikes, up to sqrt(T) x on binary columns).  1 for an empty column, NaN for a non-finite one.
__global__ __launch_bounds__(256) void i8_scales_kernel(const double* __restrict__ amax, const double* __restrict__ ss, long n, double limit,
                                                        double* __restrict__ scale, int nch) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double a = amax[k], sq = ss[k];
    for (int y = 1; y < nch; ++y) { a = fmax(a, amax[y * n + k]); sq += ss[y * n + k]; }      // time chunks of the statistics pass, in order
    const double nrm = sqrt(sq) * (1.0 + 1e-12);                        // (summation error of ss: never let it loosen the bound)
    double s = 1.0;
    if (!(a < HUGE_VAL) || !(nrm < HUGE_VAL)) s = __builtin_nan("");
    else if (a > 0.0) {
        const double nn = nrm > a ? nrm : a;
        int ea, en, el;
        (void)frexp(a, &ea);                                             // a = m 2^ea, m in [0.5, 1): a 2^(ELEM_BITS - ea) < 2^ELEM_BITS
        const double mn = frexp(nn, &en), ml = frexp(limit, &el);        // nn 2^e <= limit  <=>  mn 2^(en + e) <= ml 2^el
        const int e_norm = el - en - (mn > ml ? 1 : 0);
        const int e = min(ELEM_BITS - ea, e_norm);
        s = ldexp(1.0, e);
    }
    scale[k] = s;
}

// ---- the same scales for a whole batch of neurons without a pass over X per group (pgl_sweep.hip): the sums of squares of the columns of
// omega_n X come from ONE fp64 MFMA contraction per batch, ss[n][d] = sum_t omega_nt^2 x_td^2 (PGL_GEMM_SQUARES), and the largest element is
// replaced by the bound max_t omega_nt * max_t |x_td| (and by the norm itself, whichever is smaller).  Both only ever make a scale SMALLER than
// the exact statistics would, so every range argument above holds; the element bound matters only for a column that one element dominates
// (|v|_max > 0.59 |v|_2 with 13 moduli).  The products of squares are rounded differently from the squares of the rounded products the planes
// kernel forms and the MFMA sums them sequentially: relative (T + 16) 2^-53 at most, which the norm is inflated by before it is used.
__global__ __launch_bounds__(256) void i8_colmax_kernel(const double* __restrict__ Om, long ldo, int T, int n, unsigned long long* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const int per = (T + (int)gridDim.y - 1) / (int)gridDim.y, t0 = (int)blockIdx.y * per, t1 = min(T, t0 + per);
    unsigned long long m = 0ull;
    for (int t = t0; t < t1; ++t) {
        const unsigned long long v = (unsigned long long)__double_as_longlong(Om[(long)t * ldo + c]) & 0x7fffffffffffffffull;   // |.|: ordered like the integers
        m = v > m ? v : m;                                                                                                  // (a NaN sorts above every number)
    }
    atomicMax(&out[c], m);               // a maximum does not depend on the order it is taken in
}

__global__ __launch_bounds__(256) void i8_scales_bound_kernel(const double* __restrict__ ss, long ldss, const double* __restrict__ ommax,
                                                              const double* __restrict__ xmax, int D, int G, double limit, double inflate,
                                                              double* __restrict__ scale) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= (long)G * D) return;
    const int g = (int)(k / D), d = (int)(k % D);
    const double sq = ss[(long)g * ldss + d];
    const double nrm = sqrt(sq) * inflate;
    const double bound = ommax[g] * xmax[d];
    const double a = bound < nrm ? bound : nrm;          // (a NaN / inf anywhere in the column shows in sq)
    double s = 1.0;
    if (!(nrm < HUGE_VAL)) s = __builtin_nan("");
    else if (a > 0.0) {
        int ea, en, el;
        (void)frexp(a, &ea);
        const double mn = frexp(nrm, &en), ml = frexp(limit, &el);
        const int e_norm = el - en - (mn > ml ? 1 : 0);
        const int e = min(ELEM_BITS - ea, e_norm);
        s = ldexp(1.0, e);
    }
    scale[k] = s;
}

// ------------------------------------------------------------------ fp64 -> residue planes
struct PlaneArgs {
    const double* X; long ldx;            // [T][ldx], or (transposed != 0) the transposed copy [D][ldx]
    int transposed;
    const double* Om; long ldo;           // [T][ldo] weights of the group's neurons (null: unweighted, one "neuron")
    const double* scale;                  // [G][D] fixed-point scale of column d of neuron g (i8_scales_kernel)
    int8_t* P;                            // [G][np] planes of Dq * Kp bytes, blocked [Dq / 16][Kp / 64][16][64]
    int T, D, Dq; long Kp; int np;
    unsigned t_base;                      // global index of time bin 0 of this call (a time slice of a data set): keys the dither of the X planes
};

// u(t, d) in [0, 1): the dither of element (time bin t, column d) of X -- two rounds of a 32-bit integer mixer ("lowbias32").  Specified here and
// restated in NumPy by tests/test_gpu_i8gram.py; keyed by the GLOBAL time bin, so a data set converted in slices gets the same integers.
__device__ __forceinline__ unsigned pgl_mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ double pgl_i8_dither(unsigned t, unsigned d) { return (double)pgl_mix32(t + pgl_mix32(d + 0x9e3779b9u)) * (1.0 / 4294967296.0); }
// round(y) with dither u: floor(y) + [frac(y) + u >= 1]  (y = x * scale is exact: the scale is a power of two)
__device__ __forceinline__ double pgl_round_dither(double y, double u) {
    const double fl = floor(y);
    return fl + (((y - fl) + u >= 1.0) ? 1.0 : 0.0);
}

// A workgroup converts 256 time bins x 16 columns (one row block, four K tiles) of X, staged ONCE in LDS, for all G neurons of the group.
// Wave w owns K tile w: lane l -> column l / 4, 16 consecutive time bins (one 16-byte chunk of the 64-byte row), so a wave stores one
// contiguous KiB -- exactly one request of the Gram kernel -- per plane.  Residue of the integer v (|v| < 2^50, carried as fp64) mod p:
//     q = rint(v / p) by the add-and-subtract of M = 1.5 2^52,     r = v - p q  (one fma, exact),
// |r| <= p/2 + 0.4 < 128: a valid signed-byte representative (not always the smallest one; the GEMM and the CRT only need congruence).
// The fma is taken on v + M, so the byte is the low byte of the result's mantissa: no conversion instruction, no integer division.
constexpr int PT_D = 16;
constexpr double MAGIC = 6755399441055744.0;      // 1.5 * 2^52
template <int PT_T>                                // time bins (= threads) per workgroup: one K tile of 64 bins per wave
__global__ __launch_bounds__(PT_T) void i8_planes_kernel(PlaneArgs a, int G) {
    __shared__ double tile[PT_D][PT_T + 1];
    __shared__ double oms[PT_T];
    const int t0 = blockIdx.x * PT_T, d0 = blockIdx.y * PT_D;
    const int tid = threadIdx.x;
    if (a.transposed) {                                    // rows of the transposed copy: one coalesced run of PT_T doubles per column
        const int t = t0 + tid;
#pragma unroll
        for (int dl = 0; dl < PT_D; ++dl) tile[dl][tid] = (t < a.T && d0 + dl < a.D) ? a.X[(long)(d0 + dl) * a.ldx + t] : 0.0;
    } else {                                               // 128-byte pieces of PT_T different rows
        const int dl = tid & (PT_D - 1), d = d0 + dl;
        for (int tl = tid / PT_D; tl < PT_T; tl += PT_T / PT_D) {
            const int t = t0 + tl;
            tile[dl][tl] = (t < a.T && d < a.D) ? a.X[(long)t * a.ldx + d] : 0.0;
        }
    }
    const int w = tid >> 6, l = tid & 63, r = l >> 2, tb = 64 * w + 16 * (l & 3);
    const long kt = t0 / 64 + w;
    const bool live = kt * 64 < a.Kp;
    const int d = d0 + r;
    const long nkt = a.Kp / 64;
    int8_t* const dst0 = a.P + (((long)blockIdx.y * nkt + kt) << 10) + l * 16;
    const long plane = (long)a.Dq * a.Kp;
    const bool wrow = a.Om && t0 + tid < a.T;
    const double* const omrow = a.Om ? a.Om + (long)(t0 + tid) * a.ldo : nullptr;
    double om_next = wrow ? omrow[0] : 0.0;
    for (int gz = 0; gz < G; ++gz) {
        __syncthreads();                                   // the tile is staged / the previous neuron's weights are no longer read
        if (a.Om) oms[tid] = om_next;
        if (gz + 1 < G) om_next = wrow ? omrow[gz + 1] : 0.0;         // the next neuron's weight is in flight while this one is converted
        __syncthreads();
        if (!live) continue;
        const double scale = d < a.D ? a.scale[(long)gz * a.D + d] : 0.0;
        double v[16], vm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double x = tile[r][tb + k];
            v[k] = a.Om ? rint((x * oms[tb + k]) * scale)           // x * omega rounded to fp64 first, as X*omega[:,None] is
                        : pgl_round_dither(x * scale, pgl_i8_dither(a.t_base + (unsigned)(t0 + tb + k), (unsigned)d));     // the planes of X: dithered
            vm[k] = v[k] + MAGIC;                                   // exact: integers below 2^53
        }
        int8_t* dst = dst0 + (long)gz * a.np * plane;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (q < a.np) {                                  // (uniform: the launch's number of moduli)
                unsigned b[16];
                if (q == 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) b[k] = (unsigned)__double2loint(vm[k]);      // mod 256: the low byte of the integer itself
                } else {
                    const double pd = (double)MT.p[q], ip = 1.0 / pd;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        // three instructions per residue.  (Left to the compiler, the second fma becomes a two-address v_fmac_f64 plus a
                        // v_mov_b64 that copies v + M into its destination every time: a fourth.)
                        const double qq = fma(v[k], ip, MAGIC) - MAGIC;
                        double rr;
                        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(rr) : "s"(-pd), "v"(qq), "v"(vm[k]));
                        b[k] = (unsigned)__double2loint(rr);
                    }
                }
                v4i out;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned lo = __builtin_amdgcn_perm(b[4 * j + 1], b[4 * j], 0x0c0c0400u);
                    const unsigned hi = __builtin_amdgcn_perm(b[4 * j + 3], b[4 * j + 2], 0x0c0c0400u);
                    out[j] = (int)__builtin_amdgcn_perm(hi, lo, 0x05040100u);
                }
                // non-temporal: the planes are read next by another kernel, from HBM either way (53 GB per group); keeping them out of the
                // write-back caches measured 5 % faster (12.95-13.2 vs 13.5-14.2 ms per group on one box)
                __builtin_nontemporal_store(out, reinterpret_cast<v4i*>(dst + (long)q * plane));
            }
        }
    }
}

// The same conversion from the TRANSPOSED design matrix Xt [D][ldx], without the LDS tile: a lane's 16 consecutive time bins of its row are
// 128 contiguous bytes of Xt, so it loads them straight into registers (a wave: 16 rows x 512 B) and keeps them for all G neurons.  The
// weights of ALL G neurons for the workgroup's bins are staged in LDS once (64 contiguous bytes per bin in Om), so a tile has one barrier
// instead of two per neuron, and the waves of a workgroup run independently of each other afterwards: the loads of one overlap the stores
// of another.  (With the tile in LDS the pass over X cost 1.2-2 ms per call on top of ~1.2 ms per neuron of stores, unoverlapped.)
template <int PT_T>
__global__ __launch_bounds__(PT_T) void i8_planes_t_kernel(PlaneArgs a, int G) {
    __shared__ double oms[CS_G][PT_T + 2];
    const int t0 = blockIdx.x * PT_T, d0 = blockIdx.y * PT_D;
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, r = l >> 2, tb = 64 * w + 16 * (l & 3);
    const long kt = t0 / 64 + w;
    const bool live = kt * 64 < a.Kp;
    const int d = d0 + r;
    double x[16];
    {
        const int t = t0 + tb;
        const double* src = a.X + (long)d * a.ldx + t;
        if (live && d < a.D && t + 16 <= a.T) {                      // (rows of Xt are 16-byte aligned: ldx even, t a multiple of 16)
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                const double2 v2 = *reinterpret_cast<const double2*>(src + k);
                x[k] = v2.x; x[k + 1] = v2.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = (live && d < a.D && t + k < a.T) ? src[k] : 0.0;
        }
    }
    if (a.Om) {
        const int t = t0 + tid;
        const double* src = a.Om + (long)t * a.ldo;
        for (int g = 0; g < G; ++g) oms[g][tid] = t < a.T ? src[g] : 0.0;
        __syncthreads();
    }
    if (!live) return;
    const long nkt = a.Kp / 64;
    int8_t* const dst0 = a.P + (((long)blockIdx.y * nkt + kt) << 10) + l * 16;
    const long plane = (long)a.Dq * a.Kp;
    for (int gz = 0; gz < G; ++gz) {
        const double scale = d < a.D ? a.scale[(long)gz * a.D + d] : 0.0;
        double v[16], vm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v[k] = a.Om ? rint((x[k] * oms[gz][tb + k]) * scale)              // x * omega rounded to fp64 first, as X*omega[:,None] is
                        : pgl_round_dither(x[k] * scale, pgl_i8_dither(a.t_base + (unsigned)(t0 + tb + k), (unsigned)d));
            vm[k] = v[k] + MAGIC;
        }
        int8_t* dst = dst0 + (long)gz * a.np * plane;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (q < a.np) {
                unsigned b[16];
                if (q == 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) b[k] = (unsigned)__double2loint(vm[k]);
                } else {
                    const double pd = (double)MT.p[q], ip = 1.0 / pd;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const double qq = fma(v[k], ip, MAGIC) - MAGIC;
                        double rr;
                        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(rr) : "s"(-pd), "v"(qq), "v"(vm[k]));
                        b[k] = (unsigned)__double2loint(rr);
                    }
                }
                v4i out;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned lo = __builtin_amdgcn_perm(b[4 * j + 1], b[4 * j], 0x0c0c0400u);
                    const unsigned hi = __builtin_amdgcn_perm(b[4 * j + 3], b[4 * j + 2], 0x0c0c0400u);
                    out[j] = (int)__builtin_amdgcn_perm(hi, lo, 0x05040100u);
                }
                __builtin_nontemporal_store(out, reinterpret_cast<v4i*>(dst + (long)q * plane));
            }
        }
    }
}

// ------------------------------------------------------------------ int8 Gram of the planes, reduced mod p
constexpr int BKB = 64;                                      // bytes (= time bins) per row of a K tile
constexpr int KCH = 2000;                                    // K tiles between re-reductions: 128 + 128000 * 128 * 128 < 2^31
constexpr int CH = 32;                               // flat work list (ragged groups): chunk per XCD
constexpr int SB = 6;                                // clustered tile order of the flat list: super-blocks of SB x SB tiles (measured on one box, ms per
                                                     // launch at cfg3: 4: 101.4 / 100.1, 5: 101.1, 6: 102.0 / 100.1, 8: 105.2, 16: 103.7)

struct GramArgs {
    const int8_t* PA;                     // [np] planes
    const int8_t* PB;                     // [G][np] planes
    int8_t* R;                            // [G][np][Dq][Dq]
    int Dq; long Kp; int G; int np;       // np = number of moduli in use (the first np of the table)
    int sbr, sbc;                         // super-blocks of the clustered tile order: tile rows x tile columns
    int nkt;                              // K tiles to multiply: the time bins rounded up to 64 (<= Kp / 64: very short slices are padded to 4 tiles of zeros)
    int kt0;                              // first K tile of this pass (passes of KCH tiles; later ones accumulate)
    int* sched;                           // 8 per-XCD work counters, zeroed before the launch
    long KpA; int ka0;                    // the X planes may be longer than this slice of the omega X planes: their bytes per row, first K tile
    int accum;                            // add to the residues already in R (a later time slice of the same product)
    int8_t* Rx;                           // split of the LAST plane into K quarters: quarters 1..3 of neuron g's residues
                                          // go to Rx[g][quarter - 1][Dq][Dq] (i8_crt adds the four); NULL: no split
#ifdef PGL_AB
#define PGL_KPARTS(g_) ((g_).kparts)
#else
#define PGL_KPARTS(g_) 0                  // the shipped kernel is compiled without the experiment (same code as before it existed)
#endif
    int kparts;                           // A/B experiment (-DPGL_AB builds, PGL_I8_KPARTS): > 0 cuts the items of EVERY plane into kparts pieces of K,
                                          // piece-major within a plane, the pieces of a tile adding up in place in R (see pgl_k_i8_gram); 0: off
};

__device__ __forceinline__ int isqrt_tri_i(int t) {
    int r = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)(r + 1) * (r + 2) / 2 <= t) ++r;
    while ((long)r * (r + 1) / 2 > t) --r;
    return r;
}

// tile t of a plane's lower triangle in CLUSTERED order: super-blocks of sbr x sbc tiles (tile rows = strips of the X planes, tile columns
// = strips of the neuron's omega X planes), block row by block row (boustrophedon), row-major inside a block and only the tiles on or below
// the diagonal -- so that a run of ~32 consecutive tiles touches few distinct 320-row strips: the workgroups of an XCD, which take such a
// run together, share the strips through their L2 (with the plain triangular order every workgroup streamed its own two strips: 2.2
// instead of 3.7 POPS)
__device__ __forceinline__ void clustered_tile(int t, int ntm, int sbr, int sbc, int& tm, int& tn) {
    const int nbr = (ntm + sbr - 1) / sbr;
    for (int I = 0; I < nbr; ++I) {
        const int r0 = I * sbr, r1 = min(ntm, r0 + sbr);              // tile rows [r0, r1)
        const int nbc = (r1 - 1) / sbc + 1;                            // column blocks that reach the diagonal of this block row
        for (int jj = 0; jj < nbc; ++jj) {
            const int J = (I & 1) ? nbc - 1 - jj : jj;
            const int c0 = J * sbc, c1 = min(ntm, c0 + sbc);
            for (int r = max(r0, c0); r < r1; ++r) {
                const int cnt = min(c1, r + 1) - c0;                   // tiles (r, c0 .. min(c1 - 1, r))
                if (t < cnt) { tm = r; tn = c0 + t; return; }
                t -= cnt;
            }
        }
    }
    tm = tn = 0;
}

__device__ __forceinline__ int chunk_swz(int qd) { return (0x1320 >> (4 * qd)) & 3; }     // g = {0, 2, 3, 1}

// ------------------------------------------------------------------ the same product on 320 x 320 tiles, ONE wave per SIMD
// Under the package power limit the rate is set by the energy per operation, and a third of the 256 x 256
// kernel's energy is operand movement.  This variant holds a 320 x 320 tile per workgroup: 4 waves = 2 x 2, wave tile 160 x 160 = 10 x 10
// accumulators of 16 x 16 -- 400 accumulator registers per lane, which only fit with one wave per SIMD and BOTH register files: the
// first 64 accumulators live in AGPRs, the other 36 in VGPRs.  hipcc will not split MFMA accumulators over the two files on its own
// (it picks the AGPR form for the whole function and shuttles the overflow with v_accvgpr moves), so the MFMAs are issued as inline
// assembly with register-class constraints ("+a" / "+v"); everything else -- LDS reads, DMA requests, addressing -- stays C++.
// Per K tile and wave: 100 MFMAs for 20 fragment reads (0.20 per MFMA instead of 0.375) and 10 DMA requests (L2 -> LDS bytes per
// operation -20 %).  Three LDS stages of 640 rows x 64 B, requests one tile ahead (a fourth stage is slower, see pgl_k_i8_gram); the B fragments are single-buffered (each is reloaded for the next tile one
// MFMA after its last use), the A fragments rotate through five register sets, read two rows ahead.  One barrier per K tile, after row 4:
// it publishes tile kt+1 (whose fragments are first read in rows 8 and 9) and frees the stage of tile kt-1 for the requests of tile kt+2.
constexpr int BT = 320, BROWS = 2 * BT;
constexpr int BSTAGE = BROWS * BKB;                          // 40 KiB per stage (the work ticket aliases stage 0 between items)

template <int IDX>
__device__ __forceinline__ void big_mfma(v4i& acc, const v4i& a, const v4i& b) {
    if constexpr (IDX < 64) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// TRI: the wave tile straddles the diagonal (a diagonal wave of a diagonal tile): the 16 x 16 blocks strictly above it (J > I) are
// never read, so their MFMAs are not issued (45 of 100); the slots keep their places
// ORDER OF ISSUE.  Step J of row I multiplies FA[I] by FB[JJ].  Row-major (JJ = J) changes the B operand on every instruction and BOTH operands
// at every row change; under the package power limit the multipliers' input switching is energy, i.e. time.  Odd rows (but the last) therefore
// run their columns backwards: the B fragment of a row's last MFMA is the first of the next row, and only the A operand changes there.  Rows 0
// and 9 keep the forward order, because the B fragments of the next K tile are reloaded behind row 9's MFMAs in the order row 0 wants them.
// (tools/ubench_i8_order.hip, register-only 8 x 4 blocks on random bytes: row-major 3.98 POP/s, boustrophedon 4.07, both operands changing
// on every instruction 3.84.)  Same products, same exact integer sums.
#ifndef PGL_I8_SNAKE
#define PGL_I8_SNAKE 1
#endif
template <bool TRI, int I, int J, typename F>
__device__ __forceinline__ void big_rows(v4i (&acc)[10][10], v4i (&FA)[5], v4i (&FB)[10], F&& slot) {
    if constexpr (I < 10) {
        constexpr int JJ = (PGL_I8_SNAKE && (I & 1) && I != 9) ? 9 - J : J;
        if constexpr (!TRI || JJ <= I) big_mfma<I * 10 + JJ>(acc[I][JJ], FA[I % 5], FB[JJ]);
        slot(std::integral_constant<int, I>{}, std::integral_constant<int, J>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (J == 9) big_rows<TRI, I + 1, 0>(acc, FA, FB, slot);
        else big_rows<TRI, I, J + 1>(acc, FA, FB, slot);
    }
}

// kpart < 0: the whole K range of this pass; 0..3: that quarter of it (split items of the last plane, see i8_gram_kernel)
constexpr int BNST = 3;                                      // LDS stages (requests one tile ahead; a fourth stage measured 6.7 % slower, see pgl_k_i8_gram)
__device__ __forceinline__ void i8_gram_item_big(const GramArgs& g, const int gz, const int q, const int tm, const int tn, char* lds, const int kpart = -1) {
    const int m0 = tm * BT, n0 = tn * BT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long plane = (long)g.Dq * g.Kp;
    const int8_t* A = g.PA + (long)q * g.Dq * g.KpA + (long)g.ka0 * 1024;
    const int8_t* B = g.PB + ((long)gz * g.np + q) * plane;
    const int nkt = g.nkt;
    // DMA: per K tile 640 rows x 64 B = 40 requests of 1 KiB; wave w issues requests w, w+4, ..., w+36 (row blocks 0..19 = A, 20..39 = B).
    // A request = wave-uniform base (scalar registers) + one per-lane byte offset shared by all requests
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ chunk_swz((lane >> 4) & 3)) * 16));
    const int fr = lane & 15, fc = lane >> 4;
    auto frag = [&](int stage, int row) {
        const int pc = fc ^ chunk_swz((row >> 2) & 3);
        return *reinterpret_cast<const v4i*>(lds + stage * BSTAGE + row * BKB + pc * 16);
    };
    auto rdA = [&](int stage, int f) { return frag(stage, wm * 160 + f * 16 + fr); };
    auto rdB = [&](int stage, int f) { return frag(stage, BT + wn * 160 + f * 16 + fr); };
    const double pq = (double)c_mod[q], ipq = 1.0 / pq;
    auto reduce = [&](int v) {                              // symmetric representative of v mod p (exact: |v| < 2^31)
        const double x = (double)v;
        return (int)fma(-pq, rint(x * ipq), x);
    };
    const bool inplace = PGL_KPARTS(g) > 0;                     // every plane in K pieces that add up in place (experiment)
    int8_t* R = (kpart > 0 && !inplace) ? g.Rx + ((long)gz * 3 + (kpart - 1)) * g.Dq * g.Dq : g.R + ((long)gz * g.np + q) * g.Dq * g.Dq;
    // long data sets: the host launches one pass per chunk of KCH K tiles (the int32 sums cannot overflow within one); a pass adds its
    // residues to the previous passes' through the output bytes.  (A chunk loop in here makes hipcc spill the 400 accumulators.)
    {
        int kc = g.kt0;
        int nk = min(nkt - kc, KCH);
        if (kpart >= 0) {                                  // a quarter of the pass (quarters tile it: the last may be short)
            const int parts = inplace ? PGL_KPARTS(g) : 4;
            const int quarter = (nk + parts - 1) / parts, kend = kc + nk;
            kc += kpart * quarter;
            nk = min(quarter, kend - kc);
            if (nk <= 0) {                                 // (only for passes of fewer than 4 K tiles: an empty quarter contributes zero)
                if (inplace && kpart > 0) return;
                if (g.kt0 == 0 && !g.accum)
                    for (int e = threadIdx.x; e < BT * BT; e += 256) R[(long)(m0 + e / BT) * g.Dq + n0 + e % BT] = 0;
                return;
            }
        }
        const char* sb[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int rq = wv + 4 * i;
            const int8_t* base = rq < 20 ? A + (long)(m0 + 16 * rq) * g.KpA : B + (long)(n0 + 16 * (rq - 20)) * g.Kp;
            sb[i] = reinterpret_cast<const char*>(base) + (long)kc * 1024;
        }
        long gadv = nk >= 2 ? 1024 : 0;   // 0 once the pass's last tile has been requested: the cursors stop (redundant re-requests into a dead stage)
        auto dma_piece = [&](int stage, int i) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(sb[i] + voff), (lds_ptr_t)(lds + stage * BSTAGE + (wv + 4 * i) * 1024), 16, 0, 0);
            sb[i] += gadv;
        };
        // prologue: tiles 0 .. BNST-2 of the pass
#pragma unroll
        for (int i = 0; i < 10; ++i) dma_piece(0, i);
        if (nk < 3) gadv = 0;
#pragma unroll
        for (int i = 0; i < 10; ++i) dma_piece(1, i);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // tile 0 has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // On a diagonal tile the wave that owns rows 0..159 x columns 160..319 lies wholly above the diagonal and nothing ever reads its
        // residues: it keeps its share of the requests and the barriers and issues no MFMA, no fragment read and no store (under the
        // power limit what an idle SIMD does not burn is clock for the other three: 3 % of the launch's MFMA energy)
        if (tm == tn && wv == 1) {
            int cur = 0;
            for (int kt = 0; kt < nk; ++kt) {
                const int dst = cur == 0 ? BNST - 1 : cur - 1;
                if (kt + BNST >= nk) gadv = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 10; ++i) dma_piece(dst, i);
                cur = cur == BNST - 1 ? 0 : cur + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            return;
        }
        v4i acc[10][10];
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int j = 0; j < 10; ++j) acc[i][j] = v4i{0, 0, 0, 0};
        v4i FA[5], FB[10];
#pragma unroll
        for (int f = 0; f < 10; ++f) FB[f] = rdB(0, f);
        FA[0] = rdA(0, 0);
        FA[1] = rdA(0, 1);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");       // accumulator initialisation (VALU writes) ahead of the first MFMA
        auto k_loop = [&](auto tri_c) {
        constexpr bool TRI = decltype(tri_c)::value;
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int nxt = cur == BNST - 1 ? 0 : cur + 1;
            const int dst = cur == 0 ? BNST - 1 : cur - 1;           // the stage of tile kt-1 takes tile kt + BNST - 1
            if (kt + BNST >= nk) gadv = 0;
            big_rows<TRI, 0, 0>(acc, FA, FB, [&](auto ic, auto jc) {
                constexpr int I = decltype(ic)::value, J = decltype(jc)::value;
                // A fragment of row I+2 (rows 10, 11 = rows 0, 1 of the next tile), two rows ahead; five register sets in rotation
                // (10 rows = 2 x 5: the rotation comes back to set 0 at every tile)
                if constexpr (J == 1) { if constexpr (I + 2 < 10) FA[(I + 2) % 5] = rdA(cur, I + 2); else FA[(I + 2) % 5] = rdA(nxt, I + 2 - 10); }
                // mid-tile: tile kt+1 has landed (own requests; the 10 of tile kt+2 may stay in flight), everybody is past tile kt-1
                if constexpr (I == 4 && J == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // requests of tile kt + BNST - 1: two per row in rows 4..8
                // (measured alternatives, same box, ms per launch: rows 5..9 106.7, all ten in rows 4 and 5 102.1, this 100.1-102.0)
                if constexpr (I >= 4 && I <= 8 && (J == 3 || J == 7)) dma_piece(dst, 2 * (I - 4) + (J == 7));
                // B fragments of the next tile: fragment J-1 one MFMA after its last use (row 9); fragment 9 below
                if constexpr (I == 9 && J >= 1) FB[J - 1] = rdB(nxt, J - 1);
            });
            FB[9] = rdB(nxt, 9);
            cur = nxt;
        }
        };
        if (tm == tn && wm == wn) k_loop(std::true_type{});
        else k_loop(std::false_type{});
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // last (redundant) requests landed; MFMA results readable
        __builtin_amdgcn_s_barrier();                      // every wave is done with the stages before the next item refills them
        auto store = [&](auto accum_c) {
            constexpr bool ACCUM = decltype(accum_c)::value;
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int j = 0; j < 10; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + wm * 160 + i * 16 + 4 * (lane >> 4) + r;
                        const int col = n0 + wn * 160 + j * 16 + (lane & 15);
                        int8_t* dst8 = R + (long)row * g.Dq + col;
                        int v = reduce(acc[i][j][r]);
                        if constexpr (ACCUM) v = reduce(v + (int)*dst8);
                        *dst8 = (int8_t)(v & 0xff);          // |.| <= p/2 <= 128; +128 (p = 256 only) wraps to -128, congruent
                    }
        };
        if (g.kt0 == 0 && !g.accum && !(inplace && kpart > 0)) store(std::false_type{});
        else store(std::true_type{});
    }
}

// persistent: one workgroup per CU pulls (neuron of the group, plane, tile in clustered order) items from per-XCD work lists; a workgroup
// reads the XCC it runs on and takes the next item of that XCD's list, stealing from the next XCD when its own list is exhausted.
//   G a multiple of 8 (every full group): XCD y owns NEURONS y, y + 8, ... -- its list is, plane by plane, those neurons' tiles in order.
//     With one neuron per XCD (G = 8: large D) the 32 workgroups of an XCD walk one column block of one plane together (two omega X strips
//     of their neuron, shared through their L2, against all the X strips below the diagonal), and the eight XCDs walk the SAME plane and
//     column block at the same time, each for its own neuron: the X-plane strips (half of every item's operand stream, identical for all
//     neurons) are fetched from HBM once and found in the memory-side cache by the other seven.  With several neurons per XCD (G = 16 .. 64: small D, where a plane has only a few tiles and 8
//     neurons would not fill the CUs for more than a round or two) the whole chip still walks one plane at a time.
//   otherwise (a ragged last group): the flat item list is cut into chunks of CH, chunk c belongs to XCD c % 8.
// Placement is used for speed only -- any placement gives the same result.
__global__ __launch_bounds__(256) void i8_gram_kernel(GramArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int* ticket = reinterpret_cast<int*>(lds);             // (aliases stage 0, which is idle between items)
    const int ntm = g.Dq / BT;
    const int ntiles = ntm * (ntm + 1) / 2;
    const int npx = (g.G % 8 == 0) ? g.G / 8 : 0;          // neurons per XCD list (0: flat chunked list)
    // Balance of the last rounds (per-XCD lists): a list is np x npx x ntiles equal items for 32 CUs -- at cfg3 13 x 136 = 55.25 rounds, and
    // the quarter round at the end leaves 24 of 32 CUs idle for a whole item (1.4 % of the launch).  The LAST plane's items are
    // therefore cut into four K quarters each (their residues land in four slots that i8_crt adds up: modular sums are exact in any
    // order), taken 32 tiles x one quarter at a time so that co-running items still share their strips: 12 x 136 = 51 x 32 full items and
    // 4 x 136 = 17 x 32 quarter items -- no partial round at cfg3, and never more than a quarter item of imbalance after the last full item.
    const bool split = npx > 0 && g.Rx != nullptr && PGL_KPARTS(g) == 0;
    const bool pieces = npx > 0 && PGL_KPARTS(g) > 0;
    const int nfull = pieces ? 0 : ntiles * npx * (split ? g.np - 1 : g.np);
    const int per_xcd = pieces ? g.np * PGL_KPARTS(g) * npx * ntiles : nfull + (split ? 4 * ntiles * npx : 0);
    const int total = ntiles * g.np * g.G;                 // (flat list)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    for (;;) {
        __syncthreads();                                   // also drains the previous item's last (redundant) requests: vmcnt(0)
        if (threadIdx.x == 0) {
            int w = -1, y = 0;
            for (int hop = 0; hop < 8 && w < 0; ++hop) {
                y = (int)((xcc + hop) & 7u);
                const int it = atomicAdd(&g.sched[y], 1);
                if (npx) { if (it < per_xcd) w = it; }
                else {
                    const long cand = ((long)(it / CH) * 8 + y) * CH + it % CH;
                    if (cand < total) w = (int)cand;
                }
            }
            if (w >= 0) {
                int tm, tn, gz, q, kpart = -1, tile;
                if (!npx) { const int pair = w / ntiles; tile = w % ntiles; gz = pair / g.np; q = pair % g.np; }
                else if (pieces) {                         // plane-major, then the piece of K, then the XCD's neurons, then the tiles
                    const int per_piece = npx * ntiles, per_plane = PGL_KPARTS(g) * per_piece, r = w % per_plane, r2 = r % per_piece;
                    q = w / per_plane; kpart = r / per_piece; gz = y + 8 * (r2 / ntiles); tile = r2 % ntiles;
                }
                else if (w < nfull) {                      // plane-major, then the XCD's neurons, then the tiles of a plane
                    const int per_plane = npx * ntiles, r = w % per_plane;
                    q = w / per_plane; gz = y + 8 * (r / ntiles); tile = r % ntiles;
                } else {                                   // per neuron: groups of 32 tiles (the last may be smaller), quarter-major inside a group
                    const int r0 = w - nfull, j = r0 / (4 * ntiles), r = r0 % (4 * ntiles), grp = r / 128, within = r - grp * 128;
                    const int tg = min(32, ntiles - grp * 32);
                    kpart = within / tg;
                    tile = grp * 32 + within % tg;
                    gz = y + 8 * j; q = g.np - 1;
                }
                clustered_tile(tile, ntm, g.sbr, g.sbc, tm, tn);
                ticket[1] = gz; ticket[2] = tm; ticket[3] = tn; ticket[4] = kpart; ticket[5] = q;
            }
            ticket[0] = w;
        }
        __syncthreads();
        if (ticket[0] < 0) break;
        const int gz = __builtin_amdgcn_readfirstlane(ticket[1]), q = __builtin_amdgcn_readfirstlane(ticket[5]);
        const int tm = __builtin_amdgcn_readfirstlane(ticket[2]), tn = __builtin_amdgcn_readfirstlane(ticket[3]);
        const int kpart = __builtin_amdgcn_readfirstlane(ticket[4]);
        __syncthreads();                                   // the ticket lives in stage 0: everybody has read it before the first request lands there
        i8_gram_item_big(g, gz, q, tm, tn, lds, kpart);
    }
}

// ------------------------------------------------------------------ CRT reconstruction into J
struct CrtArgs {
    const int8_t* R;                      // [G][np][Dq][Dq]
    const double* sA; const double* sB;   // scales: [D], [G][D]
    double* J; long ldj; long strideJ;    // [G] slots
    int D, Dq, G, accumulate, np;
    const int8_t* Rx;                     // [G][3][Dq][Dq]: the other three K quarters of the last plane's residues (or NULL)
};

// 1 / x for a scale: a power of two (exponent arithmetic: no division), 1 for an empty column, NaN for a non-finite one
__device__ __forceinline__ double recip_scale(double x) {
    const long long b = __double_as_longlong(x);
    const bool pow2 = (b & 0x000FFFFFFFFFFFFFll) == 0 && b > 0x0010000000000000ll && b < 0x7FD0000000000000ll;
    return pow2 ? __longlong_as_double(0x7FE0000000000000ll - b) : 1.0 / x;      // (anything else takes the division: NaN stays NaN)
}

// A thread reconstructs FOUR consecutive entries of a row: the residues of a plane are one 4-byte load instead of four single bytes
// (the byte-per-thread version issued 13 one-byte loads per entry and ran at 1.7 TB/s of its 21 bytes per entry), the result is two
// 16-byte stores.
__global__ __launch_bounds__(256) void i8_crt_kernel(CrtArgs a) {
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4, i = blockIdx.y, gz = blockIdx.z;
    if (j0 > i || i >= a.D) return;
    const int8_t* R = a.R + ((long)gz * a.np) * a.Dq * a.Dq + (long)i * a.Dq + j0;
    int w[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) w[q] = q < a.np ? *reinterpret_cast<const int*>(R + (long)q * a.Dq * a.Dq) : 0;
    int wx[3] = {0, 0, 0};
    if (a.Rx) {
        const int8_t* Rx = a.Rx + (long)gz * 3 * a.Dq * a.Dq + (long)i * a.Dq + j0;
#pragma unroll
        for (int k = 0; k < 3; ++k) wx[k] = *reinterpret_cast<const int*>(Rx + (long)k * a.Dq * a.Dq);
    }
    const double ra = recip_scale(a.sA[i]);
    double out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int v[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) v[q] = (int)(int8_t)((unsigned)w[q] >> (8 * e));
        // the last plane came in four K quarters: any representative of the sum will do (it only enters the last digit's reduction)
        if (a.Rx) {
            const int x = (int)(int8_t)((unsigned)wx[0] >> (8 * e)) + (int)(int8_t)((unsigned)wx[1] >> (8 * e)) + (int)(int8_t)((unsigned)wx[2] >> (8 * e));
#pragma unroll
            for (int q = 1; q < NP; ++q) if (q == a.np - 1) v[q] += x;
        }
        // Garner: mixed-radix digits with symmetric representatives, S = v0 + v1 p0 + v2 p0 p1 + ...;  digit q =
        // (r_q - sum_{k<q} v_k (p_0..p_{k-1} mod p_q)) (p_0..p_{q-1})^-1 mod p_q: the sum is accumulated unreduced (14 terms of at most
        // 128 * 128), so a digit costs q multiply-adds and ONE reduction instead of q reductions
#pragma unroll
        for (int q = 1; q < NP; ++q) {
            if (q < a.np) {
                const int p = MT.p[q];
                int u = v[q];
#pragma unroll
                for (int k = 0; k < q; ++k) u -= v[k] * MT.w[k][q];                          // |u| < 2^18
                int t = u * MT.pinv[q] % p;                                                   // < 2^26 in magnitude
                if (t > p / 2) t -= p; else if (t < -(p / 2)) t += p;
                v[q] = t;
            }
        }
        double s = 0.0;
#pragma unroll
        for (int q = NP - 1; q >= 0; --q)
            if (q < a.np) s = s * (double)MT.p[q] + (double)v[q];
        // non-finite data (a diverged chain): the scale of that column is NaN and so is every entry it takes part in, as the fp64 product
        // would give -- never a finite number made of garbage residues.  Two factors, applied one after the other: the product of two
        // scales may overflow.  The scales are powers of two, so multiplying by their reciprocals IS the division.
        const int j = j0 + e;
        const double rb = j < a.D ? recip_scale(a.sB[(long)gz * a.D + j]) : 0.0;
        out[e] = (s * ra) * rb;
    }
    double* dst = a.J + (long)gz * a.strideJ + (long)i * a.ldj + j0;
    if (j0 + 3 <= i && !a.accumulate && (reinterpret_cast<uintptr_t>(dst) % 16 == 0)) {
        *reinterpret_cast<d2_t*>(dst) = d2_t{out[0], out[1]};
        *reinterpret_cast<d2_t*>(dst + 2) = d2_t{out[2], out[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j0 + e <= i) dst[e] = a.accumulate ? dst[e] + out[e] : out[e];
    }
}

}  // namespace

// ------------------------------------------------------------------ host side
int pgl_k_i8_padded_rows(int D) { return (D + BT - 1) / BT * BT; }     // planes and residues are padded to the product kernel's tile edge

// time bins per plane row: a multiple of the 64-byte K tile, at least four tiles (the Gram pipeline keeps three requested ahead)
static long pgl_i8_kp(int T) { const long k = ((long)T + 63) / 64 * 64; return k < 256 ? 256 : k; }

// The integer columns' norms are kept at or below limit(K, T), the largest value with
//     (limit (1 + 1e-9) + sqrt(T) + 1)^2 <= prod(p_0..p_{K-1}) / 2
// (rounding a scaled column of omega X to the nearest integers adds at most sqrt(T)/2 to its norm, the dithered rounding of a column of X at
// most sqrt(T), the fp64 products another sqrt(T)/8 at most: (L + 1.125 sqrt T)(L + 0.625 sqrt T) <= (L + sqrt T)^2), so
// |S_ij| <= |A_i||B_j| stays inside the symmetric CRT range.  nu = floor(log2(limit)): 46 / 50 / 54 / 58 for K = 12 / 13 / 14 / 15.
double pgl_k_i8_norm_limit(int nplanes, int T) {
    double l2 = 0.0;
    for (int q = 0; q < nplanes && q < NP; ++q) l2 += std::log2((double)MT.p[q]);
    const double lim = (std::exp2((l2 - 1.0) * 0.5) - std::sqrt((double)(T < 1 ? 1 : T)) - 1.0) / (1.0 + 1e-9);
    return lim > 0.0 ? lim : 0.0;
}
int pgl_k_i8_nu(int nplanes, int T) {
    const double lim = pgl_k_i8_norm_limit(nplanes, T);
    if (!(lim >= 2.0)) return 0;
    return (int)std::floor(std::log2(lim) - 1e-12);
}
int pgl_k_i8_max_planes(void) { return NP; }
// fewest moduli that keep the rounding error of the scaled operands at the fp64 level: column norms >= 2^49 need nu >= 50
int pgl_k_i8_min_planes(int T) {
    for (int k = 1; k <= NP; ++k)
        if (pgl_k_i8_nu(k, T) >= 50) return k;
    return NP + 1;
}

long pgl_k_i8_kp(int T) { return pgl_i8_kp(T); }
// is the last plane of a product over G neurons cut into K quarters (whose residues go to the Rx slots)?  Groups that fill the per-XCD lists only
static int pgl_k_i8_kparts() { static const int v = pgl_ab_int("PGL_I8_KPARTS", 0); return v >= 2 && v <= 16 ? v : 0; }
static bool pgl_k_i8_split(int G, int nplanes) { return G % 8 == 0 && nplanes >= 2 && pgl_k_i8_kparts() == 0; }
size_t pgl_k_i8_plane_bytes(int D, int T) {
    const long Dq = pgl_k_i8_padded_rows(D), Kp = pgl_i8_kp(T);
    return (size_t)NP * Dq * Kp;
}
size_t pgl_k_i8_residue_bytes(int D) {
    const long Dq = pgl_k_i8_padded_rows(D);
    return (size_t)NP * Dq * Dq;
}

int pgl_k_i8_colstats(const double* X, long ldx, const double* Om, long ldo, int T, int D, int G, double* amax, double* ss, hipStream_t st) {
    if (G > CS_G || G < 1) { pgl_set_error("i8 colstats: %d weight columns per call (max %d)", G, CS_G); return PGL_ERR_ARG; }
    const dim3 grid((D + CS_COLS - 1) / CS_COLS), block(CS_COLS * CS_LANES);
#define PGL_CS(g_) case g_: hipLaunchKernelGGL((i8_colstats_kernel<g_, true>), grid, block, 0, st, X, ldx, Om, ldo, T, D, amax, ss); break;
    if (!Om) hipLaunchKernelGGL((i8_colstats_kernel<1, false>), grid, block, 0, st, X, ldx, Om, ldo, T, D, amax, ss);
    else switch (G) { PGL_CS(1) PGL_CS(2) PGL_CS(3) PGL_CS(4) PGL_CS(5) PGL_CS(6) PGL_CS(7) PGL_CS(8) default: break; }
#undef PGL_CS
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

// statistics + scales of omega_g X in one go, the time axis cut into NCH chunks so that the pass fills the chip: scratch holds 2 NCH G D doubles
// (partial maxima, then partial sums of squares).  The sums are formed in a fixed order (per chunk as in pgl_k_i8_colstats, then over the
// chunks), so the scales do not depend on launch timing; they can differ from pgl_k_i8_colstats + pgl_k_i8_scales in the last bit of a norm,
// i.e. only for a column whose norm sits on a power-of-two boundary.
constexpr int CS_NCH = 4;
size_t pgl_k_i8_stats_scratch_doubles(int D, int G) { return (size_t)2 * CS_NCH * G * D; }
int pgl_k_i8_colstats_scales(const double* X, long ldx, const double* Om, long ldo, int T, int D, int G, int nplanes, double* scratch, double* scale,
                             hipStream_t st) {
    if (G > CS_G || G < 1 || !Om) { pgl_set_error("i8 colstats+scales: %d weight columns per call (max %d)", G, CS_G); return PGL_ERR_ARG; }
    if (pgl_k_i8_nu(nplanes, T) < 8) { pgl_set_error("i8 scales: %d moduli leave no room for T = %d", nplanes, T); return PGL_ERR_ARG; }
    const int nch = T >= 4096 ? CS_NCH : 1;
    double* pm = scratch;
    double* pq = scratch + (size_t)CS_NCH * G * D;
    const dim3 grid((D + CS_COLS - 1) / CS_COLS, nch), block(CS_COLS * CS_LANES);
#define PGL_CS2(g_) case g_: hipLaunchKernelGGL((i8_colstats_kernel<g_, true>), grid, block, 0, st, X, ldx, Om, ldo, T, D, pm, pq); break;
    switch (G) { PGL_CS2(1) PGL_CS2(2) PGL_CS2(3) PGL_CS2(4) PGL_CS2(5) PGL_CS2(6) PGL_CS2(7) PGL_CS2(8) default: break; }
#undef PGL_CS2
    PGL_CHECK_LAUNCH();
    const long n = (long)G * D;
    hipLaunchKernelGGL(i8_scales_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pm, pq, n, pgl_k_i8_norm_limit(nplanes, T), scale, nch);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

// largest |omega| per weight column: out[c] (bit pattern of a non-negative double) for the n columns of Om, zeroed here
int pgl_k_i8_colmax(const double* Om, long ldo, int T, int n, double* out, hipStream_t st) {
    if (hipMemsetAsync(out, 0, (size_t)n * sizeof(double), st) != hipSuccess) { pgl_set_error("memset failed"); return PGL_ERR_HIP; }
    hipLaunchKernelGGL(i8_colmax_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(T >= 8192 ? 128 : 1)), dim3(256), 0, st, Om, ldo, T, n,
                       reinterpret_cast<unsigned long long*>(out));
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

// scales of G weight columns from their sums of squares ss[g][d] (leading dimension ldss) and the element bound ommax[g] * xmax[d]
int pgl_k_i8_scales_bound(const double* ss, long ldss, const double* ommax, const double* xmax, int D, int G, int T, int nplanes, double* scale,
                          hipStream_t st) {
    if (pgl_k_i8_nu(nplanes, T) < 8) { pgl_set_error("i8 scales: %d moduli leave no room for T = %d", nplanes, T); return PGL_ERR_ARG; }
    const long n = (long)G * D;
    hipLaunchKernelGGL(i8_scales_bound_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ss, ldss, ommax, xmax, D, G,
                       pgl_k_i8_norm_limit(nplanes, T), 1.0 + ((double)T + 16.0) * 1.2e-16, scale);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_i8_scales(const double* amax, const double* ss, long n, int T, int nplanes, double* scale, hipStream_t st) {
    if (pgl_k_i8_nu(nplanes, T) < 8) { pgl_set_error("i8 scales: %d moduli leave no room for T = %d", nplanes, T); return PGL_ERR_ARG; }
    hipLaunchKernelGGL(i8_scales_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, amax, ss, n, pgl_k_i8_norm_limit(nplanes, T), scale, 1);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

// residue planes of X (Om == null, G = 1) or of omega_g X for the G weight columns Om[:, 0..G)
int pgl_k_i8_planes(const double* X, long ldx, int transposed, const double* Om, long ldo, const double* scale, int8_t* P, int T, int D, int G,
                    int nplanes, long t_base, hipStream_t st) {
    const int Dq = pgl_k_i8_padded_rows(D);
    const long Kp = pgl_i8_kp(T);
    // 512 time bins per workgroup: 8 KiB contiguous per plane and row block.  Measured on one box, ms per group of 8 at cfg3, with the
    // non-temporal stores: 256 bins 13.6, 512 bins 12.5, 1024 bins 12.4 (with ordinary stores the three were within 2 %).
    const bool aligned = (ldx % 2 == 0) && (reinterpret_cast<uintptr_t>(X) % 16 == 0);
    for (int g0 = 0; g0 < G; g0 += CS_G) {                     // (the kernels stage at most CS_G weight columns; larger groups go in pieces)
        const int gz = G - g0 < CS_G ? G - g0 : CS_G;
        PlaneArgs a{X, ldx, transposed, Om ? Om + g0 : nullptr, ldo, scale + (long)g0 * D, P + (long)g0 * nplanes * Dq * Kp, T, D, Dq, Kp, nplanes, (unsigned)t_base};
        if (transposed && aligned)
            // measured on one box per variant pair, ms per group of 8 at cfg3 (the LDS-tile kernel on Xt: 11.9): 768 threads 11.3-11.4, 512 10.7-10.8
            // (137 VGPRs, one workgroup per CU; forced to 128 VGPRs with 16 spills: 11.3-11.4), 256 9.85-10.1, 128 10.3-10.6
            hipLaunchKernelGGL(i8_planes_t_kernel<256>, dim3((unsigned)((Kp + 255) / 256), Dq / PT_D), dim3(256), 0, st, a, gz);
        else
            hipLaunchKernelGGL(i8_planes_kernel<512>, dim3((unsigned)((Kp + 511) / 512), Dq / PT_D), dim3(512), 0, st, a, gz);
        PGL_CHECK_LAUNCH();
    }
    return PGL_OK;
}

// One time slice of the product: PB (and R's geometry) belong to the slice of T bins; the X planes start at K tile ka0 of rows that are KpA
// bytes long (KpA = 0: they are a slice of their own, same geometry as PB).  accumulate: add to the residues of the earlier slices.
int pgl_k_i8_gram(const int8_t* PA, long KpA, int ka0, const int8_t* PB, int8_t* R, int8_t* Rx, int T, int D, int G, int nplanes, int accumulate, hipStream_t st) {
    const int Dq = pgl_k_i8_padded_rows(D);
    const long Kp = pgl_i8_kp(T);
    if (KpA <= 0) { KpA = Kp; ka0 = 0; }
    const int nkt = (T + BKB - 1) / BKB;
    if (KpA % BKB != 0 || (long)(ka0 + nkt) * BKB > KpA) { pgl_set_error("i8 gram: slice of %d tiles at tile %d outside the X planes (%ld bytes per row)", nkt, ka0, KpA); return PGL_ERR_ARG; }
    // Three LDS stages (requests one tile ahead).  A fourth stage (two tiles ahead, all 160 KiB) measured 6.7 % SLOWER on real planes (107.2 vs
    // 100.5 ms per launch of 8 neurons at cfg3): the 32 workgroups of an XCD then hold 3 x 40 KiB in flight each, which is the whole 4 MiB
    // L2, and the strips they share fall out of it.
    static PglPerDevice attr;
    if (int rc = pgl_set_dynamic_lds(reinterpret_cast<const void*>(i8_gram_kernel), BNST * BSTAGE, attr)) return rc;
    const int n_cu = pgl_device_cus(pgl_device());
    const int ntm = Dq / BT, ntiles = ntm * (ntm + 1) / 2;
    const long total = (long)ntiles * nplanes * G;
    if (total <= 0) return PGL_OK;
    if (total * 4 > 0x7fffffffL) { pgl_set_error("i8 gram: %ld work items", total); return PGL_ERR_ARG; }
    const unsigned grid = (unsigned)(total < n_cu ? total : n_cu);
    // Tile order.  Per-XCD lists (an XCD owns whole neurons): FULL-HEIGHT column blocks two tiles wide -- a neuron's omega X strips (the
    // per-neuron half of the operand stream: nothing shares them but the co-walking workgroups of the XCD) are then fetched once per plane,
    // while the X strips, which every column block walks again, are shared by all eight XCDs through the memory-side cache.  Same-box A/B at
    // cfg3, ms per launch of 8 neurons: 6 x 6 100.3-100.6 | 8 x 4 101.5-103.7 | 4 x 8 101.1 | 3 x 12 103.5 | 12 x 3 99.2 | 16 x 4 98.9 |
    // 16 x 3 98.7 | 16 x 2 98.2-98.6 | 16 x 1 98.5-98.9 (profiles/archive/r04_i8_tile_order_ab.txt).  The flat list of a ragged group keeps 6 x 6.
    // (Pacing the eight XCDs' lists to within one to three rounds of each other -- a bounded wait at item boundaries on the other lists'
    // counters -- changed nothing: 96.9-97.1 ms without, 96.9-98.2 with, profiles/archive/r04_i8_xcd_pacing_ab.txt: they stay together on their own.)
    const bool lists = G % 8 == 0;
    static const int sbr_ab = pgl_ab_int("PGL_I8_SBR", 0), sbc_ab = pgl_ab_int("PGL_I8_SBC", 0);
    const int sbr = sbr_ab > 0 ? sbr_ab : (lists ? ntm : SB), sbc = sbc_ab > 0 ? sbc_ab : (lists ? 2 : SB);
    for (int kt0 = 0; kt0 < nkt; kt0 += KCH) {
        GramArgs g{PA, PB, R, Dq, Kp, G, nplanes, sbr, sbc, nkt, kt0, pgl_sched_slot(st), KpA, ka0, accumulate ? 1 : 0, pgl_k_i8_split(G, nplanes) ? Rx : nullptr, lists ? pgl_k_i8_kparts() : 0};
        if (!g.sched) { pgl_set_error("i8 gram: scheduler scratch unavailable"); return PGL_ERR_HIP; }
        hipLaunchKernelGGL(i8_gram_kernel, dim3(grid), dim3(256), BNST * BSTAGE, st, g);
        PGL_CHECK_LAUNCH();
    }
    return PGL_OK;
}

// Rx: the three extra residue slots per neuron of a product that ran with the last plane split (pgl_k_i8_gram with the same Rx and a group
// that is a multiple of 8); NULL otherwise
int pgl_k_i8_crt(const int8_t* R, const int8_t* Rx, const double* sA, const double* sB, double* J, long ldj, long strideJ, int D, int G, int nplanes,
                 int accumulate, hipStream_t st) {
    const int Dq = pgl_k_i8_padded_rows(D);
    if (G <= 0) return PGL_OK;
    CrtArgs c{R, sA, sB, J, ldj, strideJ, D, Dq, G, accumulate, nplanes, (Rx != nullptr && pgl_k_i8_split(G, nplanes)) ? Rx : nullptr};     // = the condition in pgl_k_i8_gram
    hipLaunchKernelGGL(i8_crt_kernel, dim3((D + 1023) / 1024, D, G), dim3(256), 0, st, c);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}
