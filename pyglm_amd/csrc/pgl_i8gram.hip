// The omega-weighted Gram  J_n = X' diag(omega_n) X  by exact integer arithmetic on the int8 MFMA (an OPT-IN alternative to the fp64
// kernel of pgl_gemm.hip).  The fp64 operands are scaled column by column to 50-bit integers,
//     A[t][i] = rint(x_ti 2^eA_i),      B_n[t][j] = rint(omega_nt x_tj 2^fB_nj),
// the integer Gram S = A'B_n is computed modulo 15 pairwise coprime moduli p <= 255 -- one int8 GEMM per modulus on residues in
// [-p/2, p/2], int32 accumulation (exact for K <= 131072) -- and reconstructed exactly by the Chinese remainder theorem
// (|S| <= K 2^100 < prod(p)/2 = 2^116.6);  J = S 2^-(eA_i + fB_nj).  The only approximation is the rounding of the operands to
// 50-bit fixed point per column: error ~1e-15 |a_i||b_j|, the level of an fp64 product at K = 1e5 (tools/ozaki2_accuracy.py).
//
//   i8_planes_kernel   fp64 (t-major) -> 15 residue planes, K (time) contiguous:  PA[q][d][t] for X (once per data set),
//                      PB[g][q][d][t] for omega_g X (per neuron, per sweep); one pass over X per neuron
//   i8_gram_kernel     R[g][q] = (PA[q] PB[g][q]') mod p_q on lower 256 x 256 tiles (v_mfma_i32_32x32x32_i8; 8 waves = 2 x 4, wave tile
//                      128 x 64; 64-byte K tiles DMA-staged into 3 LDS stages, 16-byte chunks XOR-swizzled: conflict-free ds_read_b128)
//   i8_crt_kernel      15 residues -> mixed-radix digits (Garner) -> fp64 by Horner -> scaled into the lower triangle of J
#include "pgl_common.h"

namespace {

constexpr int NP = 15;
// compile-time table: every use below sits in a fully unrolled loop, so reductions mod p become multiply-shift sequences
struct ModTable {
    int p[NP] = {255, 254, 253, 251, 247, 241, 239, 233, 229, 227, 223, 211, 199, 197, 193};
    int inv[NP][NP] = {};             // inv[j][i] = p_j^-1 mod p_i
    int m17[NP] = {}, m34[NP] = {};   // 2^17 mod p, 2^34 mod p
    constexpr ModTable() {
        for (int j = 0; j < NP; ++j)
            for (int i = 0; i < NP; ++i) {
                if (i == j) continue;
                const int a = p[j] % p[i];
                int x = 1;
                while ((a * x) % p[i] != 1) ++x;
                inv[j][i] = x;
            }
        for (int i = 0; i < NP; ++i) {
            long v = 1;
            for (int k = 0; k < 17; ++k) v = (v * 2) % p[i];
            m17[i] = (int)v;
            m34[i] = (int)((v * v) % p[i]);
        }
    }
};
constexpr ModTable MT{};
__constant__ int c_mod[NP] = {255, 254, 253, 251, 247, 241, 239, 233, 229, 227, 223, 211, 199, 197, 193};   // run-time indexed (Gram epilogue)
constexpr int BETA = 50;             // bits of the scaled integer operands

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ int scale_exp(double maxabs) {       // e with |v 2^e| < 2^BETA for every |v| <= maxabs
    if (!(maxabs > 0.0)) return 0;
    int ex;
    (void)frexp(maxabs, &ex);                                    // maxabs = m 2^ex, m in [0.5, 1)
    return BETA - ex;
}

// ------------------------------------------------------------------ column maxima of |X| (per data set) and of omega (per neuron)
__global__ __launch_bounds__(256) void colmax_kernel(const double* __restrict__ V, long ldv, int T, int ncol, double* __restrict__ out) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    __shared__ double red[4][64];
    double m = 0.0;
    if (c < ncol)
        for (int t = blockIdx.y * 4 + part; t < T; t += gridDim.y * 4) m = fmax(m, fabs(V[(long)t * ldv + c]));
    red[part][threadIdx.x & 63] = m;
    __syncthreads();
    if (part == 0 && c < ncol) {
        m = fmax(fmax(red[0][threadIdx.x], red[1][threadIdx.x]), fmax(red[2][threadIdx.x], red[3][threadIdx.x]));
        // non-negative doubles order like their bit patterns
        atomicMax(reinterpret_cast<unsigned long long*>(out + c), (unsigned long long)__double_as_longlong(m));
    }
}

// ------------------------------------------------------------------ fp64 -> residue planes
struct PlaneArgs {
    const double* X; long ldx;            // [T][ldx]
    const double* Om; long ldo;           // [T][ldo] weights of the group's neurons (null: unweighted, one "neuron")
    const double* xmax;                   // [D]
    const double* wmax;                   // [G] (weighted only)
    int8_t* P;                            // [G][NP][Dq][Kp]
    int T, D, Dq; long Kp;
};

// tile = 256 time bins x 16 columns of X in LDS (33 KB: four workgroups per CU keep enough loads in flight), read ONCE and converted for
// all G neurons of the group; a lane owns 4 consecutive time bins of one column and stores one packed dword per plane (256
// contiguous bytes per wave and plane row)
constexpr int PT_D = 16;
__global__ __launch_bounds__(256) void i8_planes_kernel(PlaneArgs a, int G) {
    __shared__ double tile[PT_D][257];
    const int t0 = blockIdx.x * 256, d0 = blockIdx.y * PT_D;
    const int tid = threadIdx.x;
    {
        const int dl = tid & (PT_D - 1), d = d0 + dl;
        for (int tl = tid / PT_D; tl < 256; tl += 256 / PT_D) {
            const int t = t0 + tl;
            tile[dl][tl] = (t < a.T && d < a.D) ? a.X[(long)t * a.ldx + d] : 0.0;
        }
    }
    __syncthreads();
    const int tg = tid & 63, tb = t0 + 4 * tg;
    if (tb >= a.Kp) return;
    for (int gz = 0; gz < G; ++gz) {
        double om[4] = {1.0, 1.0, 1.0, 1.0};
        if (a.Om) {
#pragma unroll
            for (int k = 0; k < 4; ++k) om[k] = tb + k < a.T ? a.Om[(long)(tb + k) * a.ldo + gz] : 0.0;
        }
        for (int dl = tid >> 6; dl < PT_D; dl += 4) {
            const int d = d0 + dl;
            const double scale = d < a.D ? ldexp(1.0, scale_exp(a.Om ? a.wmax[gz] * a.xmax[d] : a.xmax[d])) : 0.0;
            // the scaled integers as sign + three limbs |I| = a 2^34 + b 2^17 + c; residue = sign * ((a m34 + b m17 + c) mod p), folded
            // to [-p/2, p/2] -- integer multiply-adds and a division by a compile-time constant per plane
            int la[4], lb[4], lc[4], sg[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double v = rint(tile[dl][4 * tg + k] * om[k] * scale);      // x * omega rounded to fp64 first, as X*omega[:,None] is
                const double av = fabs(v);
                const double hi = floor(av * 0x1p-34);                             // exact: av is an integer < 2^50
                const double rem = av - hi * 0x1p34;
                const double mid = floor(rem * 0x1p-17);
                sg[k] = v < 0.0 ? -1 : 1;
                la[k] = (int)hi; lb[k] = (int)mid; lc[k] = (int)(rem - mid * 0x1p17);
            }
            int8_t* dst = a.P + (((long)gz * NP) * a.Dq + d) * a.Kp + tb;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int p = MT.p[q];
                unsigned w = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int r = (la[k] * MT.m34[q] + lb[k] * MT.m17[q] + lc[k]) % p;      // < 2^16 * 255 + 2^17 * 255 + 2^17 < 2^26
                    if (r > p / 2) r -= p;
                    r *= sg[k];
                    w |= (unsigned)(r & 0xff) << (8 * k);
                }
                *reinterpret_cast<unsigned*>(dst + (long)q * a.Dq * a.Kp) = w;
            }
        }
    }
}

// ------------------------------------------------------------------ int8 Gram of the planes, reduced mod p
constexpr int TM = 256, TN = 256, BKB = 64, NST = 4;
constexpr int STAGE_BYTES = (TM + TN) * BKB;
constexpr int GRAM_LDS = NST * STAGE_BYTES;

struct GramArgs {
    const int8_t* PA;                     // [NP][Dq][Kp]
    const int8_t* PB;                     // [G][NP][Dq][Kp]
    int8_t* R;                            // [G][NP][Dq][Dq]
    int Dq; long Kp; int G;
};

__device__ __forceinline__ int isqrt_tri_i(int t) {
    int r = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)(r + 1) * (r + 2) / 2 <= t) ++r;
    while ((long)r * (r + 1) / 2 > t) --r;
    return r;
}

__global__ __launch_bounds__(512) void i8_gram_kernel(GramArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int ntm = g.Dq / TM;
    const int ntiles = ntm * (ntm + 1) / 2;
    // item = (neuron of the group, plane, tile), tile fastest: the workgroups resident together work on one plane pair and share its
    // 256-row strips through L2 (plane-fastest order streamed every strip from HBM: 517 GB per launch, 5.3 TB/s, memory-bound)
    const int tile = blockIdx.x % ntiles, q = (blockIdx.x / ntiles) % NP, gz = blockIdx.x / (ntiles * NP);
    if (gz >= g.G) return;
    const int tm = isqrt_tri_i(tile), tn = tile - tm * (tm + 1) / 2;
    const int m0 = tm * TM, n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int8_t* A = g.PA + (long)q * g.Dq * g.Kp;
    const int8_t* B = g.PB + ((long)gz * NP + q) * g.Dq * g.Kp;
    const int nkt = (int)(g.Kp / BKB);

    // DMA: per K tile 512 rows x 64 B = 32 requests of 1 KiB (16 rows); wave w issues requests w, w+8, w+16, w+24.  Lane l of a request
    // fills row 16 rq + l / 4, physical 16-byte chunk l % 4, with logical chunk (l % 4) ^ ((row >> 2) & 3)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const char* gp[4];
    int loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rq = wv + 8 * i, row = 16 * rq + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int8_t* base = row < TM ? A + (long)(m0 + row) * g.Kp : B + (long)(n0 + row - TM) * g.Kp;
        gp[i] = reinterpret_cast<const char*>(base) + lc * 16;
        loff[i] = rq * 1024;
    }
    auto dma = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[i], (lds_ptr_t)(lds + stage * STAGE_BYTES + loff[i]), 16, 0, 0);
            gp[i] += BKB;
        }
    };
    v16i acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    const int fr = lane & 31, fk = lane >> 5;
    auto frag = [&](int stage, int row, int kk) {
        const int pc = (kk * 2 + fk) ^ ((row >> 2) & 3);
        return *reinterpret_cast<const v4i*>(lds + stage * STAGE_BYTES + row * BKB + pc * 16);
    };
    // Pipeline: NST - 1 = 3 tiles requested ahead; fragments double-buffered in registers, every LDS read and DMA request issued in
    // the shadow of an MFMA (order pinned with sched_barrier); the barrier sits in the MIDDLE of a K tile (after its first k-step):
    // it publishes tile kt+1, whose first fragments are fetched during the second k-step, and frees the stage of tile kt-1 for the
    // requests of tile kt+3.  Bare s_barrier: a __syncthreads() would drain the outstanding requests (vmcnt(0)).
    long gadv = BKB;                 // 0 once the last tile has been requested: the cursors stop and the last tile is requested again
    auto dma_piece = [&](int stage, int i) {      // (into a stage nobody reads any more) -- no branch in front of a request
        __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[i], (lds_ptr_t)(lds + stage * STAGE_BYTES + loff[i]), 16, 0, 0);
        gp[i] += gadv;
    };
    for (int p = 0; p < NST - 1; ++p) if (p < nkt) dma(p);
    if (nkt >= NST - 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v4i F[2][6];
    auto rd = [&](int stage, int kk, int f) {
        return f < 4 ? frag(stage, wm * 128 + f * 32 + fr, kk) : frag(stage, TM + wn * 64 + (f - 4) * 32 + fr, kk);
    };
#pragma unroll
    for (int f = 0; f < 6; ++f) F[0][f] = rd(0, 0, f);
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int nxt = cur == NST - 1 ? 0 : cur + 1;
        const int dst = cur == 0 ? NST - 1 : cur - 1;          // the stage of tile kt-1 takes tile kt+NST-1
        if (kt + NST >= nkt) gadv = 0;
        // ---- first k-step (set 0): prefetch the fragments of this tile's second k-step
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[0][m >> 1], F[0][4 + (m & 1)], acc[m >> 1][m & 1], 0, 0, 0);
            if (m < 6) F[1][m] = rd(cur, 1, m);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // tile kt+1 has landed; the 4 requests of tile kt+2 may stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- second k-step (set 1): prefetch the first fragments of tile kt+1, request tile kt+3
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[1][m >> 1], F[1][4 + (m & 1)], acc[m >> 1][m & 1], 0, 0, 0);
            if (m < 6) F[0][m] = rd(nxt, 0, m);
            if (m & 1) dma_piece(dst, m >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = nxt;
    }
    // epilogue: reduce mod p (symmetric) and store bytes.  C/D layout of 32x32 i32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const double p = (double)c_mod[q], ip = 1.0 / p;
    int8_t* R = g.R + ((long)gz * NP + q) * g.Dq * g.Dq;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                const double v = (double)acc[i][j][r];
                double rr = fma(-p, rint(v * ip), v);
                if (rr > 0.5 * p) rr -= p; else if (rr < -0.5 * p) rr += p;
                R[(long)row * g.Dq + col] = (int8_t)(int)rr;
            }
}

// ------------------------------------------------------------------ CRT reconstruction into J
struct CrtArgs {
    const int8_t* R;                      // [G][NP][Dq][Dq]
    const double* xmax; const double* wmax;
    double* J; long ldj; long strideJ;    // [G] slots
    int D, Dq, G, accumulate;
};

__global__ __launch_bounds__(256) void i8_crt_kernel(CrtArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, gz = blockIdx.z;
    if (j > i || i >= a.D) return;
    const int8_t* R = a.R + ((long)gz * NP) * a.Dq * a.Dq + (long)i * a.Dq + j;
    int v[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) v[q] = (int)R[(long)q * a.Dq * a.Dq];
    // Garner: mixed-radix digits with symmetric representatives, S = v0 + v1 p0 + v2 p0 p1 + ...
#pragma unroll
    for (int q = 1; q < NP; ++q) {
        const int p = MT.p[q];
        int t = v[q];
#pragma unroll
        for (int r = 0; r < q; ++r) t = (t - v[r]) * MT.inv[r][q] % p;                // |t - v| < 2^9, inverse < 2^8: no overflow
        if (t > p / 2) t -= p; else if (t < -(p / 2)) t += p;
        v[q] = t;
    }
    double s = (double)v[NP - 1];
#pragma unroll
    for (int q = NP - 2; q >= 0; --q) s = s * (double)MT.p[q] + (double)v[q];
    const int e = scale_exp(a.xmax[i]) + scale_exp(a.wmax[gz] * a.xmax[j]);
    const double val = ldexp(s, -e);
    double* dst = a.J + (long)gz * a.strideJ + (long)i * a.ldj + j;
    *dst = a.accumulate ? *dst + val : val;
}

}  // namespace

// ------------------------------------------------------------------ host side
// time bins per plane row: a multiple of the 64-byte K tile, at least four tiles (the Gram pipeline keeps three requested ahead)
static long pgl_i8_kp(int T) { const long k = ((long)T + 63) / 64 * 64; return k < 256 ? 256 : k; }

size_t pgl_k_i8_plane_bytes(int D, int T) {
    const long Dq = (D + 255) / 256 * 256, Kp = pgl_i8_kp(T);
    return (size_t)15 * Dq * Kp;
}
size_t pgl_k_i8_residue_bytes(int D) {
    const long Dq = (D + 255) / 256 * 256;
    return (size_t)15 * Dq * Dq;
}

// column maxima of |V| (out must be zero-filled by the caller: maxima are merged with atomicMax)
int pgl_k_i8_colmax(const double* V, long ldv, int T, int ncol, double* out, hipStream_t st) {
    hipLaunchKernelGGL(colmax_kernel, dim3((ncol + 63) / 64, 64), dim3(256), 0, st, V, ldv, T, ncol, out);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

// residue planes of X (Om == null, G = 1) or of omega_g X for the G weight columns Om[:, 0..G)
int pgl_k_i8_planes(const double* X, long ldx, const double* Om, long ldo, const double* xmax, const double* wmax, int8_t* P, int T, int D, int G,
                    hipStream_t st) {
    const int Dq = (D + 255) / 256 * 256;
    const long Kp = pgl_i8_kp(T);
    PlaneArgs a{X, ldx, Om, ldo, xmax, wmax, P, T, D, Dq, Kp};
    hipLaunchKernelGGL(i8_planes_kernel, dim3((unsigned)((Kp + 255) / 256), Dq / PT_D), dim3(256), 0, st, a, G);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

int pgl_k_i8_gram(const int8_t* PA, const int8_t* PB, int8_t* R, const double* xmax, const double* wmax, double* J, long ldj, long strideJ,
                  int T, int D, int G, int accumulate, hipStream_t st) {
    if (T > 131072) { pgl_set_error("i8 gram: T = %d > 131072 would overflow the int32 accumulators", T); return PGL_ERR_ARG; }
    const int Dq = (D + 255) / 256 * 256;
    const long Kp = pgl_i8_kp(T);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(i8_gram_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GRAM_LDS);
        if (e != hipSuccess) { pgl_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PGL_ERR_HIP; }
        attr = true;
    }
    const int ntm = Dq / TM, ntiles = ntm * (ntm + 1) / 2;
    GramArgs g{PA, PB, R, Dq, Kp, G};
    hipLaunchKernelGGL(i8_gram_kernel, dim3((unsigned)(ntiles * 15 * G)), dim3(512), GRAM_LDS, st, g);
    PGL_CHECK_LAUNCH();
    CrtArgs c{R, xmax, wmax, J, ldj, strideJ, D, Dq, G, accumulate};
    hipLaunchKernelGGL(i8_crt_kernel, dim3((D + 255) / 256, D, G), dim3(256), 0, st, c);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}
