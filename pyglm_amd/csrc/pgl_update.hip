// Batched rank-k updates  C[b] = beta C[b] + alpha A[b]' B[b]  on the fp64 MFMA, for the two places of the sweep that are made of them:
//   * the collapsed flips (pyglm/regression.py:282-320 as a sweep tableau, pgl_flips.hip): M -= W'U over the lower triangle of every
//     neuron's tableau, rank 512 in the initial sweep, rank <= 320 after a proposal window, and W = G U in front of it;
//   * the blocked Cholesky of the weight draw (pyglm/regression.py:323-340, pgl_chol.hip): the rank-256 trailing updates (upper form).
// These are short-K, read-modify-write products on per-neuron operands: an output tile lives for K / 16 K tiles only, its operand panels
// belong to one neuron, and C itself streams through HBM once per pass.  The generic kernel of pgl_gemm.hip (128 x 128 tiles, register
// staging, two workgroups per CU) ran them at 0.66 (rank 256) to 0.79 (rank 512) of the fp64 MFMA peak.  This kernel is the pipeline of the
// omega-weighted Gram (pgl_gemm.hip, 0.93 of peak) rebuilt for them:
//   * 8 waves hold a 256 x 128 (lower / full) or 128 x 256 (upper) tile of C: two 128 x 128 halves that share the narrow operand's staged
//     tile, 32 x 128 per wave = 16 accumulators -- a third less operand traffic per flop than two independent 128 x 128 tiles;
//   * K tiles of 16 rows DMA-staged (global_load_lds, 1 KiB per wave-instruction, per-lane cursors) into three LDS stages, the barrier in
//     the middle of a K tile, every fragment read and request issued in the shadow of an MFMA (order pinned with sched_barrier);
//   * the accumulators START from the C tile (alpha = -1: from -C, stored negated -- bit-identical to accumulating -A'B onto C), so the
//     tile's read latency overlaps the first requests and the epilogue only stores;
//   * persistent workgroups, one per CU, pull items from per-XCD lists; a neuron's tiles come in super-blocks of 4 x 8 tiles, so that the
//     32 workgroups of an XCD stream 4 wide + 8 narrow strips of ONE neuron's panels together and share them through that XCD's L2
//     (row-major order: ~33 strips per 32 items);
//   * on the diagonal, the half of a tile that lies wholly outside the triangle idles (it keeps its share of the requests and barriers).
// The summation order of every output element is the one of the generic kernel (K tiles in order, four k per MFMA), so the two give the
// same bits; tests/test_gpu_update.py holds them against each other.
#include "pgl_common.h"

namespace {

constexpr int BK = 16;
constexpr int PAD = 16;
constexpr int NST = 3;
constexpr int SBMAJ = 4, SBMIN = 8;      // super-block: 4 wide-operand strips x 8 narrow-operand strips

template <bool WIDE_B>
struct UCfg {
    static constexpr int BM = WIDE_B ? 128 : 256, BN = WIDE_B ? 256 : 128;
    static constexpr int SA = BM + PAD, SB = BN + PAD;
    static constexpr int A_ELEMS = BK * SA, B_ELEMS = BK * SB, STAGE = A_ELEMS + B_ELEMS;
    static constexpr size_t LDS_BYTES = (size_t)NST * STAGE * sizeof(double) + 32;     // + the work ticket
};

struct UpdArgs {
    PglGemmArgs g;
    int nmaj, nmin;          // tiles along the wide (256) and the narrow (128) operand
    int ntiles;              // items per batch
    int cinit;               // 1: alpha = +-1, beta = 1 (accumulators start from C); 0: beta = 0
    int stagger;             // 1: XCD y starts y/8 of an item late (read-modify-write items)
};

// item t of a batch -> (major, minor) tile in super-block order; tri: only tiles with minor <= 2 major + 1 exist.
// Super-rows are skipped by their totals, so the walk is at most nI + nJ steps (one thread does it per item).
__device__ __forceinline__ int row_tiles(int m, int nmaj, int nmin, bool tri) {
    if (m >= nmaj) return 0;
    int hi = nmin - 1;
    if (tri && 2 * m + 1 < hi) hi = 2 * m + 1;
    return hi + 1;
}
__device__ __forceinline__ void decode_tile(int t, int nmaj, int nmin, bool tri, int& maj, int& mnr) {
    const int nI = (nmaj + SBMAJ - 1) / SBMAJ, nJ = (nmin + SBMIN - 1) / SBMIN;
    for (int I = 0; I < nI; ++I) {
        int rt[SBMAJ], tot = 0;
#pragma unroll
        for (int r = 0; r < SBMAJ; ++r) { rt[r] = row_tiles(I * SBMAJ + r, nmaj, nmin, tri); tot += rt[r]; }
        if (t >= tot) { t -= tot; continue; }
        for (int J = 0; J < nJ; ++J) {
            const int lo = J * SBMIN;
            int c[SBMAJ], cnt = 0;
#pragma unroll
            for (int r = 0; r < SBMAJ; ++r) {
                int n = rt[r] - lo;
                n = n < 0 ? 0 : n > SBMIN ? SBMIN : n;
                c[r] = n;
                cnt += n;
            }
            if (t < cnt) {
#pragma unroll
                for (int r = 0; r < SBMAJ; ++r) {
                    if (t < c[r]) { maj = I * SBMAJ + r; mnr = lo + t; return; }
                    t -= c[r];
                }
            }
            t -= cnt;
        }
    }
    maj = mnr = 0;
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <bool WIDE_B>
__device__ __forceinline__ void upd_item(const UpdArgs& u, const int batch, const int maj, const int mnr, double* smem) {
    using C = UCfg<WIDE_B>;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const PglGemmArgs& g = u.g;
    const int tm = WIDE_B ? mnr : maj, tn = WIDE_B ? maj : mnr;
    int Mv = g.M, Nv = g.N;
    if (g.batch_dim) {
        const int d = g.batch_dim[batch] - g.dim_off;
        if (g.dim_mode == 0) { Mv = d; Nv = d; }
        else if (g.dim_mode == 1) Nv = d;
        else if (g.dim_mode == 2) Mv = d;
        else { Mv = g.M < d ? g.M : d; Nv = d; }
    }
    const int m0 = tm * C::BM, n0 = tn * C::BN;
    if (m0 >= Mv || n0 >= Nv) return;
    int K = g.K;
    if (g.batch_k) { K = g.batch_k[batch]; if (K <= 0) return; }
    const int nkt = K / BK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wz = wv >> 2, wq = wv & 3;
    const int frow = lane >> 4, fcol = lane & 15;
    // this wave's 32 x 128 slab of the tile
    const int rloc = WIDE_B ? wq * 32 : wz * 128 + wq * 32;
    const int cloc = WIDE_B ? wz * 128 : 0;
    // a half tile wholly outside the triangle: lower (tri 1): rows of half wz = 0 end before the columns begin when mnr == 2 maj + 1;
    // upper (tri 2): columns of half wz = 0 end before the rows begin, same condition
    const bool idle = g.tri != 0 && mnr == 2 * maj + 1 && wz == 0;

    const long roff_ = g.batch_row_off ? (long)g.batch_row_off[batch] : 0;     // rows of B and C this batch starts at
    const double* __restrict__ Ab = g.A + (long)batch * g.strideA;
    const double* __restrict__ Bb = g.B + (long)batch * g.strideB + roff_ * g.ldb;

    // ---- DMA pieces: a K tile is 16 rows of (BM + BN) / 128 = 3 segments of 128 doubles; wave w moves rows w and w + 8: 6 requests.
    // Every request keeps a per-lane global cursor that moves down one K tile after use, and a wave-uniform LDS offset inside a stage.
    const char* gp[6];
    long gstep[6], gadv[6];
    int loff[6];
    {
        constexpr int ASEG = C::BM / 128;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int r = wv + 8 * (p / 3), s = p % 3;
            if (s < ASEG) {
                int col = m0 + s * 128 + lane * 2;
                col = col < g.a_cols ? col : g.a_cols - 2;           // clamped columns only feed outputs that are never stored
                gp[p] = reinterpret_cast<const char*>(Ab + (long)r * g.lda + col);
                gstep[p] = (long)BK * g.lda * 8;
                loff[p] = (r * C::SA + s * 128) * 8;
            } else {
                const int sb = s - ASEG;
                int col = n0 + sb * 128 + lane * 2;
                col = col < g.b_cols ? col : g.b_cols - 2;
                gp[p] = reinterpret_cast<const char*>(Bb + (long)r * g.ldb + col);
                gstep[p] = (long)BK * g.ldb * 8;
                loff[p] = (C::A_ELEMS + r * C::SB + sb * 128) * 8;
            }
        }
    }
    auto advance = [&](bool adv) {
#pragma unroll
        for (int p = 0; p < 6; ++p) gadv[p] = adv ? gstep[p] : 0;
    };
    auto dma_piece = [&](int stage, int p) {
        char* dst = reinterpret_cast<char*>(smem) + (size_t)stage * C::STAGE * 8 + loff[p];
        __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[p], (lds_ptr_t)dst, 16, 0, 0);
        asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(gp[p]) : "v"(gp[p]), "s"(gadv[p]));
    };
    auto dma = [&](int stage) {
#pragma unroll
        for (int p = 0; p < 6; ++p) dma_piece(stage, p);
    };

    if (idle) {
        advance(1 < nkt); dma(0);
        advance(2 < nkt); dma(1);
        advance(NST < nkt);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        lds_barrier();
        int cur = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            const int dstage = cur == 0 ? NST - 1 : cur - 1;
            if (kt + NST >= nkt) advance(false);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            dma(dstage);
            cur = cur == NST - 1 ? 0 : cur + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- accumulators: tile f = 8 i + j (i < 2 row blocks, j < 8 column blocks of 16) in acc[f >> 2][f & 3];
    // f64 C/D fragment: row = (lane >> 4) + 4 reg, col = lane & 15
    d4_t acc[4][4];
    double* __restrict__ Cb = g.C + (long)batch * g.strideC + roff_ * g.ldc + (long)(m0 + rloc + frow) * g.ldc + (n0 + cloc + fcol);
    const bool interior = m0 + C::BM <= Mv && n0 + C::BN <= Nv;
    const bool neg = g.alpha < 0.0;
    if (u.cinit) {
        if (interior) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[(8 * i + j) >> 2][j & 3][r] = Cb[(long)(i * 16 + 4 * r) * g.ldc + j * 16];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = m0 + rloc + i * 16 + frow + 4 * r, col = n0 + cloc + j * 16 + fcol;
                        acc[(8 * i + j) >> 2][j & 3][r] = (row < Mv && col < Nv) ? Cb[(long)(i * 16 + 4 * r) * g.ldc + j * 16] : 0.0;
                    }
        }
    }
    advance(1 < nkt); dma(0);
    advance(2 < nkt); dma(1);
    advance(NST < nkt);
    if (u.cinit) {
        if (neg) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = -acc[a][b];
        }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // tile 0 (and, issued before it, the C tile) has landed
    lds_barrier();

    double pa[2][2][2], pb[2][2][8];       // [register set][step in pair][fragment]
    auto rdA = [&](int buf, int kk, int i) { return smem[buf * C::STAGE + rloc + fcol + (kk * 4 + frow) * C::SA + i * 16]; };
    auto rdB = [&](int buf, int kk, int j) { return smem[buf * C::STAGE + C::A_ELEMS + cloc + fcol + (kk * 4 + frow) * C::SB + j * 16]; };
#pragma unroll
    for (int i = 0; i < 2; ++i) { pa[0][0][i] = rdA(0, 0, i); pa[0][1][i] = rdA(0, 1, i); }
#pragma unroll
    for (int j = 0; j < 8; ++j) pb[0][0][j] = rdB(0, 0, j);
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int nxt = (cur == NST - 1) ? 0 : cur + 1;
        const int dstage = (cur == 0) ? NST - 1 : cur - 1;          // the stage of tile kt-1 takes tile kt+2
        if (kt + NST >= nkt) advance(false);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ps = p, ns = p ^ 1;
            const int nbuf = (p == 0) ? cur : nxt, nk = (p == 0) ? 2 : 0;
            // even step: B fragments of this pair's odd step, A fragments of the next pair, four requests
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int i = m >> 3, j = m & 7;
                acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ps][0][i], pb[ps][0][j], acc[m >> 2][m & 3], 0, 0, 0);
                if (m < 8) pb[ps][1][m] = rdB(cur, 2 * p + 1, m);
                else if (m < 10) pa[ns][0][m - 8] = rdA(nbuf, nk, m - 8);
                else if (m < 12) pa[ns][1][m - 10] = rdA(nbuf, nk + 1, m - 10);
                else { if (p == 1) dma_piece(dstage, m - 12); }
                __builtin_amdgcn_sched_barrier(0);
            }
            // odd step: B fragments of the next pair's even step, the last two requests
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int i = m >> 3, j = m & 7;
                acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ps][1][i], pb[ps][1][j], acc[m >> 2][m & 3], 0, 0, 0);
                if (m < 8) pb[ns][0][m] = rdB(nbuf, nk, m);
                else if (m < 10) { if (p == 1) dma_piece(dstage, 4 + m - 8); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (p == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile kt+1 has landed
                lds_barrier();
            }
        }
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the redundant last requests must land before the stages are refilled

    // ---- epilogue
    if (u.cinit) {
        if (neg) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = -acc[a][b];
        }
    } else {
        const double alpha = g.alpha;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = acc[a][b] * alpha;
    }
    if (interior) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) Cb[(long)(i * 16 + 4 * r) * g.ldc + j * 16] = acc[(8 * i + j) >> 2][j & 3][r];
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = m0 + rloc + i * 16 + frow + 4 * r, col = n0 + cloc + j * 16 + fcol;
                    if (row < Mv && col < Nv) Cb[(long)(i * 16 + 4 * r) * g.ldc + j * 16] = acc[(8 * i + j) >> 2][j & 3][r];
                }
    }
}

// persistent: one workgroup per CU; the flat list of (batch, tile) items is cut into eight contiguous chunks, one per XCD (a run of whole
// neurons each); a workgroup reads the XCC it runs on and pulls the next item of that XCD's chunk, stealing from the next XCD at the end.
// Placement is used for speed only -- any placement gives the same result.
template <bool WIDE_B>
__global__ __launch_bounds__(512, 1) void update_kernel(UpdArgs u) {
    using C = UCfg<WIDE_B>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int* ticket = reinterpret_cast<int*>(smem + (size_t)NST * C::STAGE);
    const long total = (long)u.ntiles * u.g.nbatch;
    const long chunk = (total + 7) / 8;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    // The workgroups of a launch run in step (same K, same item length), and a read-modify-write item begins and ends with 256 KiB of C
    // traffic per CU: in step, all 256 CUs would hit HBM together and idle their MFMAs meanwhile (measured: 22 us of a 131 us item).
    // XCD y therefore starts y/8 of an item late -- the eight XCDs' memory phases interleave, while the 32 workgroups of one XCD stay in
    // step and keep sharing their operand strips through its L2.  One item = K/16 K tiles x 8192 cycles per CU; s_sleep(127) ~ 8128 cycles.
    if (u.stagger) {
        const int naps = (int)(xcc * (unsigned)(u.g.K / 16) / 8u);
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // thread 0 keeps the NEXT ticket in flight while the current item runs (its latency is then hidden behind a whole item)
    int ysrc = (int)xcc, hops = 0;
    long it = 0;
    if (threadIdx.x == 0) it = atomicAdd(&u.g.sched[ysrc], 1);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            long w = -1;
            while (hops < 8) {
                const long lo = (long)ysrc * chunk, hi = (lo + chunk < total) ? lo + chunk : total;
                if (lo + it < hi) { w = lo + it; break; }
                ++hops;
                ysrc = (ysrc + 1) & 7;
                if (hops < 8) it = atomicAdd(&u.g.sched[ysrc], 1);
            }
            int maj = 0, mnr = 0, batch = -1;
            if (w >= 0) {
                it = atomicAdd(&u.g.sched[ysrc], 1);
                batch = (int)(w / u.ntiles);
                decode_tile((int)(w % u.ntiles), u.nmaj, u.nmin, u.g.tri != 0, maj, mnr);
            }
            ticket[0] = batch; ticket[1] = maj; ticket[2] = mnr;
        }
        __syncthreads();
        const int batch = __builtin_amdgcn_readfirstlane(ticket[0]);
        if (batch < 0) break;
        const int maj = __builtin_amdgcn_readfirstlane(ticket[1]), mnr = __builtin_amdgcn_readfirstlane(ticket[2]);
        upd_item<WIDE_B>(u, batch, maj, mnr, smem);
    }
}

// ---- the last few rows of a lower-triangular update.  A tableau has D + 2 rows: D = N B is usually a multiple of 256 and the bias and
// potential rows then open a row block of their own -- 41 of 462 items for two rows at cfg3.  Those rows are a skinny product (nrows <= 16
// rows x all columns x K) bound by reading B once: one wave per 32 columns, fragments straight from global memory, the same MFMA and the
// same K order as the tiles (so the same bits), accumulators started from C.
__global__ __launch_bounds__(256) void update_rows_kernel(UpdArgs u, int row0, int nrows) {
    const PglGemmArgs& g = u.g;
    const int batch = blockIdx.y;
    int K = g.K;
    if (g.batch_k) { K = g.batch_k[batch]; if (K <= 0) return; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane >> 4, fcol = lane & 15;
    const int c0 = blockIdx.x * 128 + wave * 32;
    const int Nv = g.N;
    if (c0 >= Nv) return;
    const double* __restrict__ Ab = g.A + (long)batch * g.strideA;
    const double* __restrict__ Bb = g.B + (long)batch * g.strideB;
    double* __restrict__ Cb = g.C + (long)batch * g.strideC;
    const bool neg = g.alpha < 0.0;
    d4_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = frow + 4 * r, col = c0 + j * 16 + fcol;
            double v = 0.0;
            if (u.cinit && rr < nrows && col < Nv) v = Cb[(long)(row0 + rr) * g.ldc + col];
            acc[j][r] = neg ? -v : v;
        }
    int ca = row0 + fcol;
    ca = ca < g.a_cols ? ca : g.a_cols - 1;
    int cb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { cb[j] = c0 + j * 16 + fcol; cb[j] = cb[j] < g.b_cols ? cb[j] : g.b_cols - 1; }
    const double* ap = Ab + (long)frow * g.lda + ca;
    const double* bp0 = Bb + (long)frow * g.ldb + cb[0];
    const double* bp1 = Bb + (long)frow * g.ldb + cb[1];
    const long sa = 4 * g.lda, sb = 4 * g.ldb;
    for (int k0 = 0; k0 < K; k0 += 32) {             // K is a multiple of 16: 16 or 32 rows per trip, eight independent loads per operand
        double a[8], b0[8], b1[8];
        const int steps = (K - k0) >= 32 ? 8 : 4;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bool on = s < steps;
            a[s] = on ? ap[s * sa] : 0.0;
            b0[s] = on ? bp0[s * sb] : 0.0;
            b1[s] = on ? bp1[s * sb] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < steps) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b0[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b1[s], acc[1], 0, 0, 0);
            }
        }
        ap += 8 * sa; bp0 += 8 * sb; bp1 += 8 * sb;
    }
    const double alpha = g.alpha;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = frow + 4 * r, col = c0 + j * 16 + fcol;
            if (rr >= nrows || col >= Nv) continue;
            const double v = acc[j][r];
            Cb[(long)(row0 + rr) * g.ldc + col] = u.cinit ? (neg ? -v : v) : alpha * v;
        }
}

template <bool WIDE_B>
int launch_update(const PglGemmArgs& a0, hipStream_t st) {
    using C = UCfg<WIDE_B>;
    static PglPerDevice attr_set;
    auto kern = update_kernel<WIDE_B>;
    if (int rc = pgl_set_dynamic_lds(reinterpret_cast<const void*>(kern), C::LDS_BYTES, attr_set)) return rc;
    UpdArgs u{};
    u.g = a0;
    u.cinit = a0.beta == 1.0 ? 1 : 0;
    if constexpr (!WIDE_B) {
        // lower triangle of a fixed size whose last few rows would open a row block of their own: those rows go through the skinny kernel
        const int extra = a0.M % C::BM;
        if (a0.tri == 1 && a0.batch_dim == nullptr && a0.M > C::BM && extra >= 1 && extra <= 16) {
            const int row0 = a0.M - extra;
            u.g.sched = nullptr;
            hipLaunchKernelGGL(update_rows_kernel, dim3((unsigned)((a0.N + 127) / 128), (unsigned)a0.nbatch), dim3(256), 0, st, u, row0, extra);
            PGL_CHECK_LAUNCH();
            u.g.M = row0; u.g.N = row0;
        }
    }
    const PglGemmArgs& a = u.g;
    const int ntm = (a.M + C::BM - 1) / C::BM, ntn = (a.N + C::BN - 1) / C::BN;
    u.nmaj = WIDE_B ? ntn : ntm;
    u.nmin = WIDE_B ? ntm : ntn;
    long nt = 0;
    for (int m = 0; m < u.nmaj; ++m) {
        int hi = u.nmin - 1;
        if (a.tri && 2 * m + 1 < hi) hi = 2 * m + 1;
        nt += hi + 1;
    }
    u.ntiles = (int)nt;
    u.stagger = u.cinit;
    const long total = nt * a.nbatch;
    if (total <= 0) return PGL_OK;
    if (total > 0x7fffffffL) { pgl_set_error("update: %ld work items", total); return PGL_ERR_ARG; }
    u.g.sched = pgl_sched_slot(st);
    if (!u.g.sched) { pgl_set_error("update: scheduler scratch unavailable"); return PGL_ERR_HIP; }
    const int n_cu = pgl_device_cus(pgl_device());
    const long grid = total < n_cu ? total : n_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C::LDS_BYTES, st, u);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

}  // namespace

// Can this product run on the update pipeline?  (alpha = +-1 with beta = 1, or beta = 0; tri 1 = lower with the wide operand A, tri 2 =
// upper with the wide operand B, tri 0 = all tiles; square for the triangular forms.)
bool pgl_update_supported(const PglGemmArgs& a) {
    if (a.W != nullptr || a.K % BK != 0 || a.M <= 0 || a.N <= 0 || a.nbatch <= 0) return false;
    if (!((a.beta == 1.0 && (a.alpha == 1.0 || a.alpha == -1.0)) || a.beta == 0.0)) return false;
    if (a.tri != 0 && a.M != a.N) return false;
    if (a.a_cols % 2 || a.b_cols % 2 || a.lda % 2 || a.ldb % 2 || a.a_cols < 2 || a.b_cols < 2) return false;
    if (((uintptr_t)a.A % 16) || ((uintptr_t)a.B % 16) || (a.strideA % 2) || (a.strideB % 2)) return false;
    return true;
}

int pgl_launch_update(const PglGemmArgs& a, hipStream_t st) {
    if (!pgl_update_supported(a)) { pgl_set_error("update pipeline: unsupported product"); return PGL_ERR_ARG; }
    return a.tri == 2 ? launch_update<true>(a, st) : launch_update<false>(a, st);
}

// rows [row0, row0 + nrows) (nrows <= 16) of C (+)= alpha A'B over all N columns, through the skinny kernel: what the last few rows of a
// lower-triangular update cost when they are not given a tile row of their own (a tableau has D + 2 rows: two rows that would be 41 tiles
// of 861 at cfg3).  Same MFMA, same K order as the tiles: same bits.
int pgl_launch_update_rows(const PglGemmArgs& a, int row0, int nrows, hipStream_t st) {
    if (!pgl_update_supported(a) || nrows < 1 || nrows > 16 || row0 < 0 || a.batch_dim != nullptr) { pgl_set_error("update rows: unsupported product"); return PGL_ERR_ARG; }
    UpdArgs u{};
    u.g = a;
    u.cinit = a.beta == 1.0 ? 1 : 0;
    hipLaunchKernelGGL(update_rows_kernel, dim3((unsigned)((a.N + 127) / 128), (unsigned)a.nbatch), dim3(256), 0, st, u, row0, nrows);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}
