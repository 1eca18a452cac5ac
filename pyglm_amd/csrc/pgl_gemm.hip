// fp64 MFMA contraction for gfx950:  C[m][n] = beta*C + alpha * sum_k w_z[k] * A[k][m] * B[k][n]
//
// This one template is the omega-weighted Gram  J_n = X' diag(omega_n) X  (reference
// pyglm/regression.py:251-252, `XO = X*omega[:,None]; XO.T.dot(X)` -- here the T x D temporary is never
// formed: omega scales the A fragment in registers), the batched activation Psi = X W' (:195-201), the
// border sums X'omega, X'kappa (:253-260) and the symmetric rank-k updates of the flip tableau / Cholesky.
//
// Mapping to CDNA4 (measured, tools/ubench2_f64.hip): v_mfma_f64_16x16x4_f64 issues once per 64 cycles per SIMD
// (78.6 TFLOP/s chip peak at 2.4 GHz) from a single wave with a single accumulator; f64 VALU shares that pipe and every
// MFMA -> VALU -> MFMA switch costs, so f64 VALU work inside the loop is kept minimal and clustered.
//   * both operands are k-major ("TN"): a wave reads an A or B fragment as lanes (l&15) -> 16 consecutive doubles of row
//     k0+(l>>4); LDS row stride = tile width + 16 doubles (== 16 mod 32) makes every fragment read conflict-free and the
//     global->LDS copy fully coalesced (1 KiB per wave-instruction).
//   * 16 accumulator tiles (128 VGPRs) per wave, workgroup = WM x WN x WZ waves; WZ = 2 puts two neurons (two weight
//     columns) on the same staged X tiles: 32 flop per byte staged.
//   * two pipelines in gemm_item: STAGES = 2 (register-staged prefetch, one barrier per K tile, 64 x 64 wave tiles: the plain
//     contraction and the rank-k updates) and STAGES = 3, the Gram pipeline (global_load_lds DMA staging, barrier in the
//     middle of a K tile, 32 x 128 wave tiles, every LDS read and DMA piece issued in the shadow of an MFMA, one cluster of
//     omega multiplies per pair of k-steps).  DESIGN.md section 3.1 has the measurements behind each of these choices.
//     gram_fine_item is the same Gram for launches with few neurons: 4-wave workgroups, one neuron each, two per CU.
//   * XCD-aware order: work item w = xcd*chunk + slot with the batch (neuron pair) fastest, so the workgroups resident on
//     one XCD share the same X panels through that XCD's L2; the Gram is launched persistently (one workgroup per CU
//     pulling XCD-local items) because the dispatcher's round-robin drifts over long launches.
#include "pgl_common.h"
#include <type_traits>

namespace {

constexpr int BK = 16;
constexpr int PAD = 16;

template <int WM, int WN, int WZ, bool WEIGHTED, int STAGES = 2>
struct Cfg {
    static constexpr int BM = WM * 64, BN = WN * 64;
    static constexpr int SA = BM + PAD, SB = BN + PAD;
    static constexpr int THREADS = WM * WN * WZ * 64;
    static constexpr int A_ELEMS = BK * SA, B_ELEMS = BK * SB, W_ELEMS = WEIGHTED ? BK * WZ : 0;
    static constexpr int STAGE = A_ELEMS + B_ELEMS + W_ELEMS;
    static constexpr size_t LDS_BYTES = (size_t)STAGES * STAGE * sizeof(double);
    // 16-byte loads per thread per K-tile
    static constexpr int A_LD = (BK * BM / 2) / THREADS, B_LD = (BK * BN / 2) / THREADS;
    static_assert((BK * BM / 2) % THREADS == 0 && (BK * BN / 2) % THREADS == 0, "tile/threads mismatch");
};

__device__ __forceinline__ int isqrt_tri(int t) {
    // largest r with r(r+1)/2 <= t
    int r = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)(r + 1) * (r + 2) / 2 <= t) ++r;
    while ((long)r * (r + 1) / 2 > t) --r;
    return r;
}

// wave-level LDS hand-off without draining global loads that are still in flight (a plain __syncthreads() may)
__device__ __forceinline__ void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// one work item = one output tile of one batch (weighted mode: of one group of WZ weight columns)
// CINIT (2-stage pipeline, alpha = +-1, beta = 1): the accumulators START from the C tile (loads issued together with the first K
// tile, so their latency overlaps the pipeline fill) and the epilogue only stores.  The rank-k tableau / Cholesky updates are
// short (K = 128..320) read-modify-write items whose serial epilogue loads cost as much as a third of the MFMA time.
// SQ: both operands are squared element by element on their way to LDS -- C = sum_k A[k][m]^2 B[k][n]^2, the sums of squares of the columns of
// omega_n X for a whole batch of neurons as ONE contraction (the norms the integer Gram scales its operands from, pgl_sweep.hip)
template <int WM, int WN, int WZ, bool WEIGHTED, int STAGES, bool DMA = false, bool CINIT = false, bool SQ = false>
__device__ __forceinline__ void gemm_item(const PglGemmArgs& g, const long w, double* smem) {
    using C = Cfg<WM, WN, WZ, WEIGHTED, STAGES>;
    const int ntm = (g.M + C::BM - 1) / C::BM;
    int tile, batch, tm, tn;
    if (g.nbatch > 1) {
        if (WEIGHTED || (g.strideA == 0 && g.strideB == 0)) {
            // shared operands (the Gram's X): the same tile of consecutive batches runs side by side and shares its panels in L2
            tile = (int)(w / g.nbatch); batch = (int)(w % g.nbatch);
        } else {
            // per-batch operands (tableau / Cholesky panels): nothing is shared between batches, so each XCD walks one batch's tiles
            // in order -- a row of tiles re-reads the same A strip and the B strips of one batch stay L2-resident
            const int ntn_ = (g.N + C::BN - 1) / C::BN;
            const long ntiles = g.tri ? (long)ntm * (ntm + 1) / 2 : (long)ntm * ntn_;
            batch = (int)(w / ntiles); tile = (int)(w % ntiles);
        }
    } else { tile = (int)w; batch = 0; }
    if (g.tri) { tm = isqrt_tri(tile); tn = tile - tm * (tm + 1) / 2; if (g.tri == 2) { const int s_ = tm; tm = tn; tn = s_; } }
    else { tm = tile % ntm; tn = tile / ntm; }
    int Mv = g.M, Nv = g.N;
    if (g.batch_dim) {
        const int d = g.batch_dim[batch] - g.dim_off;
        if (g.dim_mode == 0) { Mv = d; Nv = d; }
        else if (g.dim_mode == 1) Nv = d;
        else if (g.dim_mode == 2) Mv = d;
        else { Mv = g.M < d ? g.M : d; Nv = d; }     // 3: a strip of at most g.M rows of a d x d remainder
    }
    if (tm * C::BM >= Mv || tn * C::BN >= Nv) return;

    int K = g.K;
    if (g.batch_k) { K = g.batch_k[batch]; if (K <= 0) return; }
    const int nkt = K / BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wz = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
    const int m0 = tm * C::BM, n0 = tn * C::BN;

    const double* __restrict__ Ab = g.A + (long)batch * g.strideA;
    const long roff_ = g.batch_row_off ? (long)g.batch_row_off[batch] : 0;     // rows of B and C this batch starts at
    const double* __restrict__ Bb = g.B + (long)batch * g.strideB + roff_ * g.ldb;
    const int zg = batch * WZ + wz;                   // weight column / output index of this wave
    const bool zvalid = !WEIGHTED || zg < g.nz_total;

    // ---- loader geometry: a row of BM (BN) doubles is BM/2 (BN/2) 16-byte pieces.  Every 16-byte load keeps a cursor that moves down
    // one K tile per use (no per-load index arithmetic in the loop); columns beyond a_cols / b_cols are clamped to a readable pair:
    // they only feed output rows / columns that are never stored.
    d2_t ra[C::A_LD], rb[C::B_LD];                   // (ext-vector type: HIP's double2 struct ends up in scratch here)
    double rw = 0.0;
    // CINIT: alpha = +-1 is folded into the A operand on its way to LDS -- as a flip of the sign bits (integer XOR: an f64 multiply
    // by -1 gives the same bits but goes through the pipe the MFMAs use, and every MFMA -> f64 VALU -> MFMA switch costs)
    const unsigned long long aflip = (CINIT && g.alpha < 0.0) ? 0x8000000000000000ull : 0ull;
    typedef unsigned long long u2_t __attribute__((ext_vector_type(2)));
    auto signed_a = [&](d2_t v) { return __builtin_bit_cast(d2_t, __builtin_bit_cast(u2_t, v) ^ u2_t{aflip, aflip}); };
    unsigned oa_[C::A_LD], ob_[C::B_LD], ow_ = 0;                  // per-thread byte offsets inside a K tile (constant)
    const char* ka_ = reinterpret_cast<const char*>(Ab);            // wave-uniform bases of the K tile to load next
    const char* kb_ = reinterpret_cast<const char*>(Bb);
    const char* kw_ = reinterpret_cast<const char*>(g.W);
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
        const int p = tid + i * C::THREADS, r = p / (C::BM / 2), c = (p % (C::BM / 2)) * 2;
        int col = m0 + c;
        col = col < g.a_cols ? col : g.a_cols - 2;
        oa_[i] = (unsigned)(((long)r * g.lda + col) * 8);
    }
#pragma unroll
    for (int i = 0; i < C::B_LD; ++i) {
        const int p = tid + i * C::THREADS, r = p / (C::BN / 2), c = (p % (C::BN / 2)) * 2;
        int col = n0 + c;
        col = col < g.b_cols ? col : g.b_cols - 2;
        ob_[i] = (unsigned)(((long)r * g.ldb + col) * 8);
    }
    if (WEIGHTED) {
        const int t_ = tid < BK * WZ ? tid : 0, r = t_ / WZ, z = t_ % WZ;
        int zc = batch * WZ + z;
        zc = zc < g.nz_total ? zc : g.nz_total - 1;
        ow_ = (unsigned)(((long)r * g.ldw + zc) * 8);
    }
    const long stepA = (long)BK * g.lda * 8, stepB = (long)BK * g.ldb * 8, stepW = (long)BK * g.ldw * 8;
    auto gload = [&](int) {
#pragma unroll
        for (int i = 0; i < C::A_LD; ++i) ra[i] = *reinterpret_cast<const d2_t*>(ka_ + oa_[i]);
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) rb[i] = *reinterpret_cast<const d2_t*>(kb_ + ob_[i]);
        if (WEIGHTED && tid < BK * WZ) rw = *reinterpret_cast<const double*>(kw_ + ow_);
        ka_ += stepA; kb_ += stepB; kw_ += stepW;
    };
    auto lstore = [&](int buf) {
        double* As = smem + buf * C::STAGE;
        double* Bs = As + C::A_ELEMS;
#pragma unroll
        for (int i = 0; i < C::A_LD; ++i) {
            const int p = tid + i * C::THREADS, r = p / (C::BM / 2), c = (p % (C::BM / 2)) * 2;
            if constexpr (CINIT) *reinterpret_cast<d2_t*>(As + r * C::SA + c) = signed_a(SQ ? ra[i] * ra[i] : ra[i]);
            else *reinterpret_cast<d2_t*>(As + r * C::SA + c) = SQ ? ra[i] * ra[i] : ra[i];
        }
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
            const int p = tid + i * C::THREADS, r = p / (C::BN / 2), c = (p % (C::BN / 2)) * 2;
            *reinterpret_cast<d2_t*>(Bs + r * C::SB + c) = SQ ? rb[i] * rb[i] : rb[i];
        }
        if (WEIGHTED && tid < BK * WZ) (Bs + C::B_ELEMS)[tid] = rw;
    };

    d4_t acc[4][4];
    const int frow = lane >> 4, fcol = lane & 15;
    const bool interior = CINIT && m0 + C::BM <= Mv && n0 + C::BN <= Nv;     // workgroup-uniform: no bounds checks inside the tile
    if constexpr (CINIT) {
        const double* __restrict__ cp = g.C + (long)batch * g.strideC + roff_ * g.ldc + (long)(m0 + wm * 64 + frow) * g.ldc + (n0 + wn * 64 + fcol);
        if (interior) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j][r] = cp[(long)(i * 16 + 4 * r) * g.ldc + j * 16];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = m0 + wm * 64 + i * 16 + frow + 4 * r, col = n0 + wn * 64 + j * 16 + fcol;
                        acc[i][j][r] = (row < Mv && col < Nv) ? cp[(long)(i * 16 + 4 * r) * g.ldc + j * 16] : 0.0;
                    }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
    auto compute = [&](int buf, int kk) {
        const double* As = smem + buf * C::STAGE + wm * 64 + fcol;
        const double* Bs = smem + buf * C::STAGE + C::A_ELEMS + wn * 64 + fcol;
        const double* Ws = smem + buf * C::STAGE + C::A_ELEMS + C::B_ELEMS;
        const int kr = kk * 4 + frow;
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[kr * C::SA + i * 16];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[kr * C::SB + j * 16];
        if (WEIGHTED) {
            const double wv = Ws[kr * WZ + wz];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] *= wv;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    if constexpr (STAGES == 2) {
        // two LDS stages, one barrier at the end of every K tile.  In the unweighted kernels (rank-k updates, activation, panel
        // products: short K, two workgroups per CU running the same code in phase) the next tile's global loads ride in the MFMA
        // shadows of the first k-step and its LDS stores in those of the last one, one request per MFMA, instead of standing as two
        // blocks of ~40 instructions between the MFMA bursts.
        gload(0);
        lstore(0);
        __syncthreads();
        if constexpr (!WEIGHTED) {
            constexpr int NLD = C::A_LD + C::B_LD;
            static_assert(NLD <= 16, "one staged load per MFMA slot of a k-step");
            auto gload_piece = [&](int q) {
                if (q < C::A_LD) ra[q] = *reinterpret_cast<const d2_t*>(ka_ + oa_[q]);
                else rb[q - C::A_LD] = *reinterpret_cast<const d2_t*>(kb_ + ob_[q - C::A_LD]);
            };
            auto lstore_piece = [&](int buf, int q) {
                double* As = smem + buf * C::STAGE;
                double* Bs = As + C::A_ELEMS;
                if (q < C::A_LD) {
                    const int p = tid + q * C::THREADS, r = p / (C::BM / 2), c = (p % (C::BM / 2)) * 2;
                    if constexpr (CINIT) *reinterpret_cast<d2_t*>(As + r * C::SA + c) = signed_a(SQ ? ra[q] * ra[q] : ra[q]);
                    else *reinterpret_cast<d2_t*>(As + r * C::SA + c) = SQ ? ra[q] * ra[q] : ra[q];
                } else {
                    const int i = q - C::A_LD;
                    const int p = tid + i * C::THREADS, r = p / (C::BN / 2), c = (p % (C::BN / 2)) * 2;
                    *reinterpret_cast<d2_t*>(Bs + r * C::SB + c) = SQ ? rb[i] * rb[i] : rb[i];
                }
            };
            auto kstep = [&](int buf, int kk, auto hook) {
                const double* As = smem + buf * C::STAGE + wm * 64 + fcol;
                const double* Bs = smem + buf * C::STAGE + C::A_ELEMS + wn * 64 + fcol;
                const int kr = kk * 4 + frow;
                double a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = As[kr * C::SA + i * 16];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bs[kr * C::SB + j * 16];
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m >> 2], b[m & 3], acc[m >> 2][m & 3], 0, 0, 0);
                    hook(m);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (CINIT && NLD <= 8) {
                // The read-modify-write products (rank-k updates of the tableau and the Cholesky).  PMC: their MFMA pipe is busy 0.81 of the time --
                // the two waves of a SIMD take the pipe turn by turn and so reach their k-step boundaries together, where each reads its eight
                // fragments and only then issues (profiles/archive/r04_update_kernel_pmc.md).  Here the fragments of k-step kk + 1 are read in the MFMA
                // shadows of k-step kk into a second register set.  That only fits because the ACCUMULATORS LIVE IN AGPRs: the MFMAs are issued as
                // inline assembly with a register-class constraint (left to itself hipcc keeps the 128 accumulator registers in VGPRs and, with a
                // second fragment set, spills accumulator tiles inside the K loop).  93 VGPRs + 128 AGPRs of the 256 a wave may have with two
                // workgroups per CU.  Same MFMAs on the same operands in the same order: same bits.  Rank 512: 63.6 -> 65.2 TFLOP/s.
                // (Also moving the barrier between k-steps 2 and 3, so that not even a tile's first fragments are read in the open, keeps the
                // staged global loads alive across it: 140+ VGPRs against the 128 left beside the accumulators -- it spills and runs at 55.)
                double fa0[4], fb0[4], fa1[4], fb1[4];
                auto rd_first = [&](int buf) {             // k-step 0 of the tile in `buf` (just published by the barrier): the one read in the open
                    const double* As = smem + buf * C::STAGE + wm * 64 + fcol;
                    const double* Bs = smem + buf * C::STAGE + C::A_ELEMS + wn * 64 + fcol;
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa0[i] = As[frow * C::SA + i * 16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb0[j] = Bs[frow * C::SB + j * 16];
                };
                auto kstep_a = [&](int buf, auto kk_c, auto hook) {       // k-step KK on register set KK & 1; KK < 3: prefetches k-step KK + 1
                    constexpr int KK = decltype(kk_c)::value;
                    double (&ca)[4] = (KK & 1) ? fa1 : fa0;
                    double (&cb)[4] = (KK & 1) ? fb1 : fb0;
                    double (&na)[4] = (KK & 1) ? fa0 : fa1;
                    double (&nb)[4] = (KK & 1) ? fb0 : fb1;
                    const double* As = smem + buf * C::STAGE + wm * 64 + fcol;
                    const double* Bs = smem + buf * C::STAGE + C::A_ELEMS + wn * 64 + fcol;
                    const int krn = (KK + 1) * 4 + frow;
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[m >> 2][m & 3]) : "v"(ca[m >> 2]), "v"(cb[m & 3]));
                        if constexpr (KK < 3) {
                            if (m < 4) na[m] = As[krn * C::SA + m * 16];
                            else if (m < 8) nb[m - 4] = Bs[krn * C::SB + (m - 4) * 16];
                        }
                        hook(m);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
                using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
                auto nohook = [](int) {};
                rd_first(0);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // the accumulators were just written (C tile): wait states before the first MFMA reads them
                for (int kt = 0; kt + 1 < nkt; ++kt) {
                    const int buf = kt & 1;
                    kstep_a(buf, K0{}, [&](int m) { if (m >= 8 && m - 8 < NLD) gload_piece(m - 8); });
                    ka_ += stepA; kb_ += stepB;
                    kstep_a(buf, K1{}, nohook);
                    kstep_a(buf, K2{}, nohook);
                    kstep_a(buf, K3{}, [&](int m) { if (m >= 16 - NLD) lstore_piece(buf ^ 1, m - (16 - NLD)); });
                    __syncthreads();
                    rd_first(buf ^ 1);
                }
                {
                    const int buf = (nkt - 1) & 1;
                    kstep_a(buf, K0{}, nohook);
                    kstep_a(buf, K1{}, nohook);
                    kstep_a(buf, K2{}, nohook);
                    kstep_a(buf, K3{}, nohook);
                }
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // MFMA results readable
            } else {
            for (int kt = 0; kt + 1 < nkt; ++kt) {
                const int buf = kt & 1;
                kstep(buf, 0, [&](int m) { if (m < NLD) gload_piece(m); });
                ka_ += stepA; kb_ += stepB;
                compute(buf, 1);
                compute(buf, 2);
                kstep(buf, 3, [&](int m) { if (m >= 16 - NLD) lstore_piece(buf ^ 1, m - (16 - NLD)); });
                __syncthreads();
            }
            {
                const int buf = (nkt - 1) & 1;
#pragma unroll
                for (int kk = 0; kk < BK / 4; ++kk) compute(buf, kk);
            }
            }
        } else {
            for (int kt = 0; kt < nkt; ++kt) {
                const int buf = kt & 1;
                if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
                for (int kk = 0; kk < BK / 4; ++kk) compute(buf, kk);
                if (kt + 1 < nkt) lstore(buf ^ 1);
                __syncthreads();
            }
        }
    } else {
        // ---- the Gram pipeline: three LDS stages filled by DMA, the barrier in the MIDDLE of a K tile.  Tile kt+1 (issued a whole
        // tile earlier) is published by the barrier after the first pair of k-steps of tile kt; the DMA of tile kt+2 then reuses the
        // stage last read in tile kt-1, which every wave had finished before it could pass the barrier.  Because tile kt+1 is
        // readable from the middle of tile kt on, the fragment pipeline never drains at a tile boundary.
        static_assert(DMA && WEIGHTED && (STAGES == 3 || STAGES == 4), "the 3/4-stage pipeline is the DMA-staged weighted Gram");
        // DMA staging (global_load_lds): one wave-instruction moves one 1-KiB tile row straight into LDS (lane-linear
        // destination = exactly our row layout; the padded row stride only moves the per-instruction base).  Out-of-range
        // columns are clamped to a readable one: they only feed outputs that are never stored.
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        typedef const __attribute__((address_space(1))) void* glb_ptr_t;
        // one tile = five DMA wave-instructions per wave: pieces 0..3 = (A row, B row) x 2, piece 4 = this wave's 8 weight dwords.
        // Every piece keeps a per-lane global cursor that is advanced by one K tile after each request, and a wave-uniform LDS
        // offset inside a stage: a request is {m0 = stage base + offset; global_load_lds; 64-bit add}.  (Scalar bases + 32-bit lane
        // offsets instead of cursors measured 0.5 % slower: the compiler rebuilds a 64-bit vector address per request.)  (Recomputing
        // (krow + r) * ld per piece cost a dozen dependent scalar instructions and a branch in front of every request -- longer than
        // the MFMA shadow it sat in, and both waves of a SIMD reach the same slot together: the MFMA pipe idled ~9 %.)  Past the last
        // tile the cursors stop and the last tile is simply requested again into a stage nobody reads any more.
        static_assert(C::BM == 128 && C::BN == 128 && C::THREADS == 512, "DMA staging is written for the Gram tile");
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const char* gp[5];
        long gstep[5];
        int loff[5];
        {
            int ca = m0 + lane * 2, cb = n0 + lane * 2;
            ca = ca < g.a_cols ? ca : g.a_cols - 2;      // out-of-range columns are clamped to a readable one: they only feed
            cb = cb < g.b_cols ? cb : g.b_cols - 2;      // outputs that are never stored
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = wv + 8 * (p >> 1);
                if ((p & 1) == 0) { gp[p] = reinterpret_cast<const char*>(Ab + (long)r * g.lda + ca); gstep[p] = (long)BK * g.lda * 8; loff[p] = r * C::SA * 8; }
                else { gp[p] = reinterpret_cast<const char*>(Bb + (long)r * g.ldb + cb); gstep[p] = (long)BK * g.ldb * 8; loff[p] = (C::A_ELEMS + r * C::SB) * 8; }
            }
            // 16 x WZ doubles = 64 dwords, 8 per wave (lanes 0-7)
            const int dw = wv * 8 + (lane & 7), dbl = dw >> 1, r = dbl / WZ, zq = dbl % WZ;
            int zc = batch * WZ + zq;
            zc = zc < g.nz_total ? zc : g.nz_total - 1;
            gp[4] = reinterpret_cast<const char*>(g.W + (long)r * g.ldw + zc) + 4 * (dw & 1);
            gstep[4] = (long)BK * g.ldw * 8;
            loff[4] = (C::A_ELEMS + C::B_ELEMS) * 8 + wv * 32;
        }
        long gadv[5];
        auto advance = [&](bool adv) {       // cursor step of the requests that follow (0 once the last tile has been requested)
#pragma unroll
            for (int p = 0; p < 5; ++p) gadv[p] = adv ? gstep[p] : 0;
        };
        auto dma_piece = [&](int stage, int p) {
            char* dst = reinterpret_cast<char*>(smem) + (size_t)stage * C::STAGE * 8 + loff[p];
            if (p < 4) __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[p], (lds_ptr_t)dst, 16, 0, 0);
            else if (lane < 8) __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[4], (lds_ptr_t)dst, 4, 0, 0);
            // the 64-bit add stays in this slot (left to the compiler, all five end up in one clump at the loop latch)
            asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(gp[p]) : "v"(gp[p]), "s"(gadv[p]));
        };
        auto dma = [&](int stage) {
#pragma unroll
            for (int p = 0; p < 5; ++p) dma_piece(stage, p);
        };
        // (A fourth stage -- requests a whole tile earlier -- measured the same 72.8 TFLOP/s: request latency is not what is left.)
        // prologue: tiles 0 .. STAGES-2 requested, tile 0 awaited (past the end of K a request repeats the last tile into a stage
        // that is never read)
        advance(1 < nkt);
        dma(0);
        advance(2 < nkt);
        dma(1);
        if constexpr (STAGES == 4) { advance(3 < nkt); dma(2); }
        advance(STAGES < nkt);
        if constexpr (STAGES == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __syncthreads();
        int cur = 0;
        {
            // Wide wave tile for the Gram: each of a neuron's four waves owns 32 rows x all 128 columns of the workgroup tile
            // (2 A fragments x 8 B fragments per k-step instead of 4 x 4).  The omega multiplies scale A fragments, so this
            // halves them (4 per pair of k-steps per wave, and no two waves scale the same A data any more) at the price of
            // two more LDS reads per k-step.  Accumulator tile f = i*8 + j lives in acc[f >> 2][f & 3].
            // Paired k-steps as below: all fragments of the NEXT pair are fetched in the MFMA shadows of the current pair, one
            // cluster of multiplies per pair.
            const int wq = wmn;                              // 0..3: 32-row slab of this wave
            double pa[2][2][2], pb[2][2][8], pw[2][2];       // [pair set][step in pair][fragment]
            auto rdA = [&](int buf, int kk, int i) { return smem[buf * C::STAGE + wq * 32 + fcol + (kk * 4 + frow) * C::SA + i * 16]; };
            auto rdB = [&](int buf, int kk, int j) { return smem[buf * C::STAGE + C::A_ELEMS + fcol + (kk * 4 + frow) * C::SB + j * 16]; };
            auto rdW = [&](int buf, int kk) { return smem[buf * C::STAGE + C::A_ELEMS + C::B_ELEMS + (kk * 4 + frow) * WZ + wz]; };
#pragma unroll
            for (int i = 0; i < 2; ++i) { pa[0][0][i] = rdA(0, 0, i); pa[0][1][i] = rdA(0, 1, i); }
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[0][0][j] = rdB(0, 0, j);
            pw[0][0] = rdW(0, 0); pw[0][1] = rdW(0, 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) { pa[0][0][i] *= pw[0][0]; pa[0][1][i] *= pw[0][1]; }
            for (int kt = 0; kt < nkt; ++kt) {
                const int nxt = (cur == STAGES - 1) ? 0 : cur + 1;
                const int dstage = (cur == 0) ? STAGES - 1 : cur - 1;          // the stage of tile kt-1 takes tile kt+STAGES-1
                if (kt + STAGES >= nkt) advance(false);                          // (taken for the last tiles only)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int ps = p, ns = p ^ 1;
                    const int nbuf = (p == 0) ? cur : nxt, nk = (p == 0) ? 2 : 0;
                    // ---- even step: B fragments of this pair's odd step, A fragments + weights of the next pair, 4 DMA pieces
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const int i = m >> 3, j = m & 7;
                        acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ps][0][i], pb[ps][0][j], acc[m >> 2][m & 3], 0, 0, 0);
                        if (m < 8) pb[ps][1][m] = rdB(cur, 2 * p + 1, m);
                        else if (m < 10) pa[ns][0][m - 8] = rdA(nbuf, nk, m - 8);
                        else if (m == 10) pw[ns][0] = rdW(nbuf, nk);
                        else if (m == 11) pw[ns][1] = rdW(nbuf, nk + 1);
                        else { if (p == 1) dma_piece(dstage, m - 12); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // ---- odd step: remaining fragments of the next pair, last DMA piece, the 4 multiplies in one cluster
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const int i = m >> 3, j = m & 7;
                        acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ps][1][i], pb[ps][1][j], acc[m >> 2][m & 3], 0, 0, 0);
                        if (m < 2) pa[ns][1][m] = rdA(nbuf, nk + 1, m);
                        else if (m < 10) pb[ns][0][m - 2] = rdB(nbuf, nk, m - 2);
                        else if (m == 10) { if (p == 1) dma_piece(dstage, 4); }
                        else if (m == 13) {
#pragma unroll
                            for (int q = 0; q < 2; ++q) { pa[ns][0][q] *= pw[ns][0]; pa[ns][1][q] *= pw[ns][1]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (p == 0) {
                        // tile kt+1 must have landed; with four stages the requests of tile kt+2 may stay in flight
                        if constexpr (STAGES == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        block_sync_lds();
                    }
                }
                cur = nxt;
            }
        }
    }

    // ---- epilogue.  f64 C/D fragment: row = (lane>>4) + 4*reg, col = lane&15 (verified on hardware)
    if (!zvalid) return;
    double* __restrict__ Cb = g.C + (WEIGHTED ? (long)zg : (long)batch) * g.strideC + roff_ * g.ldc;
    const double alpha = g.alpha, beta = g.beta;
    // accumulator tile (i, j) of acc[4][4] -> offsets inside the workgroup tile: 64 x 64 wave tiles, or (wide Gram layout)
    // tile f = 4 i + j of a 32 x 128 wave tile
    constexpr bool WIDE = STAGES >= 3;
    auto roff = [&](int i, int j) { return WIDE ? wmn * 32 + ((4 * i + j) >> 3) * 16 : wm * 64 + i * 16; };
    auto coff = [&](int i, int j) { return WIDE ? ((4 * i + j) & 7) * 16 : wn * 64 + j * 16; };
    if constexpr (CINIT) {
        double* __restrict__ cq = Cb + (long)(m0 + wm * 64 + frow) * g.ldc + (n0 + wn * 64 + fcol);
        if (interior) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) cq[(long)(i * 16 + 4 * r) * g.ldc + j * 16] = acc[i][j][r];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = m0 + wm * 64 + i * 16 + frow + 4 * r, col = n0 + wn * 64 + j * 16 + fcol;
                        if (row < Mv && col < Nv) cq[(long)(i * 16 + 4 * r) * g.ldc + j * 16] = acc[i][j][r];
                    }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double cv[4][4];
        if (beta != 0.0) {   // all 16 read-modify-write loads of this row block in flight before the first use
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = m0 + roff(i, j) + frow + 4 * r, col = n0 + coff(i, j) + fcol;
                    cv[r][j] = (row < Mv && col < Nv) ? Cb[(long)row * g.ldc + col] : 0.0;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = m0 + roff(i, j) + frow + 4 * r, col = n0 + coff(i, j) + fcol;
                if (row >= Mv || col >= Nv) continue;
                double v = alpha * acc[i][j][r];
                if (beta != 0.0) v += beta * cv[r][j];
                Cb[(long)row * g.ldc + col] = v;
            }
        }
    }
}

// ---- plain launch: one workgroup per work item, XCD-aware order (block b runs on XCD b % 8: consecutive slots of one XCD
// walk the batch index of the same tile, so co-resident workgroups share operand panels through that XCD's L2)
template <int WM, int WN, int WZ, bool WEIGHTED, int STAGES, bool CINIT = false, bool SQ = false>
__global__ __launch_bounds__(WM* WN* WZ * 64, 2) void gemm_tn_f64(PglGemmArgs g) {
    using C = Cfg<WM, WN, WZ, WEIGHTED, STAGES>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ntm = (g.M + C::BM - 1) / C::BM, ntn = (g.N + C::BN - 1) / C::BN;
    const long total = (long)(g.tri ? ntm * (ntm + 1) / 2 : ntm * ntn) * g.nbatch;
    const long chunk = (total + 7) / 8;
    long w = (long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((long)(blockIdx.x >> 3) >= chunk || w >= total) return;
    gemm_item<WM, WN, WZ, WEIGHTED, STAGES, false, CINIT, SQ>(g, w, smem);
}

// ---- persistent launch (long launches: the hardware dispatcher's round-robin drifts after a few hundred rounds and the
// co-residency above is lost -- measured 1.5x panel reuse instead of ~30x at cfg3).  One workgroup per CU; each reads the
// XCC it runs on and pulls consecutive items of that XCD's chunk from a counter, stealing from the next XCD when its own
// chunk is exhausted.  Placement is used for speed only: any placement gives the same result.
template <int WM, int WN, int WZ, bool WEIGHTED, int STAGES, bool DMA>
__global__ __launch_bounds__(WM* WN* WZ * 64, 2) void gemm_tn_f64_persistent(PglGemmArgs g) {
    using C = Cfg<WM, WN, WZ, WEIGHTED, STAGES>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    long* ticket = reinterpret_cast<long*>(smem + (size_t)STAGES * C::STAGE);
    const int ntm = (g.M + C::BM - 1) / C::BM, ntn = (g.N + C::BN - 1) / C::BN;
    const long total = (long)(g.tri ? ntm * (ntm + 1) / 2 : ntm * ntn) * g.nbatch;
    const long chunk = (total + 7) / 8;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            long w = -1;
            for (int hop = 0; hop < 8 && w < 0; ++hop) {
                const int y = (int)((xcc + hop) & 7u);
                const long lo = (long)y * chunk, hi = (lo + chunk < total) ? lo + chunk : total;
                if (lo >= hi) continue;
                const long it = atomicAdd(&g.sched[y], 1);
                if (lo + it < hi) w = lo + it;
            }
            ticket[0] = w;
        }
        __syncthreads();
        const long w = ticket[0];
        if (w < 0) break;
        gemm_item<WM, WN, WZ, WEIGHTED, STAGES, DMA>(g, w, smem);
    }
}

// ---- the Gram with INDEPENDENT workgroups on a SIMD ("fine" pipeline).  In the 8-wave Gram above the two waves of a SIMD belong to
// one workgroup: they reach every barrier and every DMA wait together, and the MFMA pipe idles while they do (rocprofv3:
// SQ_VALU_MFMA_BUSY_CYCLES = 90.6 % with 18 % of the wave cycles in s_waitcnt).  Here a workgroup is 4 waves (one neuron's 128 x 128
// tile, 32 x 128 per wave) and TWO workgroups share a CU, so the waves that alternate on a SIMD stall independently.  To fit twice
// in LDS the stages are 8 rows deep (one pair of k-steps), four of them: tile kt+3 is requested during pair kt and published by
// the barrier at the end of pair kt+1.  Same fragment/DMA slotting as the pipeline above; 5 DMA pieces per wave per pair.
constexpr int FK = 8, FST = 4, FSA = 128 + PAD;
constexpr int F_A = FK * FSA, F_B = FK * FSA, F_W = FK;
constexpr int F_STAGE = F_A + F_B + F_W;
constexpr size_t F_LDS = (size_t)FST * F_STAGE * sizeof(double) + 16;

__device__ __forceinline__ void gram_fine_item(const PglGemmArgs& g, const long w, double* smem) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int nz = g.nz_total;
    const int z = (int)(w % nz);
    const int S = g.ksplit > 1 ? g.ksplit : 1;
    const int sl = (int)((w / nz) % S), tile = (int)(w / nz / S);
    const int tm = isqrt_tri(tile), tn = tile - tm * (tm + 1) / 2;
    const int m0 = tm * 128, n0 = tn * 128;
    const long krow0 = S > 1 ? (long)sl * g.ks_rows : 0;                      // this item's slice of K: rows [krow0, krow0 + Ks)
    const int Ks = S > 1 ? (int)((long)g.K - krow0 < g.ks_rows ? (long)g.K - krow0 : g.ks_rows) : g.K;
    const int nkt = Ks / FK;
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;
    const int frow = lane >> 4, fcol = lane & 15;
    const double* __restrict__ Ab = g.A + krow0 * g.lda;
    const double* __restrict__ Bb = g.B + krow0 * g.ldb;

    const int wv = __builtin_amdgcn_readfirstlane(wq);
    const char* gp[5];
    long gstep[5], gadv[5];
    int loff[5];
    {
        int ca = m0 + lane * 2, cb = n0 + lane * 2;
        ca = ca < g.a_cols ? ca : g.a_cols - 2;
        cb = cb < g.b_cols ? cb : g.b_cols - 2;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = wv + 4 * (p >> 1);
            if ((p & 1) == 0) { gp[p] = reinterpret_cast<const char*>(Ab + (long)r * g.lda + ca); gstep[p] = (long)FK * g.lda * 8; loff[p] = r * FSA * 8; }
            else { gp[p] = reinterpret_cast<const char*>(Bb + (long)r * g.ldb + cb); gstep[p] = (long)FK * g.ldb * 8; loff[p] = (F_A + r * FSA) * 8; }
        }
        const int dw = wv * 4 + (lane & 3), r = dw >> 1;      // 8 weights = 16 dwords, 4 per wave
        gp[4] = reinterpret_cast<const char*>(g.W + (krow0 + r) * g.ldw + z) + 4 * (dw & 1);
        gstep[4] = (long)FK * g.ldw * 8;
        loff[4] = (F_A + F_B) * 8 + wv * 16;
    }
    auto advance = [&](bool adv) {
#pragma unroll
        for (int p = 0; p < 5; ++p) gadv[p] = adv ? gstep[p] : 0;
    };
    auto dma_piece = [&](int stage, int p) {
        char* dst = reinterpret_cast<char*>(smem) + (size_t)stage * F_STAGE * 8 + loff[p];
        if (p < 4) __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[p], (lds_ptr_t)dst, 16, 0, 0);
        else if (lane < 4) __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[4], (lds_ptr_t)dst, 4, 0, 0);
        gp[p] += gadv[p];
    };
    auto dma = [&](int stage) {
#pragma unroll
        for (int p = 0; p < 5; ++p) dma_piece(stage, p);
    };

    d4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

    advance(1 < nkt); dma(0);
    advance(2 < nkt); dma(1);
    advance(3 < nkt); dma(2);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // tiles 0 and 1 have landed
    block_sync_lds();

    double pa[2][2][2], pb[2][2][8], pw[2][2];       // [set][step in pair][fragment]
    auto rdA = [&](int buf, int kk, int i) { return smem[buf * F_STAGE + wq * 32 + fcol + (kk * 4 + frow) * FSA + i * 16]; };
    auto rdB = [&](int buf, int kk, int j) { return smem[buf * F_STAGE + F_A + fcol + (kk * 4 + frow) * FSA + j * 16]; };
    auto rdW = [&](int buf, int kk) { return smem[buf * F_STAGE + F_A + F_B + kk * 4 + frow]; };
#pragma unroll
    for (int i = 0; i < 2; ++i) { pa[0][0][i] = rdA(0, 0, i); pa[0][1][i] = rdA(0, 1, i); }
#pragma unroll
    for (int j = 0; j < 8; ++j) pb[0][0][j] = rdB(0, 0, j);
    pw[0][0] = rdW(0, 0); pw[0][1] = rdW(0, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) { pa[0][0][i] *= pw[0][0]; pa[0][1][i] *= pw[0][1]; }

    int cur = 0;
    for (int kt0 = 0; kt0 < nkt; kt0 += 2) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {                 // two pairs per trip so that the register sets alternate statically
            const int kt = kt0 + p;
            if (kt >= nkt) break;
            const int ps = p, ns = p ^ 1;
            const int nxt = (cur + 1) & (FST - 1), dstage = (cur + 3) & (FST - 1);
            advance(kt + 4 < nkt);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int i = m >> 3, j = m & 7;
                acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ps][0][i], pb[ps][0][j], acc[m >> 2][m & 3], 0, 0, 0);
                if (m < 8) pb[ps][1][m] = rdB(cur, 1, m);
                else if (m < 10) pa[ns][0][m - 8] = rdA(nxt, 0, m - 8);
                else if (m == 10) pw[ns][0] = rdW(nxt, 0);
                else if (m == 11) pw[ns][1] = rdW(nxt, 1);
                else dma_piece(dstage, m - 12);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int i = m >> 3, j = m & 7;
                acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[ps][1][i], pb[ps][1][j], acc[m >> 2][m & 3], 0, 0, 0);
                if (m < 2) pa[ns][1][m] = rdA(nxt, 1, m);
                else if (m < 10) pb[ns][0][m - 2] = rdB(nxt, 0, m - 2);
                else if (m == 10) dma_piece(dstage, 4);
                else if (m == 13) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) { pa[ns][0][q] *= pw[ns][0]; pa[ns][1][q] *= pw[ns][1]; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // publish tile kt+2 (requested two pairs ago); this pair's 5 requests may stay in flight (past the end of K they
            // re-request the last tile into a stage that is not read any more)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            block_sync_lds();
            cur = nxt;
        }
    }

    double* __restrict__ Cb = S > 1 ? g.Cpart + (long)z * g.part_stride_z + (long)sl * g.part_stride_s : g.C + (long)z * g.strideC;
    const double alpha = S > 1 ? 1.0 : g.alpha, beta = S > 1 ? 0.0 : g.beta;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double cv[4][4];
        if (beta != 0.0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 4 * i + j;
                    const int row = m0 + wq * 32 + (f >> 3) * 16 + frow + 4 * r, col = n0 + (f & 7) * 16 + fcol;
                    cv[r][j] = (row < g.M && col < g.N) ? Cb[(long)row * g.ldc + col] : 0.0;
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 4 * i + j;
                const int row = m0 + wq * 32 + (f >> 3) * 16 + frow + 4 * r, col = n0 + (f & 7) * 16 + fcol;
                if (row >= g.M || col >= g.N) continue;
                double v = alpha * acc[i][j][r];
                if (beta != 0.0) v += beta * cv[r][j];
                Cb[(long)row * g.ldc + col] = v;
            }
    }
}

__global__ __launch_bounds__(256, 2) void gram_fine_persistent(PglGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    long* ticket = reinterpret_cast<long*>(smem + (size_t)FST * F_STAGE);
    const int ntm = (g.M + 127) / 128;
    const long total = (long)ntm * (ntm + 1) / 2 * g.nz_total * (g.ksplit > 1 ? g.ksplit : 1);
    const long chunk = (total + 7) / 8;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            long w = -1;
            for (int hop = 0; hop < 8 && w < 0; ++hop) {
                const int y = (int)((xcc + hop) & 7u);
                const long lo = (long)y * chunk, hi = (lo + chunk < total) ? lo + chunk : total;
                if (lo >= hi) continue;
                const long it = atomicAdd(&g.sched[y], 1);
                if (lo + it < hi) w = lo + it;
            }
            ticket[0] = w;
        }
        __syncthreads();
        const long w = ticket[0];
        if (w < 0) break;
        gram_fine_item(g, w, smem);
    }
}

template <int WM, int WN, int WZ, bool WEIGHTED, int STAGES = 2, bool CINIT = false, bool SQ = false>
int launch(const PglGemmArgs& a, hipStream_t st) {
    using C = Cfg<WM, WN, WZ, WEIGHTED, STAGES>;
    if constexpr (!WEIGHTED && STAGES == 2 && !CINIT) {
        if (a.beta == 1.0 && (a.alpha == 1.0 || a.alpha == -1.0)) return launch<WM, WN, WZ, WEIGHTED, STAGES, true, SQ>(a, st);
    }
    static PglPerDevice attr_set;
    auto kern = gemm_tn_f64<WM, WN, WZ, WEIGHTED, STAGES, CINIT, SQ>;
    if (int rc = pgl_set_dynamic_lds(reinterpret_cast<const void*>(kern), C::LDS_BYTES, attr_set)) return rc;
    const int ntm = (a.M + C::BM - 1) / C::BM, ntn = (a.N + C::BN - 1) / C::BN;
    const long ntiles = a.tri ? (long)ntm * (ntm + 1) / 2 : (long)ntm * ntn;
    const long total = ntiles * a.nbatch;
    if (total <= 0) return PGL_OK;
    const long chunk = (total + 7) / 8;
    const long grid = chunk * 8;
    if (grid > 0x7fffffffL) { pgl_set_error("gemm grid too large: %ld", grid); return PGL_ERR_ARG; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::THREADS), C::LDS_BYTES, st, a);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

// scheduler scratch of the persistent launches: per device a ring of 8-counter slots (32 B each, 8 KiB in all), allocated on first use
// on that device.  (The only device memory the library itself owns -- documented in include/pyglm_hip.h; every data buffer belongs to
// the caller.)
static int* sched_slot(hipStream_t st) {
    static std::atomic<int*> bases[PGL_MAX_DEVICES];
    static std::atomic<unsigned> next{0};
    constexpr unsigned SLOTS = 256;
    std::atomic<int*>& b = bases[pgl_device() & (PGL_MAX_DEVICES - 1)];
    int* base = b.load(std::memory_order_acquire);
    if (!base) {
        int* fresh = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&fresh), SLOTS * 8 * sizeof(int)) != hipSuccess) return nullptr;
        if (b.compare_exchange_strong(base, fresh, std::memory_order_acq_rel)) base = fresh;
        else (void)hipFree(fresh);                     // another host thread got there first
    }
    int* slot = base + (size_t)(next.fetch_add(1, std::memory_order_relaxed) % SLOTS) * 8;
    if (hipMemsetAsync(slot, 0, 8 * sizeof(int), st) != hipSuccess) return nullptr;
    return slot;
}

template <int WM, int WN, int WZ, bool WEIGHTED, int STAGES, bool DMA>
int launch_persistent(const PglGemmArgs& a0, hipStream_t st) {
    using C = Cfg<WM, WN, WZ, WEIGHTED, STAGES>;
    static PglPerDevice attr_set;
    auto kern = gemm_tn_f64_persistent<WM, WN, WZ, WEIGHTED, STAGES, DMA>;
    constexpr size_t lds = C::LDS_BYTES + 16;     // + the work ticket
    if (int rc = pgl_set_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_set)) return rc;
    const int n_cu = pgl_device_cus(pgl_device());
    const int ntm = (a0.M + C::BM - 1) / C::BM, ntn = (a0.N + C::BN - 1) / C::BN;
    const long total = (a0.tri ? (long)ntm * (ntm + 1) / 2 : (long)ntm * ntn) * a0.nbatch;
    if (total <= 0) return PGL_OK;
    PglGemmArgs a = a0;
    a.sched = sched_slot(st);
    if (!a.sched) { pgl_set_error("persistent gemm: scheduler scratch unavailable"); return PGL_ERR_HIP; }
    const long per_cu = (lds * 2 <= 160 * 1024) ? 2 : 1;
    const long grid = total < n_cu * per_cu ? total : n_cu * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::THREADS), lds, st, a);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

static int launch_gram_fine(const PglGemmArgs& a0, hipStream_t st) {
    static PglPerDevice attr_set;
    if (int rc = pgl_set_dynamic_lds(reinterpret_cast<const void*>(gram_fine_persistent), F_LDS, attr_set)) return rc;
    const int n_cu = pgl_device_cus(pgl_device());
    const int ntm = (a0.M + 127) / 128;
    const long total = (long)ntm * (ntm + 1) / 2 * a0.nz_total * (a0.ksplit > 1 ? a0.ksplit : 1);
    if (total <= 0) return PGL_OK;
    PglGemmArgs a = a0;
    a.sched = sched_slot(st);
    if (!a.sched) { pgl_set_error("persistent gemm: scheduler scratch unavailable"); return PGL_ERR_HIP; }
    const long grid = total < 2L * n_cu ? total : 2L * n_cu;
    hipLaunchKernelGGL(gram_fine_persistent, dim3((unsigned)grid), dim3(256), F_LDS, st, a);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}

}  // namespace

int* pgl_sched_slot(hipStream_t st) { return sched_slot(st); }

namespace {
// C[z] (+)= part[z][0] + part[z][1] + ... in slice order, the lower 128 x 128 TILES of the M x M block: the split Gram writes only tiles on and
// below the diagonal (tri = 1), so the tiles above it hold whatever the borrowed scratch held before -- they are neither read nor written here
// (the unsplit kernel leaves J above the diagonal tiles untouched as well)
__global__ __launch_bounds__(256) void sum_k_slices_kernel(const double* __restrict__ part, long stride_z, long stride_s, int S, double* __restrict__ C,
                                                           long strideC, long ldc, int M, int accumulate) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)M * M) return;
    const int r = (int)(e / M), c = (int)(e % M), z = blockIdx.y;
    if (r / 128 < c / 128) return;
    const double* p = part + (long)z * stride_z + (long)r * ldc + c;
    double* out = C + (long)z * strideC + (long)r * ldc + c;
    double acc = accumulate ? *out : 0.0;
    for (int s = 0; s < S; ++s) acc += p[(long)s * stride_s];
    *out = acc;
}
}  // namespace

int pgl_gram_split(const double* X, long ldx, int x_cols, const double* W, long ldw, int Tp, int D, int nz, double* J, long ldj, long strideJ, int accumulate,
                   long ks_rows, double* part, long part_stride_z, hipStream_t st) {
    PGL_CHECK_ARG(X && W && J && part && Tp > 0 && Tp % 16 == 0 && D > 0 && D <= 512 && nz > 0 && ldj >= D && ldw >= nz && ks_rows >= 16 && ks_rows % 16 == 0);
    const int S = (int)((Tp + ks_rows - 1) / ks_rows);
    PGL_CHECK_ARG(S >= 2 && part_stride_z >= (long)S * ldj * ldj);
    PglGemmArgs a{};
    a.A = X; a.lda = ldx; a.B = X; a.ldb = ldx;
    a.C = J; a.ldc = ldj; a.strideC = strideJ;
    a.W = W; a.ldw = ldw;
    a.M = D; a.N = D; a.K = Tp;
    a.a_cols = x_cols & ~1; a.b_cols = x_cols & ~1;
    a.nbatch = (nz + 1) / 2; a.nz_total = nz;
    a.alpha = 1.0; a.beta = 0.0; a.tri = 1;
    a.ksplit = S; a.ks_rows = ks_rows; a.Cpart = part; a.part_stride_z = part_stride_z; a.part_stride_s = ldj * ldj;
    PGL_CHECK_ARG(((uintptr_t)a.A % 16) == 0 && a.lda % 2 == 0);
    if (int rc = launch_gram_fine(a, st)) return rc;
    hipLaunchKernelGGL(sum_k_slices_kernel, dim3((unsigned)(((long)D * D + 255) / 256), nz), dim3(256), 0, st, part, part_stride_z, a.part_stride_s, S, J, strideJ, ldj, D,
                       accumulate);
    PGL_CHECK_LAUNCH();
    return PGL_OK;
}


int pgl_launch_gemm(PglGemmKind kind, const PglGemmArgs& a, hipStream_t st) {
    PGL_CHECK_ARG(a.K % BK == 0 && a.M > 0 && a.N > 0 && a.nbatch > 0);
    PGL_CHECK_ARG(a.a_cols % 2 == 0 && a.b_cols % 2 == 0 && a.lda % 2 == 0 && a.ldb % 2 == 0);
    PGL_CHECK_ARG(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.B % 16) == 0);
    if (a.pipe && kind != PGL_GEMM_GRAM2) {
        // products marked `pipe` (the rank-k updates of the flips and the Cholesky, W = G U) may take the update pipeline of pgl_update.hip.
        // Measured at cfg3 (tools/ubench_update.hip, same bits on both kernels): the write-only form (beta = 0: W = G U) is 8-23 % faster
        // there; the read-modify-write forms are not (one persistent workgroup per CU cannot hide a tile's 512 KiB of C traffic behind
        // another workgroup's MFMAs: 62.0 vs 62.2 TFLOP/s at rank 512, 0.77x at rank 64) and stay on the generic kernel.
        // (pipe = 2, the C ABI's `kernel` argument, forces the pipeline: tests/test_gpu_update.py compares the two kernels bit for bit)
        if ((a.pipe == 2 || a.beta == 0.0) && pgl_update_supported(a)) return pgl_launch_update(a, st);
    }
    switch (kind) {
        case PGL_GEMM_GRAM2: PGL_CHECK_ARG(a.W != nullptr && a.tri == 1 && a.M == a.N && a.batch_dim == nullptr); {
            // persistent, DMA-staged 3-stage pipeline
            // launches that cannot give every CU two 8-wave items (few neurons: small models, thin shards) run as 4-wave workgroups,
            // one neuron each, two per CU: twice the items, and 53 vs 29 TFLOP/s at D = 640 with 16 neurons; at full size the
            // 8-wave pipeline (two neurons share every staged X tile) is 5 % faster
            const int n_cu = pgl_device_cus(pgl_device());
            const long ntm = (a.M + 127) / 128;
            if (ntm * (ntm + 1) / 2 * a.nbatch < 2L * n_cu) return launch_gram_fine(a, st);
            return launch_persistent<2, 2, 2, true, 3, true>(a, st);
        }
        case PGL_GEMM_PLAIN:
            PGL_CHECK_ARG(a.tri == 0);
            if (a.N <= 128) return launch<2, 2, 1, false>(a, st);     // (the 128 x 128 blocks of the pivot-block inversion: no half-empty 256-wide tile)
            // products with at most 64 output rows -- the row-panel solves and strip updates of the Cholesky (64-row sub-panels against a remainder
            // thousands of columns wide) -- on 64 x 256 tiles: the 128-row tile ran half empty there (same multiply-adds per element in the
            // same order: same bits)
            if (a.M <= 64) return launch<1, 4, 1, false>(a, st);
            return launch<2, 4, 1, false>(a, st);
        case PGL_GEMM_SQUARES:
            PGL_CHECK_ARG(a.tri == 0 && !a.pipe);
            if (a.M <= 64) return launch<1, 4, 1, false, 2, false, true>(a, st);      // few neurons per batch (BASELINE configs[4]: 4): 64-row tiles
            return launch<2, 4, 1, false, 2, false, true>(a, st);
        case PGL_GEMM_TRI1: {
            PGL_CHECK_ARG(a.M == a.N);
            // a lower triangle whose last few rows would open a tile row of their own (a tableau: D + 2 rows, 41 of 861 tiles for the bias and
            // potential rows at cfg3): those rows through the skinny kernel of pgl_update.hip (same bits), the tiles on the rest
            const int extra = a.M % 128;
            if (a.pipe && a.tri == 1 && a.batch_dim == nullptr && a.M > 128 && extra >= 1 && extra <= 16 && pgl_update_supported(a)) {
                if (int rc = pgl_launch_update_rows(a, a.M - extra, extra, st)) return rc;
                PglGemmArgs b = a;
                b.M = b.N = a.M - extra;
                return launch<2, 2, 1, false>(b, st);
            }
            return launch<2, 2, 1, false>(a, st);
        }
    }
    pgl_set_error("unknown gemm kind %d", (int)kind);
    return PGL_ERR_ARG;
}
