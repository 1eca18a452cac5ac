// Prototype for DESIGN.md section 9 item 1 (NOT part of the library): the int8 building block of a modular (Ozaki-II style) Gram,
//     C_q[i][j] = sum_t A_q[i][t] * B_q[j][t]   (int8 residue planes, K = t contiguous, int32 accumulation, lower 256 x 256 tiles)
// on v_mfma_i32_32x32x32_i8.  Checks itself against a CPU product at a small size, then times the cfg3 shape (D = 5120, K = 100096)
// for a few planes and prints the rate.
//     hipcc -O3 --offload-arch=gfx950 tools/i8gram.hip -o tools/bin/i8gram
// Layout: workgroup = 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 (4 x 2 accumulators of 32 x 32), workgroup tile 256 x 256;
// K tiles of 64 bytes staged by global_load_lds (1 KiB = 16 rows per wave-instruction) into 3 LDS stages; the 16-byte chunks of a row
// are XOR-swizzled with (row >> 2) & 3 so that ds_read_b128 fragment reads are bank-conflict free without padding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

#ifndef NST_
#define NST_ 4
#endif
constexpr int TM = 256, TN = 256, BKB = 64, NST = NST_;
constexpr int STAGE_BYTES = (TM + TN) * BKB;          // 32 KiB
constexpr int LDS_BYTES = NST * STAGE_BYTES;
constexpr int PLDS = NST == 5 ? LDS_BYTES : LDS_BYTES + 16;

struct Args {
    const int8_t* A; const int8_t* B;   // [Q][D][ldk]
    int32_t* C;                          // [Q][D][D]
    int D, K; long ldk; int Q;
};

__device__ __forceinline__ int isqrt_tri(int t) {
    int r = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)(r + 1) * (r + 2) / 2 <= t) ++r;
    while ((long)r * (r + 1) / 2 > t) --r;
    return r;
}

// ABL (timing-only ablations, results invalid): 1 = no DMA requests in the loop, 2 = no barrier / vmcnt wait, 4 = no LDS fragment reads,
// 8 = every item stages tile (0, 0) (all requests hit L2)
// tile t of a plane's lower triangle in CLUSTERED order: super-blocks of SB x SB tiles, row by row (boustrophedon), so that any run
// of ~32 consecutive tiles touches few distinct 256-row strips (they are shared through the XCD's L2)
#ifndef SB_
#define SB_ 6
#endif
constexpr int SB = SB_;
__device__ __forceinline__ void clustered_tile(int t, int ntm, int& tm, int& tn) {
    const int nsb = (ntm + SB - 1) / SB;
    for (int I = 0; I < nsb; ++I) {
        const int r0 = I * SB, nr = min(SB, ntm - r0);
        for (int jj = 0; jj <= I; ++jj) {
            const int J = (I & 1) ? I - jj : jj;
            const int c0 = J * SB, nc = min(SB, ntm - c0);
            const int cnt = I == J ? nr * (nr + 1) / 2 : nr * nc;
            if (t < cnt) {
                if (I == J) { const int r = isqrt_tri(t); tm = r0 + r; tn = c0 + t - r * (r + 1) / 2; }
                else { tm = r0 + t / nc; tn = c0 + t % nc; }
                return;
            }
            t -= cnt;
        }
    }
    tm = tn = 0;
}

// VAR: 1 = scalar request cursors + one shared 32-bit lane offset (saddr form), 2 = requests spread over both k-steps
template <int ABL, int VAR>
__device__ __forceinline__ void i8gram_item(const Args& g, const int q, const int tm, const int tn, char* lds) {
    const int m0 = tm * TM, n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;            // 2 x 4 waves
    const int8_t* A = g.A + (long)q * g.D * g.ldk;
    const int8_t* B = g.B + (long)q * g.D * g.ldk;
    const int nkt = g.K / BKB;

    // ---- DMA: per K tile 512 rows x 64 B = 32 requests of 1 KiB (16 rows); wave w issues requests w, w+8, w+16, w+24
    // request rq covers rows 16 rq .. 16 rq + 15 of the stacked [A rows 0..255 | B rows 0..255]; lane l -> row 16 rq + l / 4,
    // physical chunk l % 4, which holds logical chunk (l % 4) ^ ((row >> 2) & 3)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    constexpr long KADV = (VAR & 4) ? 1024 : BKB;
    const char* gp[4];
    int loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rq = wv + 8 * i, row = 16 * rq + (lane >> 2);      // 0..511
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int8_t* base = (ABL & 8) ? (row < TM ? A + (long)row * g.ldk : B + (long)(row - TM) * g.ldk) : (row < TM ? A + (long)(m0 + row) * g.ldk : B + (long)(n0 + row - TM) * g.ldk);
        gp[i] = (VAR & 4) ? reinterpret_cast<const char*>(base) - (long)(lane >> 2) * g.ldk + (lane >> 2) * 64 + lc * 16 : reinterpret_cast<const char*>(base) + lc * 16;
        loff[i] = rq * 1024;
    }
    // scalar form: request i starts at a wave-uniform row block; the lane part (row-in-block * ldk + swizzled chunk) is the same for all
    // VAR & 4: BLOCKED plane layout [row / 16][K tile][row % 16][64 B]: a request is one contiguous KiB, a row block's K tiles are consecutive
    const char* sp[4];
    const unsigned voff = (VAR & 4) ? (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16))
                                    : (unsigned)((lane >> 2) * g.ldk + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rq = wv + 8 * i;
        const int r0 = (ABL & 8) ? 0 : (i < 2 ? m0 : n0);
        sp[i] = reinterpret_cast<const char*>(i < 2 ? A + (long)(r0 + 16 * rq) * g.ldk : B + (long)(r0 + 16 * rq - TM) * g.ldk);
    }
    // the request in scalar-base form, written out: the compiler otherwise folds base + lane offset back into per-lane 64-bit cursors
    auto dma_s = [&](const char* sbase, char* ldst) {
        const unsigned la = (unsigned)(uintptr_t)(lds_ptr_t)ldst;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la), "v"(voff), "s"(sbase) : "memory");
    };
    auto dma = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (VAR & 1) { dma_s(sp[i], lds + stage * STAGE_BYTES + loff[i]); sp[i] += KADV; }
            else { __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[i], (lds_ptr_t)(lds + stage * STAGE_BYTES + loff[i]), 16, 0, 0); gp[i] += KADV; }
        }
    };

    v16i acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // fragment addresses: A operand: row = wm*128 + i*32 + (lane & 31), k chunk = kk*2 + (lane >> 5); B likewise with wn*64 + j*32
    const int fr = lane & 31, fk = lane >> 5;
    auto frag = [&](int stage, int row, int kk) {
        const int pc = (kk * 2 + fk) ^ ((row >> 2) & 3);
        return *reinterpret_cast<const v4i*>(lds + stage * STAGE_BYTES + row * BKB + pc * 16);
    };

    // Pipeline: NST - 1 = 3 tiles requested ahead; fragments double-buffered in registers, every LDS read and DMA request issued in
    // the shadow of an MFMA (order pinned with sched_barrier); the barrier sits in the MIDDLE of a K tile (after its first k-step):
    // it publishes tile kt+1, whose first fragments are fetched during the second k-step, and frees the stage of tile kt-1 for the
    // requests of tile kt+3.  Bare s_barrier: a __syncthreads() would drain the outstanding requests (vmcnt(0)).
    long gadv = KADV;                 // 0 once the last tile has been requested: the cursors stop and the last tile is requested again
    long gadv2 = KADV;                // the same for the pieces requested one k-step later (VAR & 2)
    auto dma_piece = [&](int stage, int i, long adv) {      // (into a stage nobody reads any more) -- no branch in front of a request
        if constexpr (VAR & 1) { dma_s(sp[i], lds + stage * STAGE_BYTES + loff[i]); sp[i] += adv; }
        else { __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[i], (lds_ptr_t)(lds + stage * STAGE_BYTES + loff[i]), 16, 0, 0); gp[i] += adv; }
    };
    if constexpr ((VAR & 2) != 0) {          // needs nkt >= 4: tiles 0, 1 and half of tile 2; the other half follows in the first k-step
        for (int p = 0; p < NST - 2; ++p) dma(p);
        dma_piece(NST - 2, 0, KADV); dma_piece(NST - 2, 1, KADV);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (NST - 3) + 2) : "memory");
    } else {
        for (int p = 0; p < NST - 1; ++p) if (p < nkt) dma(p);
        if (nkt >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (NST - 2)) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v4i F[2][6];
    auto rd = [&](int stage, int kk, int f) {
        return f < 4 ? frag(stage, wm * 128 + f * 32 + fr, kk) : frag(stage, TM + wn * 64 + (f - 4) * 32 + fr, kk);
    };
#pragma unroll
    for (int f = 0; f < 6; ++f) F[0][f] = rd(0, 0, f);
    if (ABL & 4)
#pragma unroll
        for (int f = 0; f < 6; ++f) F[1][f] = rd(0, 1, f);
    int cur = 0;
    auto kloop = [&](auto dead_c) {
    constexpr bool DEAD = decltype(dead_c)::value;
    for (int kt = 0; kt < nkt; ++kt) {
        const int nxt = cur == NST - 1 ? 0 : cur + 1;
        const int dst = cur == 0 ? NST - 1 : cur - 1;          // the stage of tile kt-1 takes tile kt+NST-1
        const int dst2 = cur >= 2 ? cur - 2 : cur + NST - 2;                               // (VAR & 2) second half of tile kt+2 -> the stage of tile kt-2
        if (kt + NST >= nkt) gadv = 0;
        if (kt + NST - 1 >= nkt) gadv2 = 0;
        // ---- first k-step (set 0): prefetch the fragments of this tile's second k-step
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (!(ABL & 16) && !DEAD) { const int bj = (VAR & 8) ? ((m & 1) ^ ((m >> 1) & 1)) : (m & 1); acc[m >> 1][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[0][m >> 1], F[0][4 + bj], acc[m >> 1][bj], 0, 0, 0); }
            if (m < 6 && !(ABL & 4) && !DEAD) F[1][m] = rd(cur, 1, m);
            if constexpr ((VAR & 2) != 0) { if ((m == 3 || m == 7) && !(ABL & 1)) dma_piece(dst2, 2 + (m >> 2), gadv2); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (NST - 3)) : "memory");       // tile kt+1 has landed; the 4 requests of tile kt+2 may stay in flight
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        // ---- second k-step (set 1): prefetch the first fragments of tile kt+1, request tile kt+3
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (!(ABL & 16) && !DEAD) { const int bj = (VAR & 8) ? ((m & 1) ^ ((m >> 1) & 1)) : (m & 1); acc[m >> 1][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[1][m >> 1], F[1][4 + bj], acc[m >> 1][bj], 0, 0, 0); }
            if (m < 6 && !(ABL & 4) && !DEAD) F[0][m] = rd(nxt, 0, m);
            if constexpr ((VAR & 2) != 0) { if ((m == 3 || m == 7) && !(ABL & 1)) dma_piece(dst, m >> 2, gadv); }
            else { if ((m & 1) && !(ABL & 1)) dma_piece(dst, m >> 1, gadv); }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = nxt;
    }
    };
    // VAR & 16: on a diagonal tile the wave tiles with rows 0..127 and columns 128..255 lie above the diagonal: those waves only move data
    if ((VAR & 16) && tm == tn && wm == 0 && wn >= 2) kloop(std::true_type{}); else kloop(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-requested last tiles must have landed before the LDS is reused
    // epilogue: C/D layout of 32x32 i32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    int32_t* C = g.C + (long)q * g.D * g.D;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                C[(long)row * g.D + col] = acc[i][j][r];
            }
}

// ---- the same item on v_mfma_i32_16x16x64_i8 (VAR & 32): 8 x 4 accumulators of 16 x 16 per wave, ONE k-step per 64-byte K tile.  On random
// operand bytes this instruction sustains 4.0 POP/s at the package power limit where the 32x32x32 form sustains 3.45 (tools/ubench_i8.hip).
// A/B fragment of 16 rows x 64 B: lane l -> row l % 16, 16-byte chunk l / 16.  Chunk swizzle c ^ g(row >> 2 & 3), g = {0, 2, 3, 1}:
// conflict-free for the 4 x 16 lane groups of ds_read_b128.  Fragment registers: A 8, B 2 x 4 (double-buffered): the second half of a
// tile (A4-7) prefetches the next tile's A0-3 and B, the first half (A0-3) fetches its own A4-7.
template <int ABL, int VAR>
__device__ __forceinline__ void i8gram_item16(const Args& g, const int q, const int tm, const int tn, char* lds) {
    const int m0 = tm * TM, n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int8_t* A = g.A + (long)q * g.D * g.ldk;
    const int8_t* B = g.B + (long)q * g.D * g.ldk;
    const int nkt = g.K / BKB;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto gsw = [](int qd) { return (0x1320 >> (4 * qd)) & 3; };          // g = {0, 2, 3, 1}
    const char* gp[4];
    int loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rq = wv + 8 * i;
        const int lc = (lane & 3) ^ gsw((lane >> 4) & 3);
        const int8_t* base = (ABL & 8) ? (i < 2 ? A + (long)(16 * rq) * g.ldk : B + (long)(16 * rq - TM) * g.ldk) : (i < 2 ? A + (long)(m0 + 16 * rq) * g.ldk : B + (long)(n0 + 16 * rq - TM) * g.ldk);
        gp[i] = reinterpret_cast<const char*>(base) + (lane >> 2) * 64 + lc * 16;
        loff[i] = rq * 1024;
    }
    v4i acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    const int fr = lane & 15, fc = lane >> 4;
    auto frag = [&](int stage, int row) {
        const int pc = fc ^ gsw((row >> 2) & 3);
        return *reinterpret_cast<const v4i*>(lds + stage * STAGE_BYTES + row * BKB + pc * 16);
    };
    auto rdA = [&](int stage, int f) { return frag(stage, wm * 128 + f * 16 + fr); };
    auto rdB = [&](int stage, int f) { return frag(stage, TM + wn * 64 + f * 16 + fr); };
    long gadv = 1024, gadv2 = 1024;
    auto dma_piece = [&](int stage, int i, long adv) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)gp[i], (lds_ptr_t)(lds + stage * STAGE_BYTES + loff[i]), 16, 0, 0);
        gp[i] += adv;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(0, i, 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(1, i, 1024);
    dma_piece(2, 0, 1024);
    dma_piece(2, 1, 1024);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v4i FA[8], FB[2][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { FA[f] = rdA(0, f); FB[0][f] = rdB(0, f); }
    if (ABL & 4)
#pragma unroll
        for (int f = 0; f < 4; ++f) { FA[4 + f] = rdA(0, 4 + f); FB[1][f] = rdB(1, f); }
    int cur = 0;
    // two tiles per iteration so that the B buffers alternate with compile-time indices
    // VAR & 64: on a diagonal tile the 16 x 16 blocks that lie entirely above the diagonal are not multiplied (47 % of such a tile)
    unsigned skip1 = 0, skip2 = 0;
    if ((VAR & (64 | 256)) && tm == tn)
        for (int m = 0; m < 16; ++m) {
            if (wm * 8 + (m >> 2) < wn * 4 + (m & 3)) skip1 |= 1u << m;
            if (wm * 8 + 4 + (m >> 2) < wn * 4 + (m & 3)) skip2 |= 1u << m;
        }
    skip1 = __builtin_amdgcn_readfirstlane(skip1); skip2 = __builtin_amdgcn_readfirstlane(skip2);
    auto tile = [&](auto pb_c, auto dg_c, const int kt) {
        constexpr int PB = decltype(pb_c)::value;
        constexpr bool DG = decltype(dg_c)::value;
        const int nxt = cur == NST - 1 ? 0 : cur + 1;
        const int dst = cur == 0 ? NST - 1 : cur - 1;
        const int dst2 = cur ^ 2;
        if (kt + NST >= nkt) gadv = 0;
        if (kt + NST - 1 >= nkt) gadv2 = 0;
        // ---- first half: A0-3 x B; fetch this tile's A4-7
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int ai = m >> 2, bj = m & 3;
            if (!DG || !((skip1 >> m) & 1)) acc[ai][bj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(FA[ai], FB[PB][bj], acc[ai][bj], 0, 0, 0);
            if (m < 4 && !(ABL & 4)) FA[4 + m] = rdA(cur, 4 + m);
            if ((m == 7 || m == 15) && !(ABL & 1)) dma_piece(dst2, 2 + (m >> 3), gadv2);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- second half: A4-7 x B; prefetch the next tile's A0-3 and B
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int ai = 4 + (m >> 2), bj = m & 3;
            if (!DG || !((skip2 >> m) & 1)) acc[ai][bj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(FA[ai], FB[PB][bj], acc[ai][bj], 0, 0, 0);
            if (m < 4 && !(ABL & 4)) FB[PB ^ 1][m] = rdB(nxt, m);
            else if (m < 8 && !(ABL & 4)) FA[m - 4] = rdA(nxt, m - 4);
            if ((m == 7 || m == 15) && !(ABL & 1)) dma_piece(dst, m >> 3, gadv);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = nxt;
    };
    int kt = 0;
    if constexpr ((VAR & 256) != 0) {          // a launch of diagonal tiles only: one code path, the skipping one
        for (; kt + 1 < nkt; kt += 2) { tile(std::integral_constant<int, 0>{}, std::true_type{}, kt); tile(std::integral_constant<int, 1>{}, std::true_type{}, kt + 1); }
        if (kt < nkt) tile(std::integral_constant<int, 0>{}, std::true_type{}, kt);
    } else if ((VAR & 64) && tm == tn) {
        for (; kt + 1 < nkt; kt += 2) { tile(std::integral_constant<int, 0>{}, std::true_type{}, kt); tile(std::integral_constant<int, 1>{}, std::true_type{}, kt + 1); }
        if (kt < nkt) tile(std::integral_constant<int, 0>{}, std::true_type{}, kt);
    } else {
        for (; kt + 1 < nkt; kt += 2) { tile(std::integral_constant<int, 0>{}, std::false_type{}, kt); tile(std::integral_constant<int, 1>{}, std::false_type{}, kt + 1); }
        if (kt < nkt) tile(std::integral_constant<int, 0>{}, std::false_type{}, kt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue: C/D layout of 16x16 i32: col = lane & 15, row = 4 (lane >> 4) + reg
    int32_t* C = g.C + (long)q * g.D * g.D;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 128 + i * 16 + 4 * (lane >> 4) + r;
                const int col = n0 + wn * 64 + j * 16 + (lane & 15);
                C[(long)row * g.D + col] = acc[i][j][r];
            }
}

template <int ABL, int VAR>
__global__ __launch_bounds__(512) void i8gram_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int ntm = g.D / TM;
    const int ntiles = ntm * (ntm + 1) / 2;
    const int q = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    const int tm = isqrt_tri(tile), tn = tile - tm * (tm + 1) / 2;
    i8gram_item<ABL, VAR>(g, q, tm, tn, lds);
}

// persistent: one workgroup per CU pulls items from its XCD's list.  Items in clustered order are cut into chunks of CH; chunk c
// belongs to XCD c % 8, so the workgroups of an XCD work through one chunk together.
constexpr int CH = 32;
template <int ABL, int VAR>
__global__ __launch_bounds__(512) void i8gram_persistent(Args g, int* sched, const int* lists, const int* lens, int maxlen) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int* ticket = reinterpret_cast<int*>(lds + (NST == 5 ? LDS_BYTES - 16 : LDS_BYTES));   // 5 stages fill the LDS: the ticket borrows the tail of the last stage between items
    const int ntm = g.D / TM;
    const int ntiles = ntm * (ntm + 1) / 2;
    const int total = ntiles * g.Q;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int w = -1;
            int packed = -1;
            for (int hop = 0; hop < 8 && w < 0; ++hop) {
                const int y = (int)((xcc + hop) & 7u);
                const int it = atomicAdd(&sched[y], 1);
                if constexpr ((VAR & 128) != 0) { if (it < lens[y]) { packed = lists[(long)y * maxlen + it]; w = 1; } }
                else {
                    const int c = (it / CH) * 8 + y, cand = c * CH + it % CH;
                    if (cand < total) w = cand;
                }
            }
            if ((VAR & 128) && w >= 0) { ticket[1] = packed >> 16; ticket[2] = (packed >> 8) & 255; ticket[3] = packed & 255; }
            else if (w >= 0) {
                int tm, tn;
                clustered_tile(w % ntiles, ntm, tm, tn);
                ticket[1] = w / ntiles; ticket[2] = tm; ticket[3] = tn;
            }
            ticket[0] = w;
        }
        __syncthreads();
        if (ticket[0] < 0) break;
        const int q = __builtin_amdgcn_readfirstlane(ticket[1]), tm = __builtin_amdgcn_readfirstlane(ticket[2]), tn = __builtin_amdgcn_readfirstlane(ticket[3]);
        if constexpr ((VAR & 32) != 0) i8gram_item16<ABL, VAR>(g, q, tm, tn, lds); else i8gram_item<ABL, VAR>(g, q, tm, tn, lds);
    }
}

__global__ void fill_kernel(int8_t* p, size_t n, uint32_t seed) {
    // random bytes in [-127, 127] (the multiplier arrays then toggle as they do on real residues: the package power limit, not the
    // issue rate, sets the speed -- zero or slowly varying operands run 1.6x faster)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t h = (i + seed) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        int v = (int)(h & 0xff) - 128;
        p[i] = (int8_t)(v == -128 ? 0 : v);
    }
}

template <int ABL, bool PERS, int VAR = 0>
static void run(int D, int K, int Q, bool check, int sustain = 0) {
    const long ldk = K;
    const size_t nel = (size_t)Q * D * ldk;
    int8_t *dA, *dB; int32_t* dC;
    CK(hipMalloc(&dA, nel)); CK(hipMalloc(&dB, nel)); CK(hipMalloc(&dC, (size_t)Q * D * D * 4));
    fill_kernel<<<65536, 256>>>(dA, nel, 1u);
    fill_kernel<<<65536, 256>>>(dB, nel, 7777u);
    CK(hipGetLastError());
    CK(hipMemset(dC, 0, (size_t)Q * D * D * 4));
    Args g{dA, dB, dC, D, K, ldk, Q};
    const int ntm = D / TM, ntiles = ntm * (ntm + 1) / 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(i8gram_kernel<ABL, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(i8gram_persistent<ABL, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS));
    int* sched; CK(hipMalloc(&sched, 64));
    // VAR & 128: host-built per-XCD item lists: row groups of RG tile rows, swept column by column (4 tiles per column), so that any 32
    // consecutive items of a list touch RG + ~8 distinct strips; row group r of the whole launch belongs to XCD r % 8
    std::vector<std::vector<int>> L(8);
    {
        const int RG = getenv("RG") ? atoi(getenv("RG")) : 4;
        int r = 0;
        for (int q = 0; q < Q; ++q)
            for (int r0 = 0; r0 < ntm; r0 += RG, ++r)
                for (int tn = 0; tn < std::min(ntm, r0 + RG); ++tn)
                    for (int tm = std::max(tn, r0); tm < std::min(ntm, r0 + RG); ++tm) {
                        if (((VAR & 256) != 0) != (tm == tn) && (VAR & 512)) continue;      // VAR & 512: split launches (256: the diagonal part)
                        L[((VAR & 256) ? (r * RG + tm) : r) % 8].push_back(q << 16 | tm << 8 | tn);
                    }
    }
    int maxlen = 0, hl[8];
    for (int y = 0; y < 8; ++y) { hl[y] = (int)L[y].size(); maxlen = std::max(maxlen, hl[y]); }
    int *dlists, *dlens;
    CK(hipMalloc(&dlists, (size_t)8 * maxlen * 4 + 4)); CK(hipMalloc(&dlens, 32));
    for (int y = 0; y < 8; ++y) CK(hipMemcpy(dlists + (size_t)y * maxlen, L[y].data(), L[y].size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dlens, hl, 32, hipMemcpyHostToDevice));
    auto launch = [&] {
        if (PERS) { CK(hipMemsetAsync(sched, 0, 32)); i8gram_persistent<ABL, VAR><<<256, 512, PLDS>>>(g, sched, dlists, dlens, maxlen); }
        else i8gram_kernel<ABL, VAR><<<ntiles * Q, 512, LDS_BYTES>>>(g);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops_exec = 2.0 * ntiles * Q * (double)TM * TN * K, ops_alg = (double)Q * D * (D + 1.0) * K;
    printf("ABL=%d PERS=%d VAR=%d D=%d K=%d Q=%d: %.3f ms  executed %.1f TOPS, algorithmic (lower triangle) %.1f TOPS\n", ABL, (int)PERS, VAR, D, K, Q, ms, ops_exec / ms * 1e-9, ops_alg / ms * 1e-9);
    for (int rep = 0; rep < sustain; ++rep) {          // sustained: does the rate hold once the package is warm / power-limited?
        CK(hipEventRecord(e0));
        for (int k = 0; k < 20; ++k) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  sustained %3d: %.3f ms per launch, executed %.1f TOPS\n", rep, ms / 20, ops_exec / (ms / 20) * 1e-9);
    }
    if (check) {
        std::vector<int8_t> hA(nel), hB(nel);
        std::vector<int32_t> hC((size_t)Q * D * D);
        CK(hipMemcpy(hA.data(), dA, nel, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), dB, nel, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int q = 0; q < Q; ++q)
            for (int i = 0; i < D; i += 7)
                for (int j = 0; j <= i; j += 5) {
                    long ref = 0;
                    for (int t = 0; t < K; ++t) {
                        auto ix = [&](int d) { return (VAR & 4) ? ((size_t)q * D * ldk + ((size_t)(d / 16) * (K / 64) + t / 64) * 1024 + (d % 16) * 64 + t % 64) : (((size_t)q * D + d) * ldk + t); };
                        ref += (int)hA[ix(i)] * (int)hB[ix(j)];
                    }
                    if ((int32_t)ref != hC[((size_t)q * D + i) * D + j]) { if (bad < 5) printf("mismatch q=%d i=%d j=%d ref=%ld got=%d\n", q, i, j, ref, hC[((size_t)q * D + i) * D + j]); ++bad; }
                }
        printf("check: %ld mismatches\n", bad);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

int main(int argc, char** argv) {
    // VAR bits: 1 scalar-base requests, 2 requests spread over both k-steps, 4 blocked plane layout, 8 snake MFMA order, 16 dead wave tiles of
    // diagonal tiles skipped, 32 v_mfma_i32_16x16x64_i8 pipeline (what the library runs), 64 16x16 blocks above the diagonal skipped,
    // 128 host-built per-XCD item lists (row groups of RG tile rows), 512 split launches: off-diagonal items (and with 256 the diagonal
    // items, one skipping code path);  ABL bits: see i8gram_item.   argv[1] = n: n x 20 sustained launches
    if (argc > 1) {
        const int n = atoi(argv[1]);
        run<0, true, 166>(5120, 100096, 60, false, n);             // all items, one launch
        run<0, true, 166 + 512>(5120, 100096, 60, false, n);       // off-diagonal items only
        run<0, true, 166 + 512 + 256>(5120, 100096, 60, false, n); // diagonal items only, blocks above the diagonal skipped
        return 0;
    }
    run<0, true, 38>(3584, 320, 2, true);
    run<0, true, 166>(3584, 320, 2, true);
    run<0, true, 166 + 512>(1024, 384, 3, false);
    run<0, true, 166 + 512 + 256>(1024, 384, 3, false);
    return 0;
}
