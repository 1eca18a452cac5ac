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
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int TM = 256, TN = 256, BKB = 64, NST = 4;
constexpr int STAGE_BYTES = (TM + TN) * BKB;          // 32 KiB
constexpr int LDS_BYTES = NST * STAGE_BYTES;

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

__global__ __launch_bounds__(512) void i8gram_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int ntm = g.D / TM;
    const int ntiles = ntm * (ntm + 1) / 2;
    const int q = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    const int tm = isqrt_tri(tile), tn = tile - tm * (tm + 1) / 2;
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
    const char* gp[4];
    int loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rq = wv + 8 * i, row = 16 * rq + (lane >> 2);      // 0..511
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int8_t* base = row < TM ? A + (long)(m0 + row) * g.ldk : B + (long)(n0 + row - TM) * g.ldk;
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

__global__ void fill_kernel(int8_t* p, size_t n, uint32_t seed) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    int v = (int)(h >> 24) - 128;
    p[i] = (int8_t)(v == -128 ? 0 : v);
}

static void run(int D, int K, int Q, bool check) {
    const long ldk = K;
    const size_t nel = (size_t)Q * D * ldk;
    int8_t *dA, *dB; int32_t* dC;
    CK(hipMalloc(&dA, nel)); CK(hipMalloc(&dB, nel)); CK(hipMalloc(&dC, (size_t)Q * D * D * 4));
    fill_kernel<<<(unsigned)((nel + 255) / 256), 256>>>(dA, nel, 1u);
    fill_kernel<<<(unsigned)((nel + 255) / 256), 256>>>(dB, nel, 77u);
    CK(hipMemset(dC, 0, (size_t)Q * D * D * 4));
    Args g{dA, dB, dC, D, K, ldk, Q};
    const int ntm = D / TM, ntiles = ntm * (ntm + 1) / 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(i8gram_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    i8gram_kernel<<<ntiles * Q, 512, LDS_BYTES>>>(g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    i8gram_kernel<<<ntiles * Q, 512, LDS_BYTES>>>(g);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops_exec = 2.0 * ntiles * Q * (double)TM * TN * K, ops_alg = (double)Q * D * (D + 1.0) * K;
    printf("D=%d K=%d Q=%d: %.3f ms  executed %.1f TOPS, algorithmic (lower triangle) %.1f TOPS\n", D, K, Q, ms, ops_exec / ms * 1e-9, ops_alg / ms * 1e-9);
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
                    for (int t = 0; t < K; ++t) ref += (int)hA[((size_t)q * D + i) * ldk + t] * (int)hB[((size_t)q * D + j) * ldk + t];
                    if ((int32_t)ref != hC[((size_t)q * D + i) * D + j]) { if (bad < 5) printf("mismatch q=%d i=%d j=%d ref=%ld got=%d\n", q, i, j, ref, hC[((size_t)q * D + i) * D + j]); ++bad; }
                }
        printf("check: %ld mismatches\n", bad);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

int main() {
    run(512, 1024, 2, true);
    run(5120, 100096, 15, false);
    return 0;
}
