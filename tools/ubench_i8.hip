// int8 MFMA (v_mfma_i32_32x32x32_i8) ceilings on gfx950 for the next-step study in DESIGN.md section 9:
//   (1) register-only issue rate,  (2) the same wave tile fed from LDS with ds_read_b128 fragments (padded rows),
// for wave tiles of TM x TN 32x32 accumulators and 1 or 2 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 ubench_i8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int TM, int TN, int THR>
__global__ __launch_bounds__(THR) void k_reg(int* out, int iters) {
    v16i acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    v4i a[TM], b[TN];
    for (int i = 0; i < TM; ++i) a[i] = v4i{(int)threadIdx.x, i, 3, 4};
    for (int j = 0; j < TN; ++j) b[j] = v4i{j, (int)threadIdx.x * 7, 1, 2};
    if (iters < 0) {        // random operand bytes (power: the multiplier arrays toggle as they do on real residues)
        iters = -iters;
        unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
        auto nx = [&] { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (int)h; };
        for (int i = 0; i < TM; ++i) a[i] = v4i{nx(), nx(), nx(), nx()};
        for (int j = 0; j < TN; ++j) b[j] = v4i{nx(), nx(), nx(), nx()};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same with v_mfma_i32_16x16x64_i8 (4 accumulator registers per 16 x 16 block): TM x TN blocks of 16 x 16
template <int TM, int TN, int THR>
__global__ __launch_bounds__(THR) void k_reg16(int* out, int iters) {
    v4i acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    v4i a[TM], b[TN];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    auto nx = [&] { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (int)h; };
    const bool rnd = iters < 0;
    if (rnd) iters = -iters;
    for (int i = 0; i < TM; ++i) a[i] = rnd ? v4i{nx(), nx(), nx(), nx()} : v4i{(int)threadIdx.x, i, 3, 4};
    for (int j = 0; j < TN; ++j) b[j] = rnd ? v4i{nx(), nx(), nx(), nx()} : v4i{j, (int)threadIdx.x * 7, 1, 2};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS-fed: every wave owns a [TM*32 + TN*32][BKB + PAD] byte region (or shares one: SHARE) and sweeps it repeatedly
template <int TM, int TN, int BKB, int PAD, bool SHARE, int THR>
__global__ __launch_bounds__(THR) void k_lds(int* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int RS = BKB + PAD;                       // row stride in bytes
    constexpr int ROWS = (TM + TN) * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = lds + (SHARE ? 0 : wave * ROWS * RS);
    for (int i = threadIdx.x; i < (SHARE ? 1 : blockDim.x / 64) * ROWS * RS / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    v16i acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    const char* pa = base + (lane & 31) * RS + (lane >> 5) * 16;
    const char* pb = pa + TM * 32 * RS;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");                  // the fragments are re-read from LDS every sweep
#pragma unroll
        for (int kk = 0; kk < BKB / 32; ++kk) {
            v4i a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const v4i*>(pa + i * 32 * RS + kk * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const v4i*>(pb + j * 32 * RS + kk * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static void timeit(const char* name, double ops_per_launch, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s %8.3f ms  %8.1f TOPS\n", name, ms, ops_per_launch / (ms * 1e-3) * 1e-12);
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    int* out; CK(hipMalloc(&out, sizeof(int) * 1024 * cus * 2));
    const double mfma_ops = 32.0 * 32 * 32 * 2;
    if (argc > 1) {     // sustained register-only rate: argv[1] = +1 constant operands, -1 random operands; ~6 s
        const int sgn = atoi(argv[1]);
        if (argc > 2) {   // 16x16x64 variant, 8 x 4 blocks (the same 128 x 64 wave tile)
            for (int rep = 0; rep < 8; ++rep)
                timeit(sgn < 0 ? "reg16 8x4 512 thr/CU, random operands, sustained" : "reg16 8x4 512 thr/CU, constant operands, sustained",
                       (double)cus * 8 * 2000000.0 * 32 * (16.0 * 16 * 64 * 2), [&] { k_reg16<8, 4, 512><<<cus, 512>>>(out, sgn * 2000000); });
            return 0;
        }
        for (int rep = 0; rep < 12; ++rep)
            timeit(sgn < 0 ? "reg 4x2 512 thr/CU, random operands, sustained" : "reg 4x2 512 thr/CU, constant operands, sustained", (double)cus * 8 * 4000000.0 * 8 * mfma_ops,
                   [&] { k_reg<4, 2, 512><<<cus, 512>>>(out, sgn * 4000000); });
        return 0;
    }
#define REG(TM, TN, THR, IT) timeit("reg  " #TM "x" #TN " tiles, " #THR " thr/CU", (double)cus * (THR / 64) * IT * TM * TN * mfma_ops, [&] { k_reg<TM, TN, THR><<<cus, THR>>>(out, IT); })
    REG(2, 2, 256, 20000); REG(4, 2, 256, 10000); REG(4, 4, 256, 5000); REG(2, 2, 512, 20000); REG(4, 2, 512, 10000);
#define LDSB(TM, TN, BKB, PAD, SHARE, THR, IT)                                                                                   \
    {                                                                                                                             \
        auto kern = k_lds<TM, TN, BKB, PAD, SHARE, THR>;                                                                               \
        size_t bytes = (size_t)(SHARE ? 1 : THR / 64) * (TM + TN) * 32 * (BKB + PAD);                                            \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));    \
        timeit("lds  " #TM "x" #TN " BK=" #BKB " pad=" #PAD " share=" #SHARE " " #THR " thr/CU",                                \
               (double)cus * (THR / 64) * IT * (BKB / 32) * TM * TN * mfma_ops, [&] { kern<<<cus, THR, bytes>>>(out, IT); });    \
    }
    LDSB(2, 2, 128, 0, false, 256, 4000); LDSB(2, 2, 128, 16, false, 256, 4000);
    LDSB(4, 2, 128, 16, false, 256, 2000); LDSB(4, 4, 64, 16, false, 256, 2000);
    LDSB(4, 2, 128, 16, true, 256, 2000); LDSB(4, 2, 128, 16, true, 512, 2000);
    LDSB(4, 4, 128, 16, true, 256, 1000); LDSB(4, 4, 128, 16, true, 512, 1000);
    LDSB(2, 4, 128, 16, true, 512, 2000); LDSB(2, 2, 128, 16, true, 512, 4000);
    return 0;
}
