// Does the ORDER in which a wave issues its block of v_mfma_i32_16x16x64_i8 matter under the package power limit?  Consecutive MFMAs that share
// an operand leave half of the multiplier inputs unchanged.  Row-major over an 8 x 4 block of accumulators (the order of i8_gram_kernel and of
// pgl_ubench_mfma) changes B on every instruction and A once per row; a boustrophedon order keeps B across the row change as well.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_i8_order.hip -o tools/bin/ubench_i8_order && tools/bin/ubench_i8_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int ORDER>
__global__ __launch_bounds__(512) void k(int* out, int iters) {
    constexpr int TM = 8, TN = 4;
    v4i acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    v4i a[TM], b[TN];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    auto word = [&] { unsigned w = 0; for (int q = 0; q < 4; ++q) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; w |= ((h >> 8) & 0xffu) << (8 * q); } return (int)w; };
    for (int i = 0; i < TM; ++i) a[i] = v4i{word(), word(), word(), word()};
    for (int j = 0; j < TN; ++j) b[j] = v4i{word(), word(), word(), word()};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < TN; ++s) {
                int ii = i, jj = s;
                if (ORDER == 1) jj = (i & 1) ? TN - 1 - s : s;                       // boustrophedon: B survives the row change
                if (ORDER == 2) { const int t = i * TN + s; jj = t / TM; ii = (jj & 1) ? TM - 1 - t % TM : t % TM; }   // column-major snake: A changes every time, B 1 in 8
                if (ORDER == 3) { ii = (i + s) % TM; }                             // diagonal: both operands change on every instruction
                if (ORDER == 4) {                                                   // two rows at a time, zigzag: A and B change alternately, one per step
                    const int t = (i & 1) * TN + s, c = t >> 1, ph = t & 1;         // pair of rows (i & ~1, i | 1), 2 TN steps
                    jj = c; ii = (i & ~1) + ((c & 1) ? 1 - ph : ph);
                }
                if (ORDER == 5) {                                                   // 2 x 2 mini-blocks in a ring: (0,0) (0,1) (1,1) (1,0), next pair of columns
                    const int t = (i & 1) * TN + s, blk = t >> 2, q = t & 3;
                    ii = (i & ~1) + (q >> 1); jj = 2 * blk + ((q == 1 || q == 2) ? 1 : 0);
                }
                acc[ii][jj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ii], b[jj], acc[ii][jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    int sum = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    int* out; CK(hipMalloc(&out, sizeof(int) * 512 * cus));
    const int iters = 1200000;
    const double ops = (double)cus * 8 * iters * 32 * (16.0 * 16 * 64 * 2);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[6] = {"row-major (A per row, B every instruction)", "boustrophedon rows", "column-major snake", "diagonal (both change)",
                            "two-row zigzag (A, B alternately)", "2 x 2 rings"};
    for (int rep = 0; rep < 3; ++rep)
        for (int o = 0; o < 6; ++o) {
            auto launch = [&](int n) {
                if (o == 0) k<0><<<cus, 512>>>(out, n); else if (o == 1) k<1><<<cus, 512>>>(out, n); else if (o == 2) k<2><<<cus, 512>>>(out, n);
                else if (o == 3) k<3><<<cus, 512>>>(out, n); else if (o == 4) k<4><<<cus, 512>>>(out, n); else k<5><<<cus, 512>>>(out, n);
            };
            launch(iters / 4);
            CK(hipEventRecord(e0));
            launch(iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("rep %d  %-46s %8.1f ms  %7.1f TOP/s\n", rep, names[o], ms, ops / (ms * 1e-3) * 1e-12);
            fflush(stdout);
        }
    return 0;
}
