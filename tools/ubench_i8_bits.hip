// Operand-width energy experiment for the integer Gram: the kernel is held by the package power limit, so its rate
// is the power budget over the energy per operation.  Does v_mfma_i32_16x16x64_i8 burn less on operands confined to 7 or 6 bits (moduli
// <= 128 / <= 64 would need 15 / 17 residue planes instead of 13)?  Bare MFMA loop on registers (8 x 4 blocks of 16 x 16, two waves per
// SIMD: the round-1 kernel's wave tile), random operand bytes drawn from a range, ~4 s per variant so that the power limiter settles.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_i8_bits.hip -o tools/bin/ubench_i8_bits
// Adopt a narrower residue range only if  rate(bits) x 13 / planes(bits)  beats rate(8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

// mode 0: bytes uniform in [-2^(bits-1), 2^(bits-1));  mode 1: uniform in [0, 2^(bits-1))  (non-negative residues)
template <int TM, int TN, int THR>
__global__ __launch_bounds__(THR) void k_reg16(int* out, int iters, int bits, int mode) {
    v4i acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    v4i a[TM], b[TN];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    auto byte = [&] {
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        const int span = mode ? (1 << (bits - 1)) : (1 << bits);
        int v = (int)((h >> 8) % (unsigned)span);
        if (!mode) v -= span / 2;
        return (unsigned)(v & 0xff);
    };
    auto word = [&] { return (int)(byte() | (byte() << 8) | (byte() << 16) | (byte() << 24)); };
    for (int i = 0; i < TM; ++i) a[i] = v4i{word(), word(), word(), word()};
    for (int j = 0; j < TN; ++j) b[j] = v4i{word(), word(), word(), word()};
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

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    int* out; CK(hipMalloc(&out, sizeof(int) * 1024 * cus));
    const int iters = 1500000;
    const double ops = (double)cus * 8 * iters * 32 * (16.0 * 16 * 64 * 2);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 2; ++mode)
            for (int bits = 8; bits >= 5; --bits) {
                k_reg16<8, 4, 512><<<cus, 512>>>(out, iters / 4, bits, mode);     // settle
                CK(hipEventRecord(e0));
                k_reg16<8, 4, 512><<<cus, 512>>>(out, iters, bits, mode);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("rep %d  %s  %d bits: %8.1f ms  %7.1f TOP/s\n", rep, mode ? "non-negative [0, 2^(b-1))" : "symmetric [-2^(b-1), 2^(b-1))", bits, ms,
                       ops / (ms * 1e-3) * 1e-12);
                fflush(stdout);
            }
    return 0;
}
