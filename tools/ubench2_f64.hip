// Cycle-accurate follow-up: cycles per v_mfma_f64_16x16x4_f64 (s_memtime) and effective clock
// (s_memtime vs wall_clock64 @100 MHz) for 1/2/4 waves per SIMD, long-running (sustained clocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int NV>
__global__ void k_cyc(double* out, long long* cyc, long long* wall, int iters, double a0, double b0) {
    d4 acc[NACC > 0 ? NACC : 1];
    double v[NV > 0 ? NV : 1];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int i = 0; i < NV; ++i) v[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    long long w0 = wall_clock64();
    long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = __builtin_fma(a, v[j], b);
    }
    long long c1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; wall[0] = w1 - w0; }
}

template <int NACC, int NV>
void run(const char* name, int cus, int threads, int wg_per_cu, int iters) {
    double* out; long long *cyc, *wall;
    CK(hipMalloc(&out, sizeof(double) * threads * cus * wg_per_cu));
    CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&wall, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_cyc<NACC, NV><<<cus * wg_per_cu, threads>>>(out, cyc, wall, 256, 0.999, 1e-3);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_cyc<NACC, NV><<<cus * wg_per_cu, threads>>>(out, cyc, wall, iters, 0.999, 1e-3);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long hc, hw; CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost));
    double waves_per_simd = (double)threads / 64 * wg_per_cu / 4;
    double nm = (double)(NACC > 0 ? NACC : NV) * iters;  // mfma per wave
    double flops = (2048.0 * NACC * (threads / 64) + 2.0 * NV * threads) * iters * cus * wg_per_cu;
    printf("%-34s waves/SIMD=%.0f  %8.2f ms  %7.2f TF  cyc/mfma/wave=%7.1f  cyc/mfma/SIMD=%6.1f  clk=%.3f GHz\n", name,
           waves_per_simd, ms, flops / (ms * 1e-3) * 1e-12, hc / nm, hc / nm / waves_per_simd, (double)hc / ((double)hw * 10e-9) * 1e-9);
    CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(wall));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    printf("wall clock rate kHz=%d\n", p.clockRate);
    const int IT = 40000;
    run<1, 0>("mfma 1acc", cus, 256, 1, IT * 4);
    run<2, 0>("mfma 2acc", cus, 256, 1, IT * 2);
    run<4, 0>("mfma 4acc", cus, 256, 1, IT);
    run<8, 0>("mfma 8acc", cus, 256, 1, IT);
    run<16, 0>("mfma 16acc", cus, 256, 1, IT / 2);
    run<4, 0>("mfma 4acc", cus, 512, 1, IT);
    run<8, 0>("mfma 8acc", cus, 512, 1, IT / 2);
    run<4, 0>("mfma 4acc", cus, 1024, 1, IT / 2);
    run<8, 0>("mfma 8acc", cus, 256, 2, IT / 2);
    run<0, 16>("fma 16", cus, 256, 1, IT * 8);
    run<0, 16>("fma 16", cus, 512, 1, IT * 4);
    run<0, 16>("fma 16", cus, 1024, 1, IT * 2);
    run<4, 8>("mix 4mfma+8fma", cus, 256, 1, IT);
    run<4, 16>("mix 4mfma+16fma", cus, 256, 1, IT);
    run<4, 32>("mix 4mfma+32fma", cus, 256, 1, IT);
    run<4, 16>("mix 4mfma+16fma", cus, 512, 1, IT);
    run<4, 32>("mix 4mfma+32fma", cus, 512, 1, IT / 2);
    run<4, 64>("mix 4mfma+64fma", cus, 512, 1, IT / 2);
    return 0;
}
