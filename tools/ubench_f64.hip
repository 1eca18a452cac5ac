// Microbenchmarks that pin the gfx950 numbers the Gram design depends on and that the
// CDNA4 guide does not list: v_mfma_f64_16x16x4_f64 issue rate, v_fma_f64 rate, whether the
// two pipes add, HBM streaming rate for 16-B loads, and the f64 MFMA C/D lane layout.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o tools/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// per MFMA, NV independent v_fma_f64 interleaved in the same wave
template <int NACC, int NV>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    double v[NACC * NV];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int i = 0; i < NACC * NV; ++i) v[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[i * NV + j] = __builtin_fma(a, v[i * NV + j], b);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NACC * NV; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// even waves MFMA-only, odd waves FMA-only (separate waves on the same SIMDs when 512 threads/WG)
__global__ __launch_bounds__(512) void k_split(double* out, int iters, double a0, double b0) {
    int wave = threadIdx.x >> 6;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    double s = 0;
    if (wave < 4) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double v[16];
        for (int i = 0; i < 16; ++i) v[i] = i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(a, v[i], b);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(a, v[i], b);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(a, v[i], b);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(a, v[i], b);
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_stream(const double2* __restrict__ in, double* out, size_t n2) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        double2 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    for (; i < n2; i += stride) { double2 a = in[i]; s += a.x + a.y; }
    if (s == 1.2345e300) out[0] = s;
}

// layout probe: one wave, A[i][k], B[k][j] chosen so C[i][j] = sum_k A[i][k]*B[k][j] is unique per (i,j)
__global__ void k_layout(double* C /*64x4*/, int* lane_row, int* lane_col) {
    int l = threadIdx.x;
    // assumption under test: A operand lane l = A[i=l&15][k=l>>4], B operand lane l = B[k=l>>4][j=l&15]
    int i = l & 15, k = l >> 4, j = l & 15;
    double a = (double)(i + 1) * (k == 0 ? 1.0 : k == 1 ? 100.0 : k == 2 ? 1e4 : 1e6);  // A[i][k]
    double b = (double)(j + 1) * (k == 0 ? 1.0 : k == 1 ? 0.5 : k == 2 ? 0.25 : 0.125) + (k == 3 ? j * j : 0);  // B[k][j] asymmetric
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[l * 4 + r] = c[r];
}

static double time_kernel(void (*launch)(hipStream_t), int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(0); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch(0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e-3;
}

static double* g_out; static int g_iters = 4096; static int g_blocks;
static const double2* g_in; static size_t g_n2;

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    int cus = p.multiProcessorCount;
    CK(hipMalloc(&g_out, sizeof(double) * 512 * cus * 16));

    // ---- layout probe ----
    {
        double* dC; CK(hipMalloc(&dC, 256 * 8));
        k_layout<<<1, 64>>>(dC, nullptr, nullptr); CK(hipDeviceSynchronize());
        std::vector<double> C(256); CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
        auto A = [](int i, int k) { return (double)(i + 1) * (k == 0 ? 1.0 : k == 1 ? 100.0 : k == 2 ? 1e4 : 1e6); };
        auto Bm = [](int k, int j) { return (double)(j + 1) * (k == 0 ? 1.0 : k == 1 ? 0.5 : k == 2 ? 0.25 : 0.125) + (k == 3 ? j * j : 0); };
        int ok_guide = 0, tot = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            double got = C[l * 4 + r];
            // find (i,j) matching
            int fi = -1, fj = -1;
            for (int i = 0; i < 16 && fi < 0; ++i) for (int j = 0; j < 16; ++j) {
                double ref = 0; for (int k = 0; k < 4; ++k) ref += A(i, k) * Bm(k, j);
                if (fabs(ref - got) <= 1e-9 * fabs(ref)) { fi = i; fj = j; break; }
            }
            int gi = (l >> 4) + 4 * r, gj = l & 15;  // guide: col=lane&15,row=(lane>>4)+4*reg
            ok_guide += (fi == gi && fj == gj); ++tot;
            if (l < 2 || l == 17 || l == 63) printf("layout lane %d reg %d -> (row %d, col %d)\n", l, r, fi, fj);
        }
        printf("LAYOUT guide-formula matches %d/%d\n", ok_guide, tot);
    }

    auto report = [&](const char* name, double sec, double flops) {
        printf("%-44s %8.3f ms  %8.2f TFLOP/s\n", name, sec * 1e3, flops / sec * 1e-12);
    };
    // ---- MFMA f64 ----
    for (int wgs_per_cu : {1, 2}) {
        g_blocks = cus * wgs_per_cu;
        double mf = 2.0 * 16 * 16 * 4;
        double s;
        s = time_kernel([](hipStream_t st) { k_mfma<1><<<g_blocks, 256, 0, st>>>(g_out, g_iters, 1.0, 1e-3); }, 5);
        printf("[wg/cu=%d] ", wgs_per_cu); report("mfma_f64_16x16x4 1 acc (dependent)", s, mf * 1 * g_iters * 4.0 * g_blocks);
        s = time_kernel([](hipStream_t st) { k_mfma<2><<<g_blocks, 256, 0, st>>>(g_out, g_iters, 1.0, 1e-3); }, 5);
        printf("[wg/cu=%d] ", wgs_per_cu); report("mfma_f64_16x16x4 2 acc", s, mf * 2 * g_iters * 4.0 * g_blocks);
        s = time_kernel([](hipStream_t st) { k_mfma<4><<<g_blocks, 256, 0, st>>>(g_out, g_iters, 1.0, 1e-3); }, 5);
        printf("[wg/cu=%d] ", wgs_per_cu); report("mfma_f64_16x16x4 4 acc", s, mf * 4 * g_iters * 4.0 * g_blocks);
        s = time_kernel([](hipStream_t st) { k_mfma<16><<<g_blocks, 256, 0, st>>>(g_out, g_iters, 1.0, 1e-3); }, 5);
        printf("[wg/cu=%d] ", wgs_per_cu); report("mfma_f64_16x16x4 16 acc", s, mf * 16 * g_iters * 4.0 * g_blocks);
        // VALU fma
        s = time_kernel([](hipStream_t st) { k_fma<16><<<g_blocks, 256, 0, st>>>(g_out, g_iters, 0.999, 1e-3); }, 5);
        printf("[wg/cu=%d] ", wgs_per_cu); report("v_fma_f64 16 indep", s, 2.0 * 16 * g_iters * 256.0 * g_blocks);
        // mixed in one wave
        s = time_kernel([](hipStream_t st) { k_mix<4, 4><<<g_blocks, 256, 0, st>>>(g_out, g_iters, 0.999, 1e-3); }, 5);
        printf("[wg/cu=%d] ", wgs_per_cu); report("mix 4 mfma + 16 fma /iter (total flops)", s, (mf * 4 * 4.0 + 2.0 * 16 * 256) * g_iters * g_blocks);
        s = time_kernel([](hipStream_t st) { k_mix<4, 8><<<g_blocks, 256, 0, st>>>(g_out, g_iters, 0.999, 1e-3); }, 5);
        printf("[wg/cu=%d] ", wgs_per_cu); report("mix 4 mfma + 32 fma /iter (total flops)", s, (mf * 4 * 4.0 + 2.0 * 32 * 256) * g_iters * g_blocks);
        s = time_kernel([](hipStream_t st) { k_mix<4, 16><<<g_blocks, 256, 0, st>>>(g_out, g_iters, 0.999, 1e-3); }, 5);
        printf("[wg/cu=%d] ", wgs_per_cu); report("mix 4 mfma + 64 fma /iter (total flops)", s, (mf * 4 * 4.0 + 2.0 * 64 * 256) * g_iters * g_blocks);
    }
    {
        g_blocks = cus;
        double mf = 2.0 * 16 * 16 * 4;
        double s = time_kernel([](hipStream_t st) { k_split<<<g_blocks, 512, 0, st>>>(g_out, g_iters, 0.999, 1e-3); }, 5);
        report("split waves: 4 mfma-waves + 4 fma-waves (total)", s, (mf * 4 * 4.0 + 2.0 * 64 * 256) * g_iters * g_blocks);
    }
    // ---- HBM stream ----
    {
        size_t bytes = (size_t)8 << 30;  // 8 GiB > 256 MiB infinity cache
        double2* in; CK(hipMalloc(&in, bytes)); CK(hipMemset(in, 0, bytes));
        g_in = in; g_n2 = bytes / 16;
        for (int bpc : {4, 8, 16}) {
            g_blocks = cus * bpc;
            double s = time_kernel([](hipStream_t st) { k_stream<<<g_blocks, 256, 0, st>>>(g_in, g_out, g_n2); }, 5);
            printf("stream read 8 GiB, %2d wg/cu: %8.3f ms  %8.2f TB/s\n", bpc, s * 1e3, bytes / s * 1e-12);
        }
        CK(hipFree(in));
    }
    return 0;
}
