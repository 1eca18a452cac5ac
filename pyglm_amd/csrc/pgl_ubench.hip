// Box calibration: what the matrix cores of THIS GPU sustain right now (gfx950), for the two MFMA instructions the hot path is priced against.
// MI355X boxes of one pool differ by ~5 % on power-limited kernels (package power limit, silicon, cooling), so a roofline fraction quoted
// against a constant cannot show a 2 % kernel gain from one run to the next; bench.py runs these loops right before its timed region and
// quotes the product kernel against both the nominal peak and this box's own rate (roofline.box_ubench_tops / frac_vs_this_box).
//   kind 0: v_mfma_i32_16x16x64_i8 on registers holding RANDOM bytes (residue planes are uniformly distributed bytes: the multiplier arrays
//           toggle, and the part sits at its power limit), 8 x 4 blocks of 16 x 16 per wave, two waves per SIMD -- the loop of
//           tools/ubench_i8_bits.hip (4.06-4.12 POP/s on the boxes of round 4 against 4.92 on constant operands);
//   kind 1: v_mfma_f64_16x16x4_f64, 8 independent accumulators per wave, one wave per SIMD (tools/ubench2_f64.hip: 77.8 of 78.6 TFLOP/s).
// A loop runs long enough for the power limiter to settle (a quarter of the time untimed first).  No memory traffic, no LDS.
#include "pgl_common.h"
#include "../../include/pyglm_hip.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void ubench_i8_kernel(int* __restrict__ out, int iters) {
    constexpr int TM = 8, TN = 4;
    v4i acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    v4i a[TM], b[TN];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    auto word = [&] {
        unsigned w = 0;
        for (int k = 0; k < 4; ++k) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; w |= ((h >> 8) & 0xffu) << (8 * k); }
        return (int)w;
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = v4i{word(), word(), word(), word()};
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = v4i{word(), word(), word(), word()};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// (no __launch_bounds__ on purpose: with a 256-thread bound hipcc keeps the accumulators in VGPRs and copies them to AGPRs and back every
// iteration -- 35 instead of 77.8 TFLOP/s; with the default bound it leaves them in place)
__global__ void ubench_f64_kernel(double* __restrict__ out, int iters, double a0, double b0) {
    constexpr int NACC = 8;
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

extern "C" int pgl_ubench_mfma(int kind, double seconds, double* rate_out, double* ms_out, void* hip_stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    PGL_CHECK_ARG((kind == 0 || kind == 1) && seconds > 0.0 && seconds <= 30.0 && rate_out != nullptr);
    const int cus = pgl_device_cus(pgl_device());
    // iterations for ~`seconds` at the nominal rate (the loop is slower under the power limit: it then simply runs longer)
    const double ops_per_iter = kind == 0 ? (double)cus * 8 * 32 * (16.0 * 16 * 64 * 2) : (double)cus * 4 * 8 * (16.0 * 16 * 4 * 2);
    const double nominal = kind == 0 ? 5.0e15 : 78.6e12;
    long iters = (long)(seconds * nominal / ops_per_iter);
    if (iters < 64) iters = 64;
    if (iters > 0x3fffffff) iters = 0x3fffffff;
    void* out = nullptr;
    if (hipMalloc(&out, (size_t)cus * 512 * sizeof(double)) != hipSuccess) { pgl_set_error("pgl_ubench_mfma: scratch allocation failed"); return PGL_ERR_HIP; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PGL_OK;
    float ms = 0.f;
    auto launch = [&](int n) {
        if (kind == 0) hipLaunchKernelGGL(ubench_i8_kernel, dim3(cus), dim3(512), 0, st, static_cast<int*>(out), n);
        else hipLaunchKernelGGL(ubench_f64_kernel, dim3(cus), dim3(256), 0, st, static_cast<double*>(out), n, 0.999, 1e-3);
    };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { pgl_set_error("pgl_ubench_mfma: event creation failed"); rc = PGL_ERR_HIP; }
    if (rc == PGL_OK) {
        launch((int)(iters / 4));                      // settle: clocks and the power limiter
        (void)hipEventRecord(e0, st);
        launch((int)iters);
        (void)hipEventRecord(e1, st);
        if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) {
            pgl_set_error("pgl_ubench_mfma: launch or timing failed");
            rc = PGL_ERR_HIP;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(out);
    if (rc != PGL_OK) return rc;
    *rate_out = ops_per_iter * (double)iters / (ms * 1e-3);        // operations (2 per multiply-add) per second
    if (ms_out) *ms_out = ms;
    return PGL_OK;
}
