#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
constexpr int NBC = 64;
template <int P> __device__ __forceinline__ double lane_bcast(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), P), hi = __builtin_amdgcn_readlane(__double2hiint(v), P);
    return __hiloint2double(hi, lo);
}
template <int P, int J> __device__ __forceinline__ void potrf_row_update(double (&a)[NBC], double lip) {
    if constexpr (J < NBC) { a[J] -= lip * lane_bcast<J>(a[P]); potrf_row_update<P, J + 1>(a, lip); }
}
template <int P> __device__ __forceinline__ void potrf_steps(double (&a)[NBC], int lane, int& bad) {
    if constexpr (P < NBC) {
        double d = lane_bcast<P>(a[P]);
        if (!(d > 0.0)) { bad = 1; d = 1.0; }
        const double r = sqrt(d), inv = 1.0 / r;
        a[P] = lane == P ? r : a[P] * inv;
        potrf_row_update<P, P + 1>(a, a[P]);
        potrf_steps<P + 1>(a, lane, bad);
    }
}
template <int K, int J>
__device__ __forceinline__ void inv_row_update(double (&x)[NBC], double lik, bool below) {
    if constexpr (J <= K) {
        // (the broadcast is taken by ALL lanes and the update applied by a select: a v_readlane under a divergent branch reads a lane
        // the compiler considers inactive -- undefined in its model, and wrong in practice)
        const double xk = lane_bcast<K>(x[J]);
        x[J] = below ? x[J] - lik * xk : x[J];                // X[i][J] -= L[i][K] X[K][J]   for the rows i > K
        inv_row_update<K, J + 1>(x, lik, below);
    }
}
template <int K>
__device__ __forceinline__ void inv_steps(const double (&a)[NBC], double (&x)[NBC], int lane) {
    if constexpr (K < NBC) {
        // row K of X is final once scaled by 1 / L[K][K]; rows below it take its contribution
        const double rk = 1.0 / lane_bcast<K>(a[K]);
#pragma unroll
        for (int j = 0; j <= K; ++j) x[j] = lane == K ? x[j] * rk : x[j];
        inv_row_update<K, 0>(x, a[K], lane > K);
        inv_steps<K + 1>(a, x, lane);
    }
}
__global__ __launch_bounds__(64) void k(const double* A, double* L, double* X, int* bad_out) {
    const int lane = threadIdx.x;
    double a[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) a[j] = j <= lane ? A[j * NBC + lane] : 0.0;
    int bad = 0;
    potrf_steps<0>(a, lane, bad);
#pragma unroll
    for (int j = 0; j < NBC; ++j) L[lane * NBC + j] = j <= lane ? a[j] : 0.0;
    double x[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) x[j] = j == lane ? 1.0 : 0.0;
    inv_steps<0>(a, x, lane);
#pragma unroll
    for (int j = 0; j < NBC; ++j) X[lane * NBC + j] = j <= lane ? x[j] : 0.0;
    if (lane == 0) *bad_out = bad;
}
int main() {
    const int n = NBC;
    std::vector<double> A(n * n), B(n * n);
    unsigned s = 1;
    for (auto& v : B) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double t = 0; for (int k = 0; k < n; ++k) t += B[i * n + k] * B[j * n + k]; A[i * n + j] = t + (i == j ? n : 0); }
    // CPU reference
    std::vector<double> Lr(n * n, 0.0);
    for (int j = 0; j < n; ++j) { double d = A[j * n + j]; for (int k = 0; k < j; ++k) d -= Lr[j * n + k] * Lr[j * n + k]; Lr[j * n + j] = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) { double t = A[i * n + j]; for (int k = 0; k < j; ++k) t -= Lr[i * n + k] * Lr[j * n + k]; Lr[i * n + j] = t / Lr[j * n + j]; } }
    double *dA, *dL, *dX; int* db;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dX, n * n * 8); hipMalloc(&db, 4);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dL, dX, db);
    std::vector<double> L(n * n), X(n * n); int bad;
    hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(&bad, db, 4, hipMemcpyDeviceToHost);
    double eL = 0, eX = 0;
    for (int i = 0; i < n * n; ++i) eL = std::fmax(eL, std::fabs(L[i] - Lr[i]));
    // check L X = I
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double t = 0; for (int k = 0; k < n; ++k) t += Lr[i * n + k] * X[k * n + j]; eX = std::fmax(eX, std::fabs(t - (i == j))); }
    printf("bad %d  max|L - Lref| %.3e  max|L X - I| %.3e\n", bad, eL, eX);
    return 0;
}
