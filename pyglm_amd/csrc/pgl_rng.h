// Device-side counter-based random stream + Polya-gamma PG(1, z) sampler for gfx950.
//
// Stands where the reference calls the third-party `pypolyagamma.pgdrawvpar`
// (/root/reference/pyglm/regression.py:501-508).  Algorithm: Polson, Scott & Windle (2013) Devroye-style
// alternating-series sampler (truncation t = 0.64); PG(b, z) with integer b is the sum of b PG(1, z) draws.
// Stream: Philox4x32-10, key = seed, counter = (j | purpose<<24, element, stream lo, stream hi); one lane
// owns one draw and walks j = 0,1,2,... -- no shared state, results independent of launch geometry and of
// how neurons are sharded over GPUs.  The same stream is specified (and implemented separately, in plain C)
// in oracle/pg_oracle.c, which is the checker for this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PGL_PURPOSE_PG 1u
#define PGL_PG_TRUNC 0.64
#define PGL_PI 3.141592653589793238462643383279502884

struct PglPhilox {
    uint32_t k0, k1;        // key
    uint32_t elem, s0, s1;  // counter words 1..3
    uint32_t j;             // counter word 0 (low 24 bits) = calls made
    uint32_t purpose;
    double buf;             // second uniform of the last call
    int have;
};

__device__ __forceinline__ void pgl_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                  uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

__device__ __forceinline__ double pgl_u64_to_unit(uint64_t x) { return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

__device__ __forceinline__ void pgl_rng_init(PglPhilox& r, uint64_t seed, uint64_t stream, uint64_t elem, uint32_t purpose) {
    r.k0 = (uint32_t)seed; r.k1 = (uint32_t)(seed >> 32);
    r.elem = (uint32_t)elem; r.s0 = (uint32_t)stream; r.s1 = (uint32_t)(stream >> 32);
    r.j = 0; r.purpose = purpose; r.have = 0; r.buf = 0.0;
}

__device__ __forceinline__ double pgl_unif(PglPhilox& r) {
    if (r.have) { r.have = 0; return r.buf; }
    uint32_t o0, o1, o2, o3;
    pgl_philox4x32_10(r.j | (r.purpose << 24), r.elem, r.s0, r.s1, r.k0, r.k1, o0, o1, o2, o3);
    r.j++;
    r.buf = pgl_u64_to_unit((uint64_t)o2 | ((uint64_t)o3 << 32));
    r.have = 1;
    return pgl_u64_to_unit((uint64_t)o0 | ((uint64_t)o1 << 32));
}
__device__ __forceinline__ double pgl_expon(PglPhilox& r) { return -log(pgl_unif(r)); }
__device__ __forceinline__ double pgl_norm(PglPhilox& r) {
    const double u1 = pgl_unif(r), u2 = pgl_unif(r);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * PGL_PI * u2);
}

__device__ __forceinline__ double pgl_log_pnorm(double x) { return log(0.5 * erfc(-x * 0.70710678118654752440)); }

__device__ __forceinline__ double pgl_pg_a(int n, double x) {
    const double K = (n + 0.5) * PGL_PI;
    if (x > PGL_PG_TRUNC) return K * exp(-0.5 * K * K * x);
    if (x > 0) {
        const double expnt = -1.5 * (log(0.5 * PGL_PI) + log(x)) + log(K) - 2.0 * (n + 0.5) * (n + 0.5) / x;
        return exp(expnt);
    }
    return 0.0;
}

__device__ __forceinline__ double pgl_pg_mass_texpon(double Z) {
    const double t = PGL_PG_TRUNC;
    const double fz = 0.125 * PGL_PI * PGL_PI + 0.5 * Z * Z;
    const double b = sqrt(1.0 / t) * (t * Z - 1);
    const double a = sqrt(1.0 / t) * (t * Z + 1) * -1.0;
    const double x0 = log(fz) + fz * t;
    const double xb = x0 - Z + pgl_log_pnorm(b);
    const double xa = x0 + Z + pgl_log_pnorm(a);
    const double qdivp = 4 / PGL_PI * (exp(xb) + exp(xa));
    return 1.0 / (1.0 + qdivp);
}

__device__ __forceinline__ double pgl_pg_rtigauss(double Z, PglPhilox& r) {
    const double t = PGL_PG_TRUNC;
    double X = t + 1.0;
    Z = fabs(Z);
    if (1.0 / t > Z) {
        double alpha = 0.0;
        while (pgl_unif(r) > alpha) {
            double E1 = pgl_expon(r), E2 = pgl_expon(r);
            while (E1 * E1 > 2 * E2 / t) { E1 = pgl_expon(r); E2 = pgl_expon(r); }
            X = 1 + E1 * t;
            X = t / (X * X);
            alpha = exp(-0.5 * Z * Z * X);
        }
    } else {
        const double mu = 1.0 / Z;
        while (X > t) {
            double Y = pgl_norm(r);
            Y *= Y;
            const double half_mu = 0.5 * mu, mu_Y = mu * Y;
            X = mu + half_mu * mu_Y - half_mu * sqrt(4 * mu_Y + mu_Y * mu_Y);
            if (pgl_unif(r) > mu / (mu + X)) X = mu * mu / X;
        }
    }
    return X;
}

__device__ __forceinline__ double pgl_pg1(double z, PglPhilox& r) {
    const double Z = fabs(z) * 0.5;
    const double fz = 0.125 * PGL_PI * PGL_PI + 0.5 * Z * Z;
    const double mass = pgl_pg_mass_texpon(Z);
    for (;;) {
        double X;
        if (pgl_unif(r) < mass) X = PGL_PG_TRUNC + pgl_expon(r) / fz;
        else X = pgl_pg_rtigauss(Z, r);
        double S = pgl_pg_a(0, X);
        const double Y = pgl_unif(r) * S;
        int n = 0;
        for (;;) {
            ++n;
            if (n & 1) { S -= pgl_pg_a(n, X); if (Y <= S) return 0.25 * X; }
            else       { S += pgl_pg_a(n, X); if (Y > S) break; }
        }
    }
}

// PG(b, z), b a non-negative integer (Bernoulli b = 1; negative-binomial b = y + xi with integer xi)
__device__ __forceinline__ double pgl_pg_draw(double b, double z, uint64_t seed, uint64_t stream, uint64_t elem) {
    PglPhilox r;
    pgl_rng_init(r, seed, stream, elem, PGL_PURPOSE_PG);
    double s = 0.0;
    const long nb = (long)b;
    for (long k = 0; k < nb; ++k) s += pgl_pg1(z, r);
    return s;
}
