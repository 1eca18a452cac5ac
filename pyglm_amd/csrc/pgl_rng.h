// Device-side counter-based random stream + Polya-gamma PG(b, z) sampler for gfx950.
//
// Stands where the reference calls the third-party `pypolyagamma.pgdrawvpar`
// (/root/reference/pyglm/regression.py:501-508; shapes b = b_func(y) are real-valued, :479-489).  PG(1, z): Polson, Scott & Windle
// (2013) Devroye-style alternating-series sampler (truncation t = 0.64).  PG(b, z) for any b > 0 by infinite divisibility:
// floor(b) <= 64 EXACT draws of PG(1, z) plus, for the fractional part only (or for the whole of b > 64 -- counts that large are rare and
// the cost of the exact sum grows with b), the sum-of-gammas representation  omega = 1/(2 pi^2) sum_k g_k / ((k - 1/2)^2 + z^2 / (4 pi^2)),
// g_k ~ Gamma(b, 1), truncated at 32 terms with the remainder drawn as ONE gamma variate matched to the remainder's exact mean and
// variance (it carries 0.6 % of the mean; its third cumulant is off by 1e-9 of the total) -- the third-party sampler itself truncates
// the same series, uncorrected, for b < 1.
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

// Gamma(alpha, 1), alpha > 0: Marsaglia & Tsang (2000), without the squeeze step; alpha < 1 through Gamma(alpha + 1) U^(1/alpha)
__device__ __forceinline__ double pgl_gamma(double alpha, PglPhilox& r) {
    double boost = 1.0;
    if (alpha < 1.0) { boost = exp(log(pgl_unif(r)) / alpha); alpha += 1.0; }
    const double d = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do { x = pgl_norm(r); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        if (log(pgl_unif(r)) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost;
    }
}

#define PGL_PG_SERIES_TERMS 32
#define PGL_PG_DEVROYE_MAX 64

// sum_{k > K} ((k - 1/2)^2 + c)^-p for p = 1, 2 by the midpoint-rule (Euler-Maclaurin) identity
//     sum_{k > K} phi(k - 1/2) = int_K^inf phi + phi'(K) / 24 - 7 phi'''(K) / 5760 + O(phi^(5)(K)):   relative error < 2e-9 at K = 32
__device__ __forceinline__ void pgl_pg_tail_sums(double c, double& S1, double& S2) {
    const double K = (double)PGL_PG_SERIES_TERMS, q = K * K + c, sc = sqrt(c), iq = 1.0 / q, iq2 = iq * iq;
    const double at = sc > 1e-6 * K ? atan(sc / K) / sc : 1.0 / K - c / (3.0 * K * K * K);          // int_K^inf dx / (x^2 + c)
    S1 = at + (-2.0 * K * iq2) * (1.0 / 24.0) - (-24.0 * K * (K * K - c) * iq2 * iq2) * (7.0 / 5760.0);
    const double i2 = c > 1e-3 * K * K ? (at - K * iq) / (2.0 * c)                                   // int_K^inf dx / (x^2 + c)^2
                                       : 1.0 / (3.0 * K * K * K) - 2.0 * c / (5.0 * K * K * K * K * K) + 3.0 * c * c / (7.0 * K * K * K * K * K * K * K);
    S2 = i2 + (-4.0 * K * iq2 * iq) * (1.0 / 24.0) - (72.0 * K * iq2 * iq2 - 192.0 * K * K * K * iq2 * iq2 * iq) * (7.0 / 5760.0);
}

__device__ __forceinline__ double pgl_pg_series(double b, double z, PglPhilox& r) {
    const double c = z * z * (1.0 / (4.0 * PGL_PI * PGL_PI));
    double s = 0.0;
    for (int k = 1; k <= PGL_PG_SERIES_TERMS; ++k) s += pgl_gamma(b, r) / ((k - 0.5) * (k - 0.5) + c);
    double S1, S2;
    pgl_pg_tail_sums(c, S1, S2);
    const double m = b * S1, v = b * S2;                       // mean and variance of the remainder (in units of 1 / (2 pi^2))
    s += (v / m) * pgl_gamma(m * m / v, r);
    return s * (1.0 / (2.0 * PGL_PI * PGL_PI));
}

// PG(b, z), b >= 0 real (Bernoulli b = 1; negative-binomial b = y + xi)
__device__ __forceinline__ double pgl_pg_draw(double b, double z, uint64_t seed, uint64_t stream, uint64_t elem) {
    PglPhilox r;
    pgl_rng_init(r, seed, stream, elem, PGL_PURPOSE_PG);
    if (!(b > 0.0)) return 0.0;
    if (b > (double)PGL_PG_DEVROYE_MAX) return pgl_pg_series(b, z, r);
    const double fl = floor(b), frac = b - fl;
    double s = 0.0;
    for (int k = 0; k < (int)fl; ++k) s += pgl_pg1(z, r);
    if (frac > 0.0) s += pgl_pg_series(frac, z, r);
    return s;
}
