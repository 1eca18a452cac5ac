// Device-side counter-based random stream + Polya-gamma PG(b, z) sampler for gfx950.
//
// Stands where the reference calls the third-party `pypolyagamma.pgdrawvpar`
// (/root/reference/pyglm/regression.py:501-508; shapes b = b_func(y) are real-valued, :479-489).  PG(1, z): Polson, Scott & Windle
// (2013) Devroye-style alternating-series sampler (truncation t = 0.64).  PG(b, z) for any b > 0 by infinite divisibility:
// 1 <= b <= 64 EXACTLY -- floor(b) - 1 draws of PG(1, z) plus one draw of PG(1 + frac(b), z) from Windle's alternate sampler (pgl_pg_alt;
// floor(b) Devroye draws when b is an integer); for b < 1 and for the whole of b > 64 (counts that large are rare and
// the cost of the exact sum grows with b) the sum-of-gammas representation  omega = 1/(2 pi^2) sum_k g_k / ((k - 1/2)^2 + z^2 / (4 pi^2)),
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
#define PGL_PG_MAX_TRIALS 10000     // bound of every rejection loop (BayesLogit's own bound); never met on finite input
#define PGL_PG_MAX_INNER 1000       // bound of the alternating partial sums (they decide within a few terms)
#define PGL_PG_FRAC_EPS 1e-9        // a shape within 1e-9 of an integer is that integer (y + xi with xi = 1.1 * 1.1 / 1.21)
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
        for (int i = 0; i < PGL_PG_MAX_TRIALS && pgl_unif(r) > alpha; ++i) {
            double E1 = pgl_expon(r), E2 = pgl_expon(r);
            for (int q = 0; q < PGL_PG_MAX_TRIALS && E1 * E1 > 2 * E2 / t; ++q) { E1 = pgl_expon(r); E2 = pgl_expon(r); }
            X = 1 + E1 * t;
            X = t / (X * X);
            alpha = exp(-0.5 * Z * Z * X);
        }
    } else {
        const double mu = 1.0 / Z;
        for (int i = 0; i < PGL_PG_MAX_TRIALS && X > t; ++i) {
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
    for (int trial = 0; trial < PGL_PG_MAX_TRIALS; ++trial) {
        double X;
        if (pgl_unif(r) < mass) X = PGL_PG_TRUNC + pgl_expon(r) / fz;
        else X = pgl_pg_rtigauss(Z, r);
        double S = pgl_pg_a(0, X);
        const double Y = pgl_unif(r) * S;
        for (int n = 1; n < PGL_PG_MAX_INNER; ++n) {
            if (n & 1) { S -= pgl_pg_a(n, X); if (Y <= S) return 0.25 * X; }
            else       { S += pgl_pg_a(n, X); if (Y > S) break; }
        }
    }
    return __longlong_as_double(0x7ff8000000000000LL);
}

// Gamma(alpha, 1), alpha > 0: Marsaglia & Tsang (2000), without the squeeze step; alpha < 1 through Gamma(alpha + 1) U^(1/alpha)
__device__ __forceinline__ double pgl_gamma(double alpha, PglPhilox& r) {
    double boost = 1.0;
    if (alpha < 1.0) { boost = exp(log(pgl_unif(r)) / alpha); alpha += 1.0; }
    const double d = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int trial = 0; trial < PGL_PG_MAX_TRIALS; ++trial) {
        double x, v;
        do { x = pgl_norm(r); v = 1.0 + c * x; } while (v <= 0.0);      // P(v <= 0) < 1e-3 per pass
        v = v * v * v;
        if (log(pgl_unif(r)) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost;
    }
    return __longlong_as_double(0x7ff8000000000000LL);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// PG(h, z) for 1 < h < 2, exactly: the "alternate" rejection sampler of Windle, Polson & Scott (2014, arXiv 1405.0506, Sec. 3).  In the
// scale x = 4 omega, z <- |z| / 2 the density is  cosh^h(z) e^{-z^2 x / 2} sum_n (-1)^n a_n(x | h)  with
//     a_n(x | h) = 2^h  Gamma(n + h) / (Gamma(n + 1) Gamma(h))  (2n + h) / sqrt(2 pi x^3)  exp(-(2n + h)^2 / (2x));
// proposal: below the switch point t(h) the n = 0 term (tilted by e^{-z^2 x / 2}: an inverse Gaussian(h / z, h^2) on (0, t]), above it
// (pi/2)^h x^(h-1) e^{-pi^2 x / 8} / Gamma(h) (tilted: a Gamma(h, pi^2/8 + z^2/2) on (t, inf)); a candidate is accepted or rejected by the
// alternating partial sums once the a_n decrease.  t(h) = where the two pieces cross, per 0.01 of h (tests/golden/make_pg_alt_table.py; any
// value inside the region where both pieces dominate the density gives an exact sampler -- checked there with 60-digit arithmetic).
// oracle/pg_oracle.c holds the separately written CPU version on the same stream: (u) piece; (E, u) per trial of the truncated gamma /
// of the normal tail; (u) per inverse-chi-square candidate or (N, u) per inverse-Gaussian candidate; (u) height under the envelope.
static __device__ const double pgl_pg_alt_trunc[101] = {
    0.6366, 0.6757, 0.7110, 0.7426, 0.7712, 0.7974, 0.8216, 0.8443, 0.8657, 0.8860,
    0.9054, 0.9240, 0.9420, 0.9594, 0.9763, 0.9928, 1.0088, 1.0245, 1.0399, 1.0550,
    1.0699, 1.0845, 1.0989, 1.1131, 1.1271, 1.1410, 1.1547, 1.1683, 1.1817, 1.1950,
    1.2081, 1.2212, 1.2342, 1.2471, 1.2598, 1.2725, 1.2851, 1.2977, 1.3101, 1.3225,
    1.3349, 1.3471, 1.3593, 1.3715, 1.3836, 1.3956, 1.4076, 1.4196, 1.4315, 1.4434,
    1.4552, 1.4670, 1.4788, 1.4905, 1.5021, 1.5138, 1.5254, 1.5370, 1.5486, 1.5601,
    1.5716, 1.5831, 1.5945, 1.6059, 1.6173, 1.6287, 1.6401, 1.6514, 1.6627, 1.6740,
    1.6853, 1.6965, 1.7078, 1.7190, 1.7302, 1.7414, 1.7526, 1.7637, 1.7748, 1.7860,
    1.7971, 1.8082, 1.8193, 1.8303, 1.8414, 1.8524, 1.8635, 1.8745, 1.8855, 1.8965,
    1.9075, 1.9184, 1.9294, 1.9404, 1.9513, 1.9622, 1.9732, 1.9841, 1.9950, 2.0059,
    2.0168};

__device__ __forceinline__ double pgl_gamma_q(double a, double x) {        // regularized upper incomplete gamma, 1 <= a <= 2
    const double lead = exp(a * log(x) - x - lgamma(a));
    if (x < a + 1.0) {                                                      // series for P
        double term = 1.0 / a, sum = term;
        for (int n = 1; n < 500 && term >= sum * 1e-17; ++n) { term *= x / (a + n); sum += term; }
        return 1.0 - lead * sum;
    }
    double b = x + 1.0 - a, c = 1e300, d = 1.0 / b, f = d;                 // continued fraction (modified Lentz)
    for (int i = 1; i < 500; ++i) {
        const double an = -(double)i * ((double)i - a);
        b += 2.0;
        d = an * d + b;  if (fabs(d) < 1e-300) d = 1e-300;
        c = b + an / c;  if (fabs(c) < 1e-300) c = 1e-300;
        d = 1.0 / d;
        const double del = d * c;
        f *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return lead * f;
}

__device__ __forceinline__ double pgl_pg_alt(double h, double zpg, PglPhilox& r) {
    const double z = 0.5 * fabs(zpg);
    int k = (int)floor((h - 1.0) * 100.0);
    k = k < 0 ? 0 : k > 100 ? 100 : k;
    const double t = pgl_pg_alt_trunc[k], rt = sqrt(t);
    const double rate = 0.125 * PGL_PI * PGL_PI + 0.5 * z * z;
    const double h2 = h * 0.69314718055994530942;                                  // h log 2
    // masses of the two proposal pieces (common factor cosh^h z dropped)
    const double m_left = exp(h2 - h * z + pgl_log_pnorm((t * z - h) / rt)) + exp(h2 + h * z + pgl_log_pnorm(-(t * z + h) / rt));
    const double m_right = exp(h * log(0.5 * PGL_PI / rate)) * pgl_gamma_q(h, rate * t);
    const double p_right = m_right / (m_left + m_right);
    const double coef = exp(h2 - 0.5 * log(2.0 * PGL_PI)), lgh = lgamma(h);
    // constants of the two proposal samplers
    const double gb = rate * t, gd = gb - h, ge = h - 1.0, gs = sqrt(gd * gd + 4.0 * gb), gc = 0.5 * (gd + gs) / gb;   // truncated gamma (Dagpunar 1978)
    const double g1c = 2.0 * ge / (gb + h + gs);                                   // 1 - gc without the cancellation (gc -> 1 as h -> 1)
    const double glm = ge * log(0.5 * (gb + h + gs)) - ge;                         // = ge log(ge / (1 - gc)) - ge
    const bool tail = z * t < h;                                                   // inverse-Gaussian mean h / z beyond t
    const double nc = h / rt, na = 0.5 * (nc + sqrt(nc * nc + 4.0));               // normal tail beyond nc (Robert 1995)
    const double mu = tail ? 0.0 : h / z, lam = h * h;
    // every loop is bounded (PGL_PG_MAX_TRIALS; the acceptance rates are > 0.3, so a bound of 10 000 is never met on finite input):
    // an exhausted loop returns NaN, which the posterior system's status word reports -- a lane must never spin
    for (int trial = 0; trial < PGL_PG_MAX_TRIALS; ++trial) {
        double X = -1.0;
        if (pgl_unif(r) < p_right) {
            for (int i = 0; i < PGL_PG_MAX_TRIALS; ++i) {
                const double x = gb + pgl_expon(r) / gc;
                const double u = pgl_unif(r);
                if (log(u) <= ge * log(x) - x * g1c - glm) { X = t * (x / gb); break; }
            }
        } else if (tail) {
            for (int i = 0; i < PGL_PG_MAX_TRIALS; ++i) {
                double G = 0.0;
                for (int q = 0; q < PGL_PG_MAX_TRIALS; ++q) { G = nc + pgl_expon(r) / na; if (pgl_unif(r) <= exp(-0.5 * (G - na) * (G - na))) break; }
                const double Xc = lam / (G * G);
                if (pgl_unif(r) <= exp(-0.5 * z * z * Xc)) { X = Xc; break; }
            }
        } else {
            for (int i = 0; i < PGL_PG_MAX_TRIALS; ++i) {
                const double N = pgl_norm(r), Y = N * N;
                double Xc = mu + 0.5 * mu * mu * Y / lam - 0.5 * mu / lam * sqrt(4.0 * mu * lam * Y + mu * mu * Y * Y);
                if (pgl_unif(r) > mu / (mu + Xc)) Xc = mu * mu / Xc;
                if (Xc <= t) { X = Xc; break; }
            }
        }
        if (!(X > 0.0)) break;
        const double lx = 1.5 * log(X), ix = 0.5 / X;
        double cn = 1.0;                                                           // Gamma(n + h) / (Gamma(n + 1) Gamma(h))
        double S = coef * exp(log(h) - lx - h * h * ix), prev = S;
        const double env = X > t ? exp(h * log(0.5 * PGL_PI) + ge * log(X) - 0.125 * PGL_PI * PGL_PI * X - lgh) : S;
        const double Y = pgl_unif(r) * env;
        for (int n = 1; n < PGL_PG_MAX_INNER; ++n) {
            cn *= (n + ge) / n;
            const double d = 2.0 * n + h;
            const double an = coef * cn * exp(log(d) - lx - d * d * ix);
            const bool dec = an <= prev;
            prev = an;
            if (n & 1) { S -= an; if (Y <= S && dec) return 0.25 * X; }
            else { S += an; if (Y > S && dec) break; }
        }
    }
    return __longlong_as_double(0x7ff8000000000000LL);
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
    if (!(fabs(z) <= 1.7976931348623157e308)) return __longlong_as_double(0x7ff8000000000000LL);   // psi = nan / +-inf: no draw, and no loop to spin in
    if (b > (double)PGL_PG_DEVROYE_MAX) return pgl_pg_series(b, z, r);
    double fl = floor(b), frac = b - fl;
    if (fl < 1.0) return pgl_pg_series(frac, z, r);
    if (frac < PGL_PG_FRAC_EPS) frac = 0.0;                                 // rounding-level fractional parts: the integer shape
    else if (frac > 1.0 - PGL_PG_FRAC_EPS) { frac = 0.0; fl += 1.0; }
    // exact for every 1 <= b <= 64: whole Devroye draws, and the fractional part as ONE draw of PG(1 + frac, z) from the alternate sampler
    const int whole = frac > 0.0 ? (int)fl - 1 : (int)fl;
    double s = 0.0;
    for (int k = 0; k < whole; ++k) s += pgl_pg1(z, r);
    if (frac > 0.0) s += pgl_pg_alt(1.0 + frac, z, r);
    return s;
}
