/*
 * oracle/pg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C + libm) of the Polya-gamma draw that the reference obtains from the
 * third-party `pypolyagamma` package at /root/reference/pyglm/regression.py:474-477 (sampler
 * construction) and :501-508 (`ppg.pgdrawvpar(self.ppgs, b_func(y), psi, omega)`).
 *
 * `pypolyagamma` is NOT vendored under /root/reference and is not installed in this image
 * (setup.py:13 lists it un-pinned), so this file restates the PUBLISHED algorithm it wraps:
 * Polson, Scott & Windle (2013), "Bayesian inference for logistic models using Polya-Gamma latent
 * variables", JASA 108 -- the Devroye-style alternating-series sampler for PG(1, z) (their
 * Algorithm in Sec. 4 / supplement, as implemented in Windle's BayesLogit `PolyaGamma.h`):
 *     truncation point t = 0.64, proposal = mixture of a right-truncated inverse-Gaussian
 *     (x <= t) and a left-truncated exponential (x > t), acceptance by the alternating partial
 *     sums S_n of the Jacobi-theta coefficients a_n(x);  PG(b, z), integer b = sum of b PG(1, z).
 * PARITY UNPINNED at this boundary: the reference's tests hold no golden PG vector and the
 * third-party RNG (GSL MT19937 seeded from npr.randint(2**16)) cannot be reproduced here.  The
 * draw is therefore pinned by analytic known answers (tests/test_oracle_pg.py: mean, variance,
 * Laplace transform, KS against the sum-of-gammas series) and the GPU kernel is pinned against
 * THIS file on a shared counter-based random stream, specified below.
 *
 * Random stream specification (shared by the HIP kernel, implemented separately there):
 *   Philox4x32-10 (Salmon et al. 2011), key = (seed lo32, seed hi32),
 *   counter = (j | purpose<<24, element index i (32 bit), stream lo32, stream hi32)
 *   where j = 0,1,2,... counts Philox calls inside ONE draw; in the model element = time bin t,
 *   stream = (global neuron index n, sweep s).  One Philox call yields two uniforms:
 *       u_a = ((x0 | x1<<32) >> 11) + 0.5) * 2^-53,  u_b likewise from (x2, x3);  u_a is used first.
 *   exponential  E = -log(u);   normal  N = sqrt(-2 log u1) * cos(2 pi u2)  (both uniforms consumed).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define PG_TRUNC 0.64
#define PG_PI 3.141592653589793238462643383279502884
#define PG_PURPOSE_PG 1u

/* ------------------------------------------------------------------ Philox4x32-10 */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k[0];
    const uint32_t n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

typedef struct {
    uint32_t key[2];
    uint32_t elem, s_lo, s_hi;
    uint32_t j;       /* Philox calls made so far in this draw */
    uint32_t purpose;
    double buf[2];
    int have;
} pg_rng;

static inline double u64_to_unit(uint64_t x) { return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

static double rng_unif(pg_rng* r) {
    if (r->have == 0) {
        uint32_t c[4] = {r->j | (r->purpose << 24), r->elem, r->s_lo, r->s_hi}, o[4];
        oracle_philox4x32_10(c, r->key, o);
        r->j++;
        r->buf[0] = u64_to_unit((uint64_t)o[0] | ((uint64_t)o[1] << 32));
        r->buf[1] = u64_to_unit((uint64_t)o[2] | ((uint64_t)o[3] << 32));
        r->have = 2;
    }
    const double u = r->buf[2 - r->have];
    r->have--;
    return u;
}
static double rng_expon(pg_rng* r) { return -log(rng_unif(r)); }
static double rng_norm(pg_rng* r) {
    const double u1 = rng_unif(r), u2 = rng_unif(r);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * PG_PI * u2);
}

/* raw words for stream-parity tests: out[4*i..4*i+3] = philox(counter (j, elem0+i, s_lo, s_hi)) */
void oracle_philox_stream(uint64_t seed, uint32_t purpose, uint32_t j, uint64_t elem0, uint64_t stream,
                          uint32_t* out, size_t n) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (size_t i = 0; i < n; ++i) {
        uint32_t c[4] = {j | (purpose << 24), (uint32_t)(elem0 + i), (uint32_t)stream, (uint32_t)(stream >> 32)};
        oracle_philox4x32_10(c, key, out + 4 * i);
    }
}

/* ------------------------------------------------------------------ PG(1, z) pieces */
static double log_pnorm(double x) { return log(0.5 * erfc(-x * 0.70710678118654752440)); }

/* Jacobi-theta coefficient a_n(x) */
static double pg_a(int n, double x) {
    const double K = (n + 0.5) * PG_PI;
    if (x > PG_TRUNC) return K * exp(-0.5 * K * K * x);
    if (x > 0) {
        const double expnt = -1.5 * (log(0.5 * PG_PI) + log(x)) + log(K) - 2.0 * (n + 0.5) * (n + 0.5) / x;
        return exp(expnt);
    }
    return 0.0;
}

/* probability that the proposal comes from the truncated-exponential (right) piece */
static double pg_mass_texpon(double Z) {
    const double t = PG_TRUNC;
    const double fz = 0.125 * PG_PI * PG_PI + 0.5 * Z * Z;
    const double b = sqrt(1.0 / t) * (t * Z - 1);
    const double a = sqrt(1.0 / t) * (t * Z + 1) * -1.0;
    const double x0 = log(fz) + fz * t;
    const double xb = x0 - Z + log_pnorm(b);
    const double xa = x0 + Z + log_pnorm(a);
    const double qdivp = 4 / PG_PI * (exp(xb) + exp(xa));
    return 1.0 / (1.0 + qdivp);
}

/* inverse-Gaussian(mu = 1/Z, lambda = 1) truncated to (0, t] */
static double pg_rtigauss(double Z, pg_rng* r) {
    const double t = PG_TRUNC;
    double X = t + 1.0;
    Z = fabs(Z);
    if (1.0 / t > Z) { /* mu > t: rejection from the Levy-type proposal */
        double alpha = 0.0;
        for (int i = 0; i < 10000 && rng_unif(r) > alpha; ++i) {
            double E1 = rng_expon(r), E2 = rng_expon(r);
            for (int q = 0; q < 10000 && E1 * E1 > 2 * E2 / t; ++q) { E1 = rng_expon(r); E2 = rng_expon(r); }
            X = 1 + E1 * t;
            X = t / (X * X);
            alpha = exp(-0.5 * Z * Z * X);
        }
    } else { /* mu <= t: Michael-Schucany-Haas, retried until X <= t */
        const double mu = 1.0 / Z;
        for (int i = 0; i < 10000 && X > t; ++i) {
            double Y = rng_norm(r);
            Y *= Y;
            const double half_mu = 0.5 * mu, mu_Y = mu * Y;
            X = mu + half_mu * mu_Y - half_mu * sqrt(4 * mu_Y + mu_Y * mu_Y);
            if (rng_unif(r) > mu / (mu + X)) X = mu * mu / X;
        }
    }
    return X;
}

static double pg_draw_one(double z, pg_rng* r) {
    const double Z = fabs(z) * 0.5;
    const double fz = 0.125 * PG_PI * PG_PI + 0.5 * Z * Z;
    for (int trial = 0; trial < 10000; ++trial) {
        double X;
        if (rng_unif(r) < pg_mass_texpon(Z)) X = PG_TRUNC + rng_expon(r) / fz;
        else X = pg_rtigauss(Z, r);
        double S = pg_a(0, X);
        const double Y = rng_unif(r) * S;
        for (int n = 1; n < 1000; ++n) {
            if (n & 1) { S -= pg_a(n, X); if (Y <= S) return 0.25 * X; }
            else       { S += pg_a(n, X); if (Y > S) break; }
        }
    }
    return NAN;
}

/* ------------------------------------------------------------------ PG(h, z) for 1 < h < 2: Windle's "alternate" sampler
 * Windle, Polson & Scott (2014), "Sampling Polya-Gamma random variates: alternate and approximate techniques" (arXiv 1405.0506), Sec. 3 -- the
 * exact rejection sampler behind BayesLogit's PolyaGammaAlt for 1 <= h <= 4, which pypolyagamma's hybrid sampler uses for small non-integer
 * shapes.  Restated from the paper's formulas.  With J*(h, z) = 4 PG(h, 2z):
 *     density   f(x | h, z) = cosh^h(z) exp(-z^2 x / 2) sum_{n >= 0} (-1)^n a_n(x | h),
 *               a_n(x | h) = 2^h  Gamma(n + h) / (Gamma(n + 1) Gamma(h))  (2n + h) / sqrt(2 pi x^3)  exp(-(2n + h)^2 / (2x))
 *     envelope  x <= t:  a_0(x | h)                                           (tilted: an inverse Gaussian(mu = h / z, lambda = h^2) on (0, t])
 *               x >  t:  (pi/2)^h x^(h-1) exp(-pi^2 x / 8) / Gamma(h)         (tilted: a Gamma(h, rate pi^2/8 + z^2/2) on (t, inf))
 *     accept    by the alternating partial sums S_n once the a_n decrease (they are unimodal in n).
 * t = t(h) is where the two envelope pieces cross (2/pi = 0.6366 at h = 1: Devroye's 0.64), tabulated per 0.01 of h: pg_alt_trunc below, made by
 * tests/golden/make_pg_alt_table.py, which also checks with 60-digit arithmetic that both pieces dominate the density well beyond the switch
 * point.  Any switch point inside that region gives an EXACT sampler; the crossing gives the highest acceptance rate.
 * Stream consumption (the device code must match it draw for draw): u -> piece; right piece: (E, u) per trial of the truncated-gamma
 * rejection; left piece, mu > t: (E, u) per trial of the normal tail + one u per candidate, mu <= t: (N, u) per candidate; then u for the
 * height under the envelope. */
static const double pg_alt_trunc[101] = {
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

/* regularized upper incomplete gamma Q(a, x), a in [1, 2], x > 0: series below a + 1, continued fraction (modified Lentz) above */
static double gamma_q(double a, double x) {
    const double lead = exp(-x + a * log(x) - lgamma(a));
    if (x < a + 1.0) {
        double term = 1.0 / a, sum = term;
        for (int n = 1; n < 500; ++n) {
            term *= x / (a + n);
            sum += term;
            if (term < sum * 1e-17) break;
        }
        return 1.0 - lead * sum;
    }
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, hh = d;
    for (int i = 1; i < 500; ++i) {
        const double an = -(double)i * ((double)i - a);
        b += 2.0;
        d = an * d + b;
        if (fabs(d) < tiny) d = tiny;
        c = b + an / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        hh *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return lead * hh;
}

/* Gamma(shape, rate) restricted to (trunc, inf), shape > 1: Dagpunar's (1978) shifted-exponential rejection */
/* 1 - c0 is formed without the subtraction: as shape -> 1 the optimal rate c0 -> 1 and 1 - c0 = 2 (a - 1) / (b + a + root) would otherwise
 * round to 0 (log M = +inf: nothing is ever accepted).  Loops are bounded (PG_MAX_TRIALS, BayesLogit's bound); exhaustion gives NaN. */
#define PG_MAX_TRIALS 10000
#define PG_MAX_INNER 1000
#define PG_FRAC_EPS 1e-9
static double rng_ltgamma(double shape, double rate, double trunc, pg_rng* r) {
    const double a = shape, b = rate * trunc, d1 = b - a, d3 = a - 1.0;
    const double root = sqrt(d1 * d1 + 4.0 * b);
    const double c0 = 0.5 * (d1 + root) / b;
    const double one_m_c0 = 2.0 * d3 / (b + a + root);
    const double lM = d3 * log(0.5 * (b + a + root)) - d3;
    for (int i = 0; i < PG_MAX_TRIALS; ++i) {
        const double x = b + rng_expon(r) / c0;
        const double u = rng_unif(r);
        if (log(u) <= d3 * log(x) - x * one_m_c0 - lM) return trunc * (x / b);
    }
    return NAN;
}

/* inverse Gaussian(mu = h / z, lambda = h^2) restricted to (0, t] */
static double pg_alt_left(double h, double z, double t, pg_rng* r) {
    if (z * t < h) {            /* mu > t: x = h^2 / G^2 with G a standard normal beyond h / sqrt(t) (Robert 1995), thinned by exp(-z^2 x / 2) */
        const double c = h / sqrt(t), ar = 0.5 * (c + sqrt(c * c + 4.0));
        for (int i = 0; i < PG_MAX_TRIALS; ++i) {
            double G = 0.0;
            for (int q = 0; q < PG_MAX_TRIALS; ++q) {
                G = c + rng_expon(r) / ar;
                if (rng_unif(r) <= exp(-0.5 * (G - ar) * (G - ar))) break;
            }
            const double X = h * h / (G * G);
            if (rng_unif(r) <= exp(-0.5 * z * z * X)) return X;
        }
        return NAN;
    }
    const double mu = h / z, lam = h * h;
    for (int i = 0; i < PG_MAX_TRIALS; ++i) {                  /* Michael, Schucany & Haas (1976), retried until x <= t */
        const double N = rng_norm(r), Y = N * N;
        double X = mu + 0.5 * mu * mu * Y / lam - 0.5 * mu / lam * sqrt(4.0 * mu * lam * Y + mu * mu * Y * Y);
        if (rng_unif(r) > mu / (mu + X)) X = mu * mu / X;
        if (X <= t) return X;
    }
    return NAN;
}

static double pg_alt_a(int n, double x, double h, double coef, double* cn) {      /* a_n(x | h); *cn carries Gamma(n + h) / (Gamma(n + 1) Gamma(h)) */
    if (n == 0) *cn = 1.0; else *cn *= (n + h - 1.0) / n;
    const double d = 2.0 * n + h;
    return coef * *cn * exp(log(d) - 1.5 * log(x) - 0.5 * d * d / x);
}

static double pg_alt(double h, double zpg, pg_rng* r) {       /* one draw of PG(h, zpg), 1 < h < 2 */
    const double z = 0.5 * fabs(zpg);
    int k = (int)floor((h - 1.0) * 100.0);
    if (k < 0) k = 0;
    if (k > 100) k = 100;
    const double t = pg_alt_trunc[k];
    const double lam = 0.125 * PG_PI * PG_PI + 0.5 * z * z;
    const double st = sqrt(t), hl2 = h * log(2.0);
    const double wl = exp(hl2 - h * z + log_pnorm((t * z - h) / st)) + exp(hl2 + h * z + log_pnorm(-(t * z + h) / st));
    const double wr = exp(h * log(0.5 * PG_PI / lam)) * gamma_q(h, lam * t);
    const double pr = wr / (wl + wr);
    const double coef = exp(hl2 - 0.5 * log(2.0 * PG_PI));
    const double lgh = lgamma(h);
    for (int trial = 0; trial < PG_MAX_TRIALS; ++trial) {
        const double X = rng_unif(r) < pr ? rng_ltgamma(h, lam, t, r) : pg_alt_left(h, z, t, r);
        if (!(X > 0.0)) break;
        double cn, S = pg_alt_a(0, X, h, coef, &cn), prev = S;
        const double env = X > t ? exp(h * log(0.5 * PG_PI) + (h - 1.0) * log(X) - 0.125 * PG_PI * PG_PI * X - lgh) : S;
        const double Y = rng_unif(r) * env;
        for (int n = 1; n < PG_MAX_INNER; ++n) {
            const double an = pg_alt_a(n, X, h, coef, &cn);
            const int dec = an <= prev;
            prev = an;
            if (n & 1) { S -= an; if (Y <= S && dec) return 0.25 * X; }
            else       { S += an; if (Y > S && dec) break; }
        }
    }
    return NAN;
}

/* ------------------------------------------------------------------ PG(b, z) for real b > 0
 * Sum-of-gammas representation (Polson, Scott & Windle 2013, eq. 2):
 *     omega = 1/(2 pi^2) sum_{k>=1} g_k / ((k - 1/2)^2 + z^2/(4 pi^2)),   g_k ~ Gamma(b, 1) i.i.d.
 * The first PG_SERIES_TERMS terms are drawn; the remainder R = sum_{k>K} g_k / d_k is replaced by one gamma variate with R's exact mean
 * b sum 1/d_k and variance b sum 1/d_k^2 (here the two sums are taken term by term up to k = K + 4000 and closed by their integrals --
 * deliberately NOT the closed forms the device code uses).  Gamma variates: Marsaglia & Tsang (2000), "A simple method for
 * generating gamma variables", ACM TOMS 26, without the squeeze; shape < 1 by the U^(1/shape) boost.
 * By infinite divisibility: 1 <= b <= PG_DEVROYE_MAX is floor(b) - 1 exact draws of PG(1, z) plus ONE exact draw of PG(1 + frac(b), z) from
 * the alternate sampler above (floor(b) Devroye draws when b is an integer) -- no approximation for any b in [1, 64], which is where
 * pypolyagamma's own samplers are exact rejection samplers too.  The truncated series is left for b < 1 (pypolyagamma truncates the same
 * series there, at 200 terms, uncorrected) and for b > PG_DEVROYE_MAX (cost independent of b). */
#define PG_SERIES_TERMS 32
#define PG_DEVROYE_MAX 64

static double rng_gamma(double alpha, pg_rng* r) {
    double boost = 1.0;
    if (alpha < 1.0) { boost = exp(log(rng_unif(r)) / alpha); alpha += 1.0; }
    const double d = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int trial = 0; trial < 10000; ++trial) {
        double x, v;
        do { x = rng_norm(r); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        if (log(rng_unif(r)) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost;
    }
    return NAN;
}

static void pg_tail_sums(double c, double* S1, double* S2) {
    const int K = PG_SERIES_TERMS, K2 = PG_SERIES_TERMS + 4000;
    double s1 = 0.0, s2 = 0.0;
    for (int k = K2; k > K; --k) {                       /* small terms first */
        const double d = (k - 0.5) * (k - 0.5) + c;
        s1 += 1.0 / d;
        s2 += 1.0 / (d * d);
    }
    /* beyond K2: int_{K2}^inf dx/(x^2+c) and int dx/(x^2+c)^2 (midpoint rule; the neglected correction is O(K2^-3) of a 1e-4 share) */
    const double x = (double)K2, sc = sqrt(c);
    const double i1 = sc > 0 ? atan(sc / x) / sc : 1.0 / x;
    const double i2 = 1.0 / (3.0 * x * x * x) - 2.0 * c / (5.0 * x * x * x * x * x);
    *S1 = s1 + i1;
    *S2 = s2 + i2;
}

static double pg_series(double b, double z, pg_rng* r) {
    const double c = z * z / (4.0 * PG_PI * PG_PI);
    double s = 0.0;
    for (int k = 1; k <= PG_SERIES_TERMS; ++k) s += rng_gamma(b, r) / ((k - 0.5) * (k - 0.5) + c);
    double S1, S2;
    pg_tail_sums(c, &S1, &S2);
    const double m = b * S1, v = b * S2;
    s += (v / m) * rng_gamma(m * m / v, r);
    return s / (2.0 * PG_PI * PG_PI);
}

/* out[i] ~ PG(b[i], z[i]); b real, >= 0 (Bernoulli: 1; negative binomial: y + xi).
 * Stands where the reference calls pgdrawvpar (regression.py:504-507). returns 0, or -1 on bad b. */
int oracle_pg_draw(const double* b, const double* z, double* out, size_t len,
                   uint64_t seed, uint64_t stream, uint64_t elem0) {
    int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (ptrdiff_t i = 0; i < (ptrdiff_t)len; ++i) {
        const double bi = b ? b[i] : 1.0;
        if (!(bi >= 0) || bi > 1e9) { bad = 1; out[i] = NAN; continue; }
        pg_rng r;
        r.key[0] = (uint32_t)seed; r.key[1] = (uint32_t)(seed >> 32);
        r.elem = (uint32_t)(elem0 + (uint64_t)i);
        r.s_lo = (uint32_t)stream; r.s_hi = (uint32_t)(stream >> 32);
        r.j = 0; r.purpose = PG_PURPOSE_PG; r.have = 0;
        double s = 0.0;
        if (bi > 0.0 && !isfinite(z[i])) s = NAN;             /* psi = nan / +-inf: no draw (and no loop to spin in) */
        else if (bi > (double)PG_DEVROYE_MAX) s = pg_series(bi, z[i], &r);
        else if (bi > 0.0) {
            double fl = floor(bi), frac = bi - fl;
            if (fl >= 1.0 && frac < PG_FRAC_EPS) frac = 0.0;   /* shapes within 1e-9 of an integer (y + xi with xi = 1.1 * 1.1 / 1.21) are that integer */
            else if (fl >= 1.0 && frac > 1.0 - PG_FRAC_EPS) { frac = 0.0; fl += 1.0; }
            if (fl < 1.0) s = pg_series(frac, z[i], &r);
            else {
                const int whole = frac > 0.0 ? (int)fl - 1 : (int)fl;
                for (int k = 0; k < whole; ++k) s += pg_draw_one(z[i], &r);
                if (frac > 0.0) s += pg_alt(1.0 + frac, z[i], &r);
            }
        }
        out[i] = s;
    }
    return bad ? -1 : 0;
}

/* fused reference-shaped helper used by the cpu_baseline leg: psi -> omega for one neuron */
int oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
