/*
 * oracle/pg_oracle_selftest.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * A small driver of oracle/pg_oracle.c for the sanitizer build (`make -C oracle asan`): draws PG(b, z) over the shapes the model uses
 * (Bernoulli b = 1, negative-binomial b = y + xi incl. fractional and large b), checks the mean against the closed form
 * E[omega] = b / (2 z) tanh(z / 2), and the Philox words against the Random123 known-answer vector.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void oracle_philox_stream(uint64_t seed, uint32_t purpose, uint32_t j, uint64_t elem0, uint64_t stream, uint32_t* out, size_t n);
int oracle_pg_draw(const double* b, const double* z, double* out, size_t len, uint64_t seed, uint64_t stream, uint64_t elem0);

int main(void) {
    /* Random123 kat_vectors: philox4x32-10, counter = key = 0 and counter = key = ff..f */
    {
        const uint32_t c0[4] = {0, 0, 0, 0}, k0[2] = {0, 0}, want0[4] = {0x6627e8d5u, 0xe169c58du, 0xbc57ac4cu, 0x9b00dbd8u};
        const uint32_t c1[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, k1[2] = {0xffffffffu, 0xffffffffu};
        const uint32_t want1[4] = {0x408f276du, 0x41c83b0eu, 0xa20bc7c6u, 0x6d5451fdu};
        uint32_t o[4];
        oracle_philox4x32_10(c0, k0, o);
        for (int i = 0; i < 4; ++i) if (o[i] != want0[i]) { fprintf(stderr, "philox KAT 0 word %d: %08x != %08x\n", i, o[i], want0[i]); return 1; }
        oracle_philox4x32_10(c1, k1, o);
        for (int i = 0; i < 4; ++i) if (o[i] != want1[i]) { fprintf(stderr, "philox KAT 1 word %d: %08x != %08x\n", i, o[i], want1[i]); return 1; }
    }
    const size_t n = 20000;
    const double shapes[] = {1.0, 2.0, 0.3, 1.02, 1.5, 2.5, 7.0, 13.7, 50.0, 63.99, 70.5};      /* incl. the alternate sampler (fractional part of 1 < b < 64) */
    const double zs[] = {0.0, 0.7, 3.0, 12.0};
    double* b = (double*)malloc(n * sizeof(double));
    double* z = (double*)malloc(n * sizeof(double));
    double* out = (double*)malloc(n * sizeof(double));
    uint32_t* words = (uint32_t*)malloc(4 * 64 * sizeof(uint32_t));
    if (!b || !z || !out || !words) return 2;
    oracle_philox_stream(7, 1, 0, 0, 3, words, 64);
    int rc = 0;
    for (size_t si = 0; si < sizeof(shapes) / sizeof(shapes[0]) && !rc; ++si)
        for (size_t zi = 0; zi < sizeof(zs) / sizeof(zs[0]) && !rc; ++zi) {
            for (size_t i = 0; i < n; ++i) { b[i] = shapes[si]; z[i] = zs[zi]; }
            if (oracle_pg_draw(b, z, out, n, 11 + si, 5 + zi, 0) != 0) { fprintf(stderr, "oracle_pg_draw failed\n"); rc = 1; break; }
            double m = 0.0, v = 0.0;
            for (size_t i = 0; i < n; ++i) m += out[i];
            m /= (double)n;
            for (size_t i = 0; i < n; ++i) v += (out[i] - m) * (out[i] - m);
            v /= (double)(n - 1);
            const double zz = zs[zi];
            const double want = zz > 0 ? shapes[si] / (2.0 * zz) * tanh(0.5 * zz) : shapes[si] * 0.25;
            if (fabs(m - want) > 6.0 * sqrt(v / (double)n)) { fprintf(stderr, "PG(%g, %g): mean %g, expected %g\n", shapes[si], zz, m, want); rc = 1; }
        }
    /* b == NULL means b = 1; a negative shape is reported, not read out of bounds */
    if (!rc && oracle_pg_draw(NULL, z, out, 16, 1, 1, 0) != 0) rc = 1;
    b[0] = -1.0;
    if (!rc && oracle_pg_draw(b, z, out, 1, 1, 1, 0) != -1) rc = 1;
    free(b); free(z); free(out); free(words);
    if (!rc) printf("pg_oracle selftest ok (ASan/UBSan clean)\n");
    return rc;
}
