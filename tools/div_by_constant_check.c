// div_by_constant_check.c -- exhaustive check over all finite floats x: does a double MULTIPLICATION by the rounded reciprocal of the
// constant give the same FLOAT result as the double DIVISION the reference's expressions perform (x / M_PI, x / (2 M_PI) + 0.5,
// fmax(1e-6f, x / M_PI))?  It does, for every x: csrc/env_shade.hip multiplies.   gcc -O2 -fopenmp -ffp-contract=off div_by_constant_check.c -lm
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#define PI 3.14159265358979323846
int main()
{
    const double inv_pi = 1.0 / PI, inv_2pi = 1.0 / (2.0 * PI);
    long long badA = 0, badB = 0, badC = 0, badA2 = 0, badB2 = 0, badC2 = 0, n = 0;
    float worstA = 0, worstB = 0, worstC = 0;
#pragma omp parallel for reduction(+:badA,badB,badC,badA2,badB2,badC2,n) schedule(static)
    for (long long bits = 0; bits < (1ll << 32); ++bits) {
        uint32_t u = (uint32_t)bits;
        float x;
        memcpy(&x, &u, 4);
        if (!(x == x) || isinf(x)) continue;
        n++;
        const double xd = (double)x;
        // C: (float)(x / pi)
        const float c0 = (float)(xd / PI), c1 = (float)(xd * inv_pi);
        if (memcmp(&c0, &c1, 4)) badC++;
        // A: (float)fmax(1e-6f, x / pi)
        const float a0 = (float)fmax((double)0.000001f, xd / PI), a1 = (float)fmax((double)0.000001f, xd * inv_pi);
        if (memcmp(&a0, &a1, 4)) badA++;
        // B: (float)(x / 2pi + 0.5)
        const float b0 = (float)(xd / (2.0 * PI) + 0.5), b1 = (float)(xd * inv_2pi + 0.5);
        if (memcmp(&b0, &b1, 4)) badB++;
        // Markstein-corrected quotients
        double q = xd * inv_pi; double e = fma(-q, PI, xd); double q2 = fma(e, inv_pi, q);
        const float c2 = (float)q2; if (memcmp(&c0, &c2, 4)) badC2++;
        const float a2 = (float)fmax((double)0.000001f, q2); if (memcmp(&a0, &a2, 4)) badA2++;
        q = xd * inv_2pi; e = fma(-q, 2.0 * PI, xd); q2 = fma(e, inv_2pi, q);
        const float b2 = (float)(q2 + 0.5); if (memcmp(&b0, &b2, 4)) badB2++;
    }
    printf("finite floats checked: %lld\n", n);
    printf("x * (1/pi)            vs x / pi           : %lld mismatches (with fmax(1e-6f, .): %lld)\n", badC, badA);
    printf("x * (1/2pi) + 0.5     vs x / 2pi + 0.5    : %lld mismatches\n", badB);
    printf("corrected quotient (2 extra fma): /pi %lld (fmax %lld), /2pi+0.5 %lld\n", badC2, badA2, badB2);
    return 0;
}
