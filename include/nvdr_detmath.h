/* nvdr_detmath.h -- deterministic single-precision elementary functions.
 *
 * Why this exists.  The Monte-Carlo shader of the reference turns uniform numbers into sample
 * directions through sincos / acos / atan2 (render/optixutils/c_src/envsampling/kernel.cu:57-79,
 * 124-138, 241-266) and then takes DISCRETE decisions from the result: the nearest env-map texel
 * (kernel.cu:177-178,197-199) and the shadow-ray hit/miss (kernel.cu:101-118).  glibc's libm and
 * ROCm's ocml differ in the last bit of those functions, which flips a texel or a visibility bit
 * in roughly one sample per 1e5 -- far above the 1e-4 per-pixel parity bar.  Both the CPU oracle
 * (plain C, gcc) and the gfx950 kernels therefore evaluate these functions through THIS header,
 * which uses nothing but correctly rounded IEEE-754 operations (+ - * / sqrt fma): the results are
 * bit-identical on the host and on the device as long as neither compiler contracts a*b+c on its
 * own (both are built with -ffp-contract=off; every fused operation below is an explicit fmaf).
 *
 * Accuracy (checked in tests/test_detmath.py against double-precision libm): <= 2 ulp for
 * sincos on [-4pi, 4pi], <= 3 ulp for acos on [-1, 1] and atan2.
 *
 * The polynomial coefficients are the classic public-domain Cephes single-precision minimax sets.
 */
#ifndef NVDR_DETMATH_H
#define NVDR_DETMATH_H

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NVDR_HD __host__ __device__ static inline
#else
#include <math.h>
#define NVDR_HD static inline
#endif

/* Correctly rounded forms.  On gfx950 hipcc's plain `/` and sqrtf() are IEEE-correct by default
 * (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn() is NOT (it maps to the native
 * approximate square root), so it must not be used here. */
#define NVDR_DIV(a, b) ((a) / (b))
#define NVDR_SQRT(a) sqrtf((a))
#define NVDR_FMA(a, b, c) fmaf((a), (b), (c))
#define NVDR_RINT(a) rintf((a))

#define NVDR_PI_F 3.14159265358979323846f
#define NVDR_PI_D 3.14159265358979323846

/* sin and cos of x for |x| up to a few multiples of pi (3-term Cody-Waite reduction by pi/2). */
NVDR_HD void nvdr_sincosf(float x, float *s, float *c)
{
    const float k = NVDR_RINT(x * 0.636619772367581343f); /* x * 2/pi */
    /* pi/2 split in three parts, the first two with trailing zero bits so k*part is exact */
    float r = NVDR_FMA(-k, 1.5703125f, x);
    r = NVDR_FMA(-k, 4.837512969970703125e-4f, r);
    r = NVDR_FMA(-k, 7.54978995489188216e-8f, r);
    const float z = r * r;
    /* sin(r), cos(r) on [-pi/4, pi/4] */
    float ps = NVDR_FMA(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = NVDR_FMA(ps, z, -1.6666654611e-1f);
    const float sr = NVDR_FMA(ps * z, r, r);
    float pc = NVDR_FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = NVDR_FMA(pc, z, 4.166664568298827e-2f);
    const float cr = NVDR_FMA(pc * z, z, NVDR_FMA(-0.5f, z, 1.0f));
    const int q = ((int)k) & 3;
    const float ss = (q & 1) ? cr : sr;
    const float cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

/* asin on [-0.5, 0.5] */
NVDR_HD float nvdr_asin_core(float x)
{
    const float z = x * x;
    float p = NVDR_FMA(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = NVDR_FMA(p, z, 4.5470025998e-2f);
    p = NVDR_FMA(p, z, 7.4953002686e-2f);
    p = NVDR_FMA(p, z, 1.6666752422e-1f);
    return NVDR_FMA(p * z, x, x);
}

/* acos(x), x clamped by the caller to [-1, 1] */
NVDR_HD float nvdr_acosf(float x)
{
    if (x > 0.5f) {
        const float t = NVDR_SQRT(0.5f * (1.0f - x));
        return 2.0f * nvdr_asin_core(t);
    }
    if (x < -0.5f) {
        const float t = NVDR_SQRT(0.5f * (1.0f + x));
        return NVDR_PI_F - 2.0f * nvdr_asin_core(t);
    }
    return 1.57079632679489661923f - nvdr_asin_core(x);
}

/* atan(x) for x >= 0 */
NVDR_HD float nvdr_atan_pos(float x)
{
    float y0;
    if (x > 2.414213562373095f) {          /* tan(3pi/8) */
        y0 = 1.57079632679489661923f;
        x = -NVDR_DIV(1.0f, x);
    } else if (x > 0.4142135623730950f) {  /* tan(pi/8) */
        y0 = 0.78539816339744830962f;
        x = NVDR_DIV(x - 1.0f, x + 1.0f);
    } else {
        y0 = 0.0f;
    }
    const float z = x * x;
    float p = NVDR_FMA(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = NVDR_FMA(p, z, 1.99777106478e-1f);
    p = NVDR_FMA(p, z, -3.33329491539e-1f);
    return y0 + NVDR_FMA(p * z, x, x);
}

/* atan2(y, x) with the usual quadrant conventions; atan2(0, 0) = 0 */
NVDR_HD float nvdr_atan2f(float y, float x)
{
    const float ay = y < 0.0f ? -y : y;
    const float ax = x < 0.0f ? -x : x;
    float a;
    if (ax == 0.0f && ay == 0.0f)
        a = 0.0f;
    else if (ax == 0.0f)
        a = 1.57079632679489661923f;
    else
        a = nvdr_atan_pos(NVDR_DIV(ay, ax));
    if (x < 0.0f)
        a = NVDR_PI_F - a;
    return y < 0.0f ? -a : a;
}

#endif /* NVDR_DETMATH_H */
