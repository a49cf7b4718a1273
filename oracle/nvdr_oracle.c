/* nvdr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the algorithm of nvdiffrecmc's Monte-Carlo direct-lighting path, used
 * only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the CHECKER of the
 * HIP kernels.  The product (nvdiffrecmc_amd/) never imports, links or calls anything in oracle/.
 *
 * What it follows (all paths relative to the reference checkout):
 *   env-shade raygen program  render/optixutils/c_src/envsampling/kernel.cu:30-542
 *   BSDF eval fwd/bwd         render/optixutils/c_src/bsdf.h:21-275
 *   vector helpers            render/optixutils/c_src/math_utils.h:80-162
 *   broadcast fetch           render/optixutils/c_src/common.h:13-27
 *   host launch / zero-init   render/optixutils/c_src/torch_bindings.cpp:123-272
 *   bilateral denoiser        render/optixutils/c_src/denoising.cu:14-130
 *   light pdf / CDF tables    render/light.py:46-59
 *
 * Parity status: PINNED against the reference itself -- oracle/_ref builds the reference's own
 * kernel.cu / bsdf.h / denoising.cu for the CPU through a small shim (oracle/ref_shim/, recipe
 * oracle/Makefile) and tests/test_oracle_pins.py compares the two in this container; the
 * resulting vectors are committed under tests/golden/.  Two things are ours by necessity:
 *   - shadow-ray visibility: closed-source OptiX in the reference (optixTrace, kernel.cu:104-116);
 *     here a brute-force loop over all triangles with the predicate of include/nvdr_raytri.h;
 *   - sin/cos/acos/atan2 come from include/nvdr_detmath.h instead of libm, so that the GPU
 *     kernels can reproduce every discrete decision (texel, visibility, lobe) bit for bit.
 *
 * Arithmetic contract: expressions are evaluated exactly as the reference source promotes them
 * (float, with the fp64 islands caused by CUDART_PI / unsuffixed literals -- SURVEY Appendix A.8),
 * left to right, with no fused multiply-add (build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "nvdr_hip.h"    /* nvdr_tensor, nvdr_env_shade_args: same structs, HOST pointers here */
#include "nvdr_detmath.h"
#include "nvdr_raytri.h"

#define PI_D 3.14159265358979323846   /* CUDART_PI: a double constant */
#define PI_F 3.14159265358979323846f  /* M_PI of bsdf.h:14-16: a float */

typedef struct { float x, y, z; } f3;

/* ---------------------------------------------------------------------------------------------
 * EXPERIMENT SWITCH (tools/fast_value_math.py; never defined in the checker build): -DORACLE_FAST_VALUE_MATH evaluates the
 * divisions whose result is only ever a VALUE -- BSDF evaluation and its adjoints, pdfs, the MIS weight -- the way a GPU fast path
 * would: a * rcp(b) with a reciprocal that is off by up to one ulp (perturbed pseudo-randomly here), float instead of the
 * fp64 islands of the reference.  Everything that feeds a DISCRETE decision (sample directions, texel indices, lobe choice, the
 * dead-sample gate, CDF inversion) keeps IEEE arithmetic.  The question it answers: how far do images and gradients move? */
#if ORACLE_FAST_VALUE_MATH
static inline float vrcp_(float b)
{
    float r = 1.0f / b;
    union { float f; uint32_t u; } x, y;
    x.f = b;
    y.f = r;
    uint32_t h = x.u * 2654435761u;
    h ^= h >> 15;
    if ((h & 3u) == 1u) y.u += 1u;            /* one ulp up / down for half of the operands */
    else if ((h & 3u) == 2u) y.u -= 1u;
    return y.f;
}
#if ORACLE_FAST_VALUE_MATH >= 3
/* levels 3 / 4 (round 5): a FAITHFUL division -- the quotient off by at most one ulp, what rcp + one Newton step + a residual correction
 * gives without the scaling / fix-up instructions of the IEEE sequence -- and the fp64 islands kept.  3: BSDF evaluation, its adjoints and
 * the MIS weight (stage 3 of the GPU pipeline); 4: the pdfs of stage 1 as well. */
static inline float vdiv1_(float a, float b)
{
    union { float f; uint32_t u; } q, h;
    q.f = a / b;
    h.f = a * 1.7f + b;
    uint32_t k = h.u * 2654435761u;
    k ^= k >> 15;
    if (q.f == q.f && q.f != 0.0f && (q.u & 0x7f800000u) != 0x7f800000u) {
        if ((k & 3u) == 1u) q.u += 1u;
        else if ((k & 3u) == 2u) q.u -= 1u;
    }
    return q.f;
}
#define VDIV(a, b) (ORACLE_FAST_VALUE_MATH >= 4 ? vdiv1_((a), (b)) : ((a) / (b)))
#define VFAST 0
#define VDIV_FAITHFUL(a, b) vdiv1_((a), (b))
#else
#define VDIV(a, b) ((a) * vrcp_(b))
#define VFAST 1
#endif
#else
#define VDIV(a, b) ((a) / (b))
#define VFAST 0
#endif
/* level 1: the pdfs and the MIS weight only; level 2: also the BSDF evaluation and its adjoints (VDIV_E / VFAST_E) */
#if ORACLE_FAST_VALUE_MATH >= 3
#define VDIV_E(a, b) VDIV_FAITHFUL(a, b)
#define VFAST_E 0
#elif ORACLE_FAST_VALUE_MATH >= 2
#define VDIV_E(a, b) VDIV(a, b)
#define VFAST_E 1
#else
#define VDIV_E(a, b) ((a) / (b))
#define VFAST_E 0
#endif

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 div3s(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
static inline float sum3(f3 a) { return a.x + a.y + a.z; }                                /* math_utils.h:80 */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         /* math_utils.h:85 */
static inline float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); } /* math_utils.h:13 */
static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

static inline void bwd_dot(f3 a, f3 b, f3 *d_a, f3 *d_b, float d_out)                       /* math_utils.h:87 */
{
    d_a->x += d_out * b.x; d_a->y += d_out * b.y; d_a->z += d_out * b.z;
    d_b->x += d_out * a.x; d_b->y += d_out * a.y; d_b->z += d_out * a.z;
}
static inline float luminance(f3 c) { return dot3(c, mk3(0.2126f, 0.7152f, 0.0722f)); }    /* math_utils.h:93 */
static inline f3 cross3(f3 a, f3 b)                                                         /* math_utils.h:98 */
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 safe_normalize(f3 v)                                                       /* math_utils.h:134 */
{
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return l > 0.0f ? div3s(v, l) : mk3(0, 0, 0);
}
/* the same where the result is only evaluated (BSDF half vector, view vector of the evaluation, pdf of a given direction) */
static inline f3 vdiv3s(f3 a, float s_) { return mk3(VDIV(a.x, s_), VDIV(a.y, s_), VDIV(a.z, s_)); }
static inline f3 safe_normalize_v(f3 v)
{
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return l > 0.0f ? vdiv3s(v, l) : mk3(0, 0, 0);
}
static inline f3 vdiv3s_e(f3 a, float s_) { return mk3(VDIV_E(a.x, s_), VDIV_E(a.y, s_), VDIV_E(a.z, s_)); }
static inline f3 safe_normalize_e(f3 v)
{
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return l > 0.0f ? vdiv3s_e(v, l) : mk3(0, 0, 0);
}
static inline void bwd_safe_normalize(f3 v, f3 *d_v, f3 d_out)                              /* math_utils.h:140 */
{
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (l > 0.0f) {
        float l2 = v.x * v.x + v.y * v.y + v.z * v.z;
        float fac = VFAST_E ? VDIV_E(1.0f, l2 * sqrtf(l2)) : (float)(1.0 / (double)(l2 * sqrtf(l2))); /* 1.0 / powf(l2, 1.5f) */
        d_v->x += (d_out.x * (v.y * v.y + v.z * v.z) - d_out.y * (v.x * v.y) - d_out.z * (v.x * v.z)) * fac;
        d_v->y += (d_out.y * (v.x * v.x + v.z * v.z) - d_out.x * (v.y * v.x) - d_out.z * (v.y * v.z)) * fac;
        d_v->z += (d_out.z * (v.x * v.x + v.y * v.y) - d_out.x * (v.z * v.x) - d_out.y * (v.z * v.y)) * fac;
    }
}
static inline void branchlessONB(f3 n, f3 *b1, f3 *b2)                                      /* math_utils.h:155 */
{
    float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    *b1 = mk3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    *b2 = mk3(b, sign + n.y * n.y * a, -n.y);
}
static inline float pow5f(float x) { float x2 = x * x; return x2 * x2 * x; } /* powf(x, 5.0f) */

/* ---------------------------------------------------------------------------------------------
 * strided / broadcast access (common.h:13-27) */
static inline const float *elem(const nvdr_tensor *t, long i0, long i1, long i2)
{
    const float *p = (const float *)t->data;
    return p + (t->size[0] == 1 ? 0 : i0 * t->stride[0]) + (t->size[1] == 1 ? 0 : i1 * t->stride[1]) +
           (t->size[2] == 1 ? 0 : i2 * t->stride[2]);
}
static inline f3 fetch3v(const nvdr_tensor *t, long i0, long i1, long i2)
{
    const float *p = elem(t, i0, i1, i2);
    if (t->size[3] == 1) return mk3(p[0], p[0], p[0]);
    return mk3(p[0], p[t->stride[3]], p[2 * t->stride[3]]);
}

/* ---------------------------------------------------------------------------------------------
 * RNG (kernel.cu:30-45) */
static inline uint32_t rand_pcg(uint32_t *s)
{
    uint32_t st = *s;
    uint32_t word = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    *s = st * 747796405u + 2891336453u;
    return (word >> 22u) ^ word;
}
static inline uint32_t hash_pcg(uint32_t a, uint32_t b) { return rand_pcg(&a) ^ rand_pcg(&b); }
static inline float uniform_pcg(uint32_t *s) { return (float)(rand_pcg(s) & 0xFFFFFF) / (float)0x1000000; }

/* ---------------------------------------------------------------------------------------------
 * BSDF evaluation, forward and backward (bsdf.h) */
#define SPECULAR_EPSILON 1e-4f

static float fwdLambert(f3 nrm, f3 wi) { return fmaxf(VDIV_E(dot3(nrm, wi), PI_F), 0.0f); }         /* bsdf.h:21 */
static void bwdLambert(f3 nrm, f3 wi, f3 *d_nrm, f3 *d_wi, float d_out)                      /* bsdf.h:26 */
{
    if (dot3(nrm, wi) > 0.0f) bwd_dot(nrm, wi, d_nrm, d_wi, VDIV_E(d_out, PI_F));
}
static f3 fwdFresnelSchlick3(f3 f0, f3 f90, float cosTheta)                                   /* bsdf.h:54 */
{
    float c = clampf(cosTheta, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float scale = pow5f(1.0f - c);
    return add3(scale3(f0, 1.0f - scale), scale3(f90, scale));
}
static void bwdFresnelSchlick3(f3 f0, f3 f90, float cosTheta, f3 *d_f0, f3 *d_f90, float *d_cos, f3 d_out) /* bsdf.h:61 */
{
    float c = clampf(cosTheta, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float scale = pow5f(fmaxf(1.0f - c, 0.0f));
    float oms = VFAST_E ? 1.0f - scale : (float)(1.0 - (double)scale);
    *d_f0 = add3(*d_f0, scale3(d_out, oms));
    *d_f90 = add3(*d_f90, scale3(d_out, scale));
    if (cosTheta >= SPECULAR_EPSILON && cosTheta < 1.0f - SPECULAR_EPSILON) {
        float omc = 1.0f - cosTheta;
        float p4 = (omc * omc) * (omc * omc);
        *d_cos += sum3(scale3(scale3(mul3(d_out, sub3(f90, f0)), -5.0f), p4));
    }
}
static float fwdNdfGGX(float alphaSqr, float cosTheta)                                        /* bsdf.h:76 */
{
    float c = clampf(cosTheta, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float d = (c * alphaSqr - c) * c + 1.0f;
    return VDIV_E(alphaSqr, d * d * PI_F);
}
static void bwdNdfGGX(float alphaSqr, float cosTheta, float *d_alphaSqr, float *d_cos, float d_out) /* bsdf.h:83 */
{
    float c = clampf(cosTheta, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float c2 = c * c;
    float base = VFAST_E ? (alphaSqr - 1.0f) * c2 + 1.0f : (float)(((double)alphaSqr - 1.0) * (double)c2 + 1.0); /* (alphaSqr - 1.0) * cosThetaSqr + 1.0f */
    float cube = base * base * base;
    *d_alphaSqr += VDIV_E(d_out * (1.0f - (alphaSqr + 1.0f) * c2), PI_F * cube);
    if (cosTheta > SPECULAR_EPSILON && cosTheta < 1.0f - SPECULAR_EPSILON)
        *d_cos += VDIV_E(d_out * -(4.0f * (alphaSqr - 1.0f) * alphaSqr * cosTheta), PI_F * cube);
}
static float fwdLambdaGGX(float alphaSqr, float cosTheta)                                     /* bsdf.h:98 */
{
    float c = clampf(cosTheta, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float c2 = c * c;
    float tan2 = VFAST_E ? VDIV_E(1.0f - c2, c2) : (float)((1.0 - (double)c2) / (double)c2);
    return 0.5f * (sqrtf(1.0f + alphaSqr * tan2) - 1.0f);
}
static void bwdLambdaGGX(float alphaSqr, float cosTheta, float *d_alphaSqr, float *d_cos, float d_out) /* bsdf.h:107 */
{
    float c = clampf(cosTheta, SPECULAR_EPSILON, 1.0f - SPECULAR_EPSILON);
    float c2 = c * c;
    float tan2 = VFAST_E ? VDIV_E(1.0f - c2, c2) : (float)((1.0 - (double)c2) / (double)c2);
    *d_alphaSqr += VFAST_E ? VDIV_E(d_out * (0.25f * tan2), sqrtf(alphaSqr * tan2 + 1.0f))
                         : (float)((double)d_out * (0.25 * (double)tan2) / (double)sqrtf(alphaSqr * tan2 + 1.0f));
    if (cosTheta > SPECULAR_EPSILON && cosTheta < 1.0f - SPECULAR_EPSILON)
        *d_cos += VFAST_E ? VDIV_E(d_out * -(0.5f * alphaSqr), (c * c * c) * sqrtf(VDIV_E(alphaSqr, c2) - alphaSqr + 1.0f))
                        : (float)((double)d_out * -(0.5 * (double)alphaSqr) /
                                  (double)((c * c * c) * sqrtf(alphaSqr / c2 - alphaSqr + 1.0f)));
}
static float fwdMaskingSmith(float alphaSqr, float cosI, float cosO)                          /* bsdf.h:122 */
{
    return VDIV_E(1.0f, 1.0f + fwdLambdaGGX(alphaSqr, cosI) + fwdLambdaGGX(alphaSqr, cosO));
}
static void bwdMaskingSmith(float alphaSqr, float cosI, float cosO, float *d_alphaSqr, float *d_cosI, float *d_cosO,
                            float d_out)                                                       /* bsdf.h:129 */
{
    float lI = fwdLambdaGGX(alphaSqr, cosI), lO = fwdLambdaGGX(alphaSqr, cosO);
    float s = 1.0f + lI + lO;
    float d_l = VDIV_E(-d_out, s * s);
    bwdLambdaGGX(alphaSqr, cosI, d_alphaSqr, d_cosI, d_l);
    bwdLambdaGGX(alphaSqr, cosO, d_alphaSqr, d_cosO, d_l);
}
static f3 fwdPbrSpecular(f3 col, f3 nrm, f3 wo, f3 wi, float alpha, float min_roughness)       /* bsdf.h:144 */
{
    float _alpha = clampf(alpha, min_roughness * min_roughness, 1.0f);
    float alphaSqr = _alpha * _alpha;
    f3 h = safe_normalize_e(add3(wo, wi));
    float woDotN = dot3(wo, nrm), wiDotN = dot3(wi, nrm), woDotH = dot3(wo, h), nDotH = dot3(nrm, h);
    float D = fwdNdfGGX(alphaSqr, nDotH);
    float G = fwdMaskingSmith(alphaSqr, woDotN, wiDotN);
    f3 F = fwdFresnelSchlick3(col, mk3(1, 1, 1), woDotH);
    f3 w = vdiv3s_e(scale3(scale3(scale3(F, D), G), 0.25f), woDotN);
    int front = (woDotN > SPECULAR_EPSILON) & (wiDotN > SPECULAR_EPSILON);
    return front ? w : mk3(0, 0, 0);
}
static void bwdPbrSpecular(f3 col, f3 nrm, f3 wo, f3 wi, float alpha, float min_roughness, f3 *d_col, f3 *d_nrm,
                           f3 *d_wo, f3 *d_wi, float *d_alpha, f3 d_out)                      /* bsdf.h:164 */
{
    float _alpha = clampf(alpha, min_roughness * min_roughness, 1.0f);
    float alphaSqr = _alpha * _alpha;
    f3 h = safe_normalize_e(add3(wo, wi));
    float woDotN = dot3(wo, nrm), wiDotN = dot3(wi, nrm), woDotH = dot3(wo, h), nDotH = dot3(nrm, h);
    float D = fwdNdfGGX(alphaSqr, nDotH);
    float G = fwdMaskingSmith(alphaSqr, woDotN, wiDotN);
    f3 F = fwdFresnelSchlick3(col, mk3(1, 1, 1), woDotH);
    int front = (woDotN > SPECULAR_EPSILON) & (wiDotN > SPECULAR_EPSILON);
    if (!front) return;
    f3 d_F = vdiv3s_e(scale3(scale3(scale3(d_out, D), G), 0.25f), woDotN);
    float d_D = sum3(vdiv3s_e(scale3(scale3(mul3(d_out, F), G), 0.25f), woDotN));
    float d_G = sum3(vdiv3s_e(scale3(scale3(mul3(d_out, F), D), 0.25f), woDotN));
    float d_woDotN = -sum3(vdiv3s_e(scale3(scale3(scale3(mul3(d_out, F), D), G), 0.25f), woDotN * woDotN));
    f3 d_f90 = mk3(0, 0, 0);
    float d_woDotH = 0, d_wiDotN = 0, d_nDotH = 0, d_alphaSqr = 0;
    bwdFresnelSchlick3(col, mk3(1, 1, 1), woDotH, d_col, &d_f90, &d_woDotH, d_F);
    bwdMaskingSmith(alphaSqr, woDotN, wiDotN, &d_alphaSqr, &d_woDotN, &d_wiDotN, d_G);
    bwdNdfGGX(alphaSqr, nDotH, &d_alphaSqr, &d_nDotH, d_D);
    f3 d_h = mk3(0, 0, 0);
    bwd_dot(nrm, h, d_nrm, &d_h, d_nDotH);
    bwd_dot(wo, h, d_wo, &d_h, d_woDotH);
    bwd_dot(wi, nrm, d_wi, d_nrm, d_wiDotN);
    bwd_dot(wo, nrm, d_wo, d_nrm, d_woDotN);
    f3 d_h_un = mk3(0, 0, 0);
    bwd_safe_normalize(add3(wo, wi), &d_h_un, d_h);
    *d_wo = add3(*d_wo, d_h_un);
    *d_wi = add3(*d_wi, d_h_un);
    if (alpha > min_roughness * min_roughness) *d_alpha += d_alphaSqr * 2 * alpha;
}
/* demodulated-diffuse PBR BSDF of the shader (bsdf.h:222-236) */
static void fwdPbrBSDF(f3 kd, f3 arm, f3 pos, f3 nrm, f3 view_pos, f3 wi, float min_roughness, f3 *diffuse, f3 *specular)
{
    f3 wo = safe_normalize_e(sub3(view_pos, pos));
    float alpha = arm.y * arm.y;
    f3 spec_col = scale3(add3(scale3(mk3(0.04f, 0.04f, 0.04f), 1.0f - arm.z), scale3(kd, arm.z)), VFAST_E ? 1.0f - arm.x : (float)(1.0 - (double)arm.x));
    float diff = fwdLambert(nrm, wi);
    *diffuse = mk3(diff, diff, diff);
    *specular = fwdPbrSpecular(spec_col, nrm, wo, wi, alpha, min_roughness);
}
static void bwdPbrBSDF(f3 kd, f3 arm, f3 pos, f3 nrm, f3 view_pos, f3 wi, float min_roughness, f3 *d_kd, f3 *d_arm,
                       f3 *d_pos, f3 *d_nrm, f3 *d_view_pos, f3 *d_wi, f3 d_diffuse, f3 d_specular) /* bsdf.h:238 */
{
    f3 _wo = sub3(view_pos, pos);
    f3 wo = safe_normalize_e(_wo);
    float alpha = arm.y * arm.y;
    f3 spec_col = scale3(add3(scale3(mk3(0.04f, 0.04f, 0.04f), 1.0f - arm.z), scale3(kd, arm.z)), VFAST_E ? 1.0f - arm.x : (float)(1.0 - (double)arm.x));
    float d_alpha = 0;
    *d_wi = mk3(0, 0, 0);
    f3 d_spec_col = mk3(0, 0, 0), d_wo = mk3(0, 0, 0);
    bwdPbrSpecular(spec_col, nrm, wo, wi, alpha, min_roughness, &d_spec_col, d_nrm, &d_wo, d_wi, &d_alpha, d_specular);
    float d_diff = sum3(d_diffuse);
    bwdLambert(nrm, wi, d_nrm, d_wi, d_diff);
    *d_kd = sub3(*d_kd, scale3(scale3(d_spec_col, arm.x - 1.0f), arm.z));
    d_arm->x += sum3(mul3(d_spec_col, sub3(scale3(sub3(mk3(0.04f, 0.04f, 0.04f), kd), arm.z), mk3(0.04f, 0.04f, 0.04f))));
    d_arm->z -= sum3(scale3(mul3(d_spec_col, sub3(kd, mk3(0.04f, 0.04f, 0.04f))), arm.x - 1.0f));
    d_arm->y += d_alpha * 2 * arm.y;
    f3 d__wo = mk3(0, 0, 0);
    bwd_safe_normalize(_wo, &d__wo, d_wo);
    *d_view_pos = add3(*d_view_pos, d__wo);
    *d_pos = sub3(*d_pos, d__wo);
}

/* ---------------------------------------------------------------------------------------------
 * sampling helpers (kernel.cu:47-397) */
typedef struct {
    const nvdr_env_shade_args *a;
    int Hl, Wl;          /* probe resolution: pdf.size(0), pdf.size(1) */
    /* scene for brute-force visibility */
    const float *trirec; /* [T,9]: v0, e1, e2 */
    long n_tris;
} shade_env;

static inline f3 tolocal(f3 a, f3 u, f3 v, f3 w) { return mk3(dot3(a, u), dot3(a, v), dot3(a, w)); }
static inline f3 toworld(f3 a, f3 u, f3 v, f3 w) { return add3(add3(scale3(u, a.x), scale3(v, a.y)), scale3(w, a.z)); }

static f3 cosine_sample(f3 N, float u, float v, float *pdf)                                   /* kernel.cu:57 */
{
    N = safe_normalize(N);
    f3 dx, dy;
    branchlessONB(N, &dx, &dy);
    float phi = (float)(2.0 * PI_D * (double)u);
    float costheta = sqrtf(v);
    float sintheta = (float)sqrt(1.0 - (double)v);
    float sp, cp;
    nvdr_sincosf(phi, &sp, &cp);
    float x = cp * sintheta, y = sp * sintheta, z = costheta;
    *pdf = VFAST ? fmaxf(0.000001f, VDIV(costheta, PI_F)) : (float)fmax((double)0.000001f, (double)costheta / PI_D);
    f3 vec = add3(add3(scale3(dx, x), scale3(dy, y)), scale3(N, z));
    return safe_normalize(vec);
}
static float albedo(f3 baseColor, f3 wo, f3 N)                                                /* kernel.cu:81 */
{
    f3 W = safe_normalize(N), U, V;
    branchlessONB(W, &U, &V);
    f3 wo_l = safe_normalize(tolocal(wo, U, V, W));
    const float cosNO = wo_l.z;
    if (!(cosNO > 0)) return 0.0f;
    return luminance(fwdFresnelSchlick3(baseColor, mk3(1.f, 1.f, 1.f), cosNO));
}
static void dir_to_tc(f3 dir, float *u, float *v)                                             /* kernel.cu:124 */
{
    *u = (float)((double)nvdr_atan2f(dir.x, -dir.z) / (2.0 * PI_D) + 0.5);
    *v = (float)((double)nvdr_acosf(clampf(dir.y, -1.0f, 1.0f)) / PI_D);
}
static f3 tc_to_dir(float u, float v)                                                         /* kernel.cu:131 */
{
    float sinphi, cosphi, sintheta, costheta;
    nvdr_sincosf((float)((double)(u * 2.0f - 1.0f) * PI_D), &sinphi, &cosphi);
    nvdr_sincosf((float)((double)v * PI_D), &sintheta, &costheta);
    return mk3(sintheta * sinphi, costheta, -sintheta * cosphi);
}
/* bisection on a CDF of `size` entries with element stride `stride` (kernel.cu:140-169) */
static float sample_cdf(const float *cdf, long stride, int size, float x, unsigned *idx, float *pdf)
{
    x = fminf(x, 0.99999994f);
    unsigned lo = 0, hi = (unsigned)(size - 1);
    unsigned m = (unsigned)((int)ceil(log2((double)(float)hi)) + 1); /* int(ceil(log2((float)_max))) + 1 */
    if (hi == 0) m = 0;                                                /* log2(0): undefined in the reference; size 1 */
    for (unsigned i = 0; i < m; ++i) {
        unsigned mid = (lo + hi) / 2;
        float c = cdf[mid * stride];
        lo = x >= c ? mid : lo;
        hi = x < c ? mid : hi;
    }
    *idx = hi;
    float sample;
    if (hi == 0) {
        *pdf = cdf[0];
        sample = x;
    } else {
        float d0 = cdf[hi * stride], d1 = cdf[(hi - 1) * stride];
        *pdf = d0 - d1;
        sample = x - d1;
    }
    return fminf(sample / *pdf, 0.99999994f);
}
static float lightPDF(const shade_env *e, f3 dir)                                              /* kernel.cu:171 */
{
    float cu, cv;
    dir_to_tc(dir, &cu, &cv);
    int x = clampi((int)(cu * (float)e->Wl), 0, e->Wl - 1);
    int y = clampi((int)(cv * (float)e->Hl), 0, e->Hl - 1);
    float s, c;
    nvdr_sincosf((float)((double)cv * PI_D), &s, &c);
    float pdf_weight = VFAST ? VDIV((float)(e->Hl * e->Wl), (2.0f * PI_F * PI_F) * fmaxf(s, 0.0001f))
                             : (float)((double)(e->Hl * e->Wl) / (2.0 * PI_D * PI_D * (double)fmaxf(s, 0.0001f)));
    const nvdr_tensor *t = &e->a->pdf;
    return ((const float *)t->data)[y * t->stride[0] + x * t->stride[1]] * pdf_weight;
}
static f3 lightSample(const shade_env *e, float u, float v, float *pdf)                        /* kernel.cu:184 */
{
    float row_pdf, col_pdf;
    unsigned x, y;
    const nvdr_tensor *rows = &e->a->rows, *cols = &e->a->cols;
    float ry = sample_cdf((const float *)rows->data, rows->stride[0], (int)rows->size[0], v, &y, &row_pdf);
    float rx = sample_cdf((const float *)cols->data + y * cols->stride[0], cols->stride[1], (int)cols->size[1], u, &x, &col_pdf);
    f3 d = tc_to_dir(((float)x + rx) / (float)e->Wl, ((float)y + ry) / (float)e->Hl);
    *pdf = lightPDF(e, d);
    return d;
}
static void light_texel(const shade_env *e, f3 dir, int *tx, int *ty)                          /* kernel.cu:195-199 */
{
    float cu, cv;
    dir_to_tc(dir, &cu, &cv);
    const nvdr_tensor *L = &e->a->light;
    cu = cu * (float)L->size[1];
    cv = cv * (float)L->size[0];
    *tx = clampi((int)cu, 0, (int)L->size[1] - 1);
    *ty = clampi((int)cv, 0, (int)L->size[0] - 1);
}
static float evalNdfGGX(float alpha, float cosTheta)                                          /* kernel.cu:217 */
{
    float a2 = alpha * alpha;
    float d = ((cosTheta * a2 - cosTheta) * cosTheta + 1);
    return VFAST ? VDIV(a2, (d * d) * PI_F) : (float)((double)a2 / ((double)(d * d) * PI_D));
}
static float evalG1GGX(float alphaSqr, float cosTheta)                                        /* kernel.cu:224 */
{
    if (cosTheta <= 0) return 0;
    float c2 = cosTheta * cosTheta;
    float tan2 = VDIV(fmaxf(1.0f - c2, 0.0f), c2);
    return VDIV(2.0f, 1 + sqrtf(1 + alphaSqr * tan2));
}
static float evalPdfGGX_VNDF(float alpha, f3 wo, f3 h)                                        /* kernel.cu:232 */
{
    float G1 = evalG1GGX(alpha * alpha, wo.z);
    float D = evalNdfGGX(alpha, h.z);
    return VDIV(G1 * D * fmaxf(0.f, dot3(wo, h)), wo.z);
}
static f3 sampleGGX_VNDF(float alpha, f3 wo, float ux, float uy, float *pdf)                  /* kernel.cu:241 */
{
    f3 Vh = safe_normalize(mk3(alpha * wo.x, alpha * wo.y, wo.z));
    f3 T1 = (Vh.z < 0.9999f) ? safe_normalize(cross3(mk3(0.f, 0.f, 1.f), Vh)) : mk3(1.f, 0.f, 0.f);
    f3 T2 = cross3(Vh, T1);
    float r = sqrtf(ux);
    float phi = (2.f * PI_F) * uy;
    float sp, cp;
    nvdr_sincosf(phi, &sp, &cp);
    float t1 = r * cp;
    float t2 = r * sp;
    float s = 0.5f * (1.f + Vh.z);
    t2 = (1.f - s) * sqrtf(1.f - t1 * t1) + s * t2;
    f3 Nh = add3(add3(scale3(T1, t1), scale3(T2, t2)), scale3(Vh, sqrtf(fmaxf(0.f, 1.f - t1 * t1 - t2 * t2))));
    f3 h = safe_normalize(mk3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.f, Nh.z)));
    *pdf = evalPdfGGX_VNDF(alpha, wo, h);
    return h;
}
static f3 ggx_sample(f3 N, f3 wo, float u, float v, float alpha, float *pdf)                  /* kernel.cu:268 */
{
    f3 W = safe_normalize(N), U, V;
    branchlessONB(W, &U, &V);
    f3 wo_l = safe_normalize(tolocal(wo, U, V, W));
    const float cosNO = wo_l.z;
    if (!(cosNO > 0)) {
        *pdf = 0.f;
        return mk3(0.f, 0.f, 0.f);
    }
    f3 h = sampleGGX_VNDF(alpha, wo_l, u, v, pdf);
    float woDotH = dot3(wo_l, h);
    f3 wi_l = sub3(scale3(scale3(h, woDotH), 2.0f), wo_l);
    *pdf = VDIV(*pdf, 4.0f * woDotH);
    f3 wi_o = toworld(wi_l, U, V, W);
    return safe_normalize(wi_o);
}
static float ggx_pdf(f3 N, f3 wo, f3 wi, float alpha)                                         /* kernel.cu:301 */
{
    f3 W = safe_normalize(N), U, V;
    branchlessONB(W, &U, &V);
    f3 wo_l = tolocal(wo, U, V, W);
    f3 wi_l = tolocal(wi, U, V, W);
    float pdf = 0.0f;
    if (wo_l.z > 0 && wi_l.z > 0) {
        f3 m = safe_normalize_v(add3(wi_l, wo_l));
        const float woDotH = dot3(m, wo_l);
        const float D = evalNdfGGX(alpha, m.z);
        float G1 = evalG1GGX(alpha * alpha, wo_l.z);
        pdf = VDIV(G1 * D * fmaxf(0.f, dot3(wo_l, m)), wo_l.z);
        pdf = VDIV(pdf, 4 * woDotH);
    }
    return pdf;
}
static void update_pdf(float *pdf, float opdf, float b)                                       /* kernel.cu:325 */
{
    if (b > 0.000001f) {
        opdf *= b;
        *pdf += opdf;
    }
}
static f3 bsdf_sample(float pDiffuse, float pSpecular, f3 N, f3 wo, f3 s, float alpha, float *pdf) /* kernel.cu:334 */
{
    *pdf = 0.0f;
    f3 wi_o;
    if (s.z < pDiffuse) {
        if (pDiffuse < 0.0001f) {
            *pdf = 1.0f;
            return N;
        }
        wi_o = cosine_sample(N, s.x, s.y, pdf);
        *pdf *= pDiffuse;
        if (pSpecular > 0) {
            float bp = ggx_pdf(N, wo, wi_o, alpha);
            update_pdf(pdf, bp, 1.0f - pDiffuse);
        }
    } else {
        wi_o = ggx_sample(N, wo, s.x, s.y, alpha, pdf);
        *pdf *= 1.f - pDiffuse;
        if (pDiffuse > 0) {
            float bp = VFAST ? VDIV(fmaxf(dot3(N, wi_o), 0.0f), PI_F) : (float)(fmax((double)dot3(N, wi_o), 0.0) / PI_D);
            update_pdf(pdf, bp, pDiffuse);
        }
    }
    return wi_o;
}
static float bsdf_pdf(float pDiffuse, float pSpecular, f3 N, f3 wo, f3 wi, float alpha)        /* kernel.cu:374 */
{
    float NdotL = dot3(N, wi), NdotV = dot3(N, wo);
    float pdf = 0.0f;
    if (fminf(NdotV, NdotL) < 1e-6f) return 1.0f;
    if (pDiffuse > 0) {
        float bp = VFAST ? VDIV(fmaxf(dot3(N, wi), 0.0f), PI_F) : (float)(fmax((double)dot3(N, wi), 0.0) / PI_D);
        update_pdf(&pdf, bp, pDiffuse);
    }
    if (pSpecular > 0) {
        float bp = ggx_pdf(N, wo, wi, alpha);
        update_pdf(&pdf, bp, 1.0f - pDiffuse);
    }
    return pdf;
}

/* ---------------------------------------------------------------------------------------------
 * visibility: brute force over all triangles with the predicate of nvdr_raytri.h
 * (stands in for optixTrace, kernel.cu:101-118; returns 1.0 when UNoccluded) */
static float shadow_test(const shade_env *e, f3 o, f3 d)
{
    const float *t = e->trirec;
    for (long k = 0; k < e->n_tris; ++k, t += 9) {
        float tn, un, vn, det;
        if (nvdr_ray_tri(o.x, o.y, o.z, d.x, d.y, d.z, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], &tn, &un, &vn, &det))
            return 0.0f;
    }
    return 1.0f;
}

void oracle_make_trirec(const float *verts, const int32_t *tris, long n_tris, float *out9)
{
    for (long k = 0; k < n_tris; ++k) {
        const float *a = verts + 3 * tris[3 * k], *b = verts + 3 * tris[3 * k + 1], *c = verts + 3 * tris[3 * k + 2];
        float *o = out9 + 9 * k;
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
        o[3] = b[0] - a[0]; o[4] = b[1] - a[1]; o[5] = b[2] - a[2];
        o[6] = c[0] - a[0]; o[7] = c[1] - a[1]; o[8] = c[2] - a[2];
    }
}

/* out_vis[r] = 1 when ray r hits nothing in (0, 1e16) */
void oracle_visibility(const float *verts, const int32_t *tris, long n_tris, const float *ro, const float *rd, long n_rays,
                       uint8_t *out_vis, int n_threads)
{
    float *rec = (float *)malloc(sizeof(float) * 9 * (size_t)n_tris);
    oracle_make_trirec(verts, tris, n_tris, rec);
    shade_env e;
    memset(&e, 0, sizeof(e));
    e.trirec = rec;
    e.n_tris = n_tris;
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1)
    for (long r = 0; r < n_rays; ++r)
        out_vis[r] = shadow_test(&e, mk3(ro[3 * r], ro[3 * r + 1], ro[3 * r + 2]), mk3(rd[3 * r], rd[3 * r + 1], rd[3 * r + 2])) > 0.5f;
    free(rec);
}

/* closest hit: t (<0 miss), original triangle index, barycentrics (u of v1, v of v2) */
void oracle_closest(const float *verts, const int32_t *tris, long n_tris, const float *ro, const float *rd, long n_rays,
                    float *out_t, int32_t *out_tri, float *out_uv, int n_threads)
{
    float *rec = (float *)malloc(sizeof(float) * 9 * (size_t)n_tris);
    oracle_make_trirec(verts, tris, n_tris, rec);
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1)
    for (long r = 0; r < n_rays; ++r) {
        float best = NVDR_RAY_TMAX, bu = 0, bv = 0;
        int bi = -1;
        const float *t = rec;
        for (long k = 0; k < n_tris; ++k, t += 9) {
            float tn, un, vn, det;
            if (nvdr_ray_tri(ro[3 * r], ro[3 * r + 1], ro[3 * r + 2], rd[3 * r], rd[3 * r + 1], rd[3 * r + 2], t[0], t[1], t[2],
                             t[3], t[4], t[5], t[6], t[7], t[8], &tn, &un, &vn, &det)) {
                float tt = tn / det;
                if (tt < best) { best = tt; bu = un / det; bv = vn / det; bi = (int)k; }
            }
        }
        out_t[r] = bi >= 0 ? best : -1.0f;
        out_tri[r] = bi;
        out_uv[2 * r] = bu;
        out_uv[2 * r + 1] = bv;
    }
    free(rec);
}

/* ---------------------------------------------------------------------------------------------
 * the raygen program (kernel.cu:403-542), one call per pixel */
typedef struct {
    f3 diff, spec;                          /* forward accumulators */
    f3 g_pos, g_nrm, g_kd, g_ks;            /* backward per-pixel gradients */
} pixel_out;

static void atomic_addf(float *p, float v)
{
#pragma omp atomic
    *p += v;
}

static void process_sample(const shade_env *e, int backward, f3 ro, f3 dir, f3 pos, f3 nrm, f3 view_pos, f3 kd, f3 ks,
                           float pdfSum, float weight, f3 dgrad, f3 sgrad, float vis, f3 *diff, f3 *spec, pixel_out *po)
{
    const nvdr_env_shade_args *a = e->a;
    int tx, ty;
    light_texel(e, dir, &tx, &ty);
    const nvdr_tensor *L = &a->light;
    const float *lp = (const float *)L->data + ty * L->stride[0] + tx * L->stride[1];
    f3 light_col = L->size[2] == 1 ? mk3(lp[0], lp[0], lp[0]) : mk3(lp[0], lp[L->stride[2]], lp[2 * L->stride[2]]);
#if ORACLE_FAST_VALUE_MATH >= 3
    float mis_weight = VDIV_FAITHFUL(1.0f, fmaxf(pdfSum, 0.0001f));
#else
    float mis_weight = VFAST ? VDIV(1.0f, fmaxf(pdfSum, 0.0001f)) : (float)(1.0 / (double)fmaxf(pdfSum, 0.0001f));
#endif
    f3 _diff = mk3(0, 0, 0), _spec = mk3(0, 0, 0);
    if (a->bsdf == 1 || a->bsdf == 2) {
        float l = fwdLambert(nrm, dir);
        _diff = mk3(l, l, l);
    } else {
        fwdPbrBSDF(kd, ks, pos, nrm, view_pos, dir, 0.08f, &_diff, &_spec);
    }
    (void)ro;
    float V = vis * a->shadow_scale + (1 - a->shadow_scale);
    if (backward) {
        f3 lg = scale3(scale3(scale3(add3(mul3(dgrad, _diff), mul3(sgrad, _spec)), V), mis_weight), weight);
        float *g = a->light_grad + ((long)ty * L->size[1] + tx) * 3;
        atomic_addf(g + 0, lg.x); atomic_addf(g + 1, lg.y); atomic_addf(g + 2, lg.z);
        f3 _dg = scale3(scale3(scale3(mul3(dgrad, light_col), V), mis_weight), weight);
        f3 _sg = scale3(scale3(scale3(mul3(sgrad, light_col), V), mis_weight), weight);
        f3 g_kd = mk3(0, 0, 0), g_ks = mk3(0, 0, 0), g_pos = mk3(0, 0, 0), g_nrm = mk3(0, 0, 0), g_vp = mk3(0, 0, 0), g_wi = mk3(0, 0, 0);
        if (a->bsdf == 1 || a->bsdf == 2) {
            bwdLambert(nrm, dir, &g_nrm, &g_wi, sum3(_dg));
        } else {
            bwdPbrBSDF(kd, ks, pos, nrm, view_pos, dir, 0.08f, &g_kd, &g_ks, &g_pos, &g_nrm, &g_vp, &g_wi, _dg, _sg);
        }
        po->g_pos = add3(po->g_pos, g_pos);
        po->g_nrm = add3(po->g_nrm, g_nrm);
        po->g_kd = add3(po->g_kd, g_kd);
        po->g_ks = add3(po->g_ks, g_ks);
    }
    *diff = scale3(scale3(scale3(mul3(_diff, light_col), V), mis_weight), weight);
    *spec = scale3(scale3(scale3(mul3(_spec, light_col), V), mis_weight), weight);
}

/* Runs the whole launch grid (W, H, N) (torch_bindings.cpp:182-183,266-267).
 *   verts/tris : the mesh the shadow rays are tested against (brute force)
 *   vis_in     : optional uint8 [N*H*W, 2S] visibility to use INSTEAD of tracing (1 = unoccluded)
 *   vis_out    : optional uint8 [N*H*W, 2S] visibility that was used
 *   dbg        : optional f32 [N*H*W, 2S, 4] (dir.xyz, pdf_light + pdf_bsdf) per sample
 * Outputs are zero-filled first, like torch::zeros in the reference.  Returns the number of covered pixels. */
long oracle_env_shade_frozen(const nvdr_env_shade_args *a, const nvdr_env_shade_args *samp, const float *verts, const int32_t *tris,
                             long n_tris, int backward, int n_threads, const uint8_t *vis_in, uint8_t *vis_out, float *dbg);

long oracle_env_shade(const nvdr_env_shade_args *a, const float *verts, const int32_t *tris, long n_tris, int backward,
                      int n_threads, const uint8_t *vis_in, uint8_t *vis_out, float *dbg)
{
    return oracle_env_shade_frozen(a, a, verts, tris, n_tris, backward, n_threads, vis_in, vis_out, dbg);
}

/* The same program with FROZEN SAMPLES: everything that decides where the samples go and what their pdfs are (normal,
 * view vector, roughness, lobe probabilities -- kernel.cu:490-526) is taken from `samp`'s G-buffer, everything that is
 * EVALUATED at a sample (process_sample, kernel.cu:403-461) from `a`'s.  With samp == a this is the reference program.  The
 * backward pass of the reference differentiates the evaluation only, so a finite-difference check of its gradients
 * must hold the samples fixed while the evaluated inputs move: tests/test_oracle_selfconsistency.py. */
long oracle_env_shade_frozen(const nvdr_env_shade_args *a, const nvdr_env_shade_args *samp, const float *verts, const int32_t *tris,
                             long n_tris, int backward, int n_threads, const uint8_t *vis_in, uint8_t *vis_out, float *dbg)
{
    const long N = a->ro.size[0], H = a->ro.size[1], W = a->ro.size[2];
    shade_env e;
    e.a = a;
    e.Hl = (int)a->pdf.size[0];
    e.Wl = (int)a->pdf.size[1];
    float *rec = NULL;
    if (!vis_in) {
        rec = (float *)malloc(sizeof(float) * 9 * (size_t)n_tris);
        oracle_make_trirec(verts, tris, n_tris, rec);
    }
    e.trirec = rec;
    e.n_tris = n_tris;
    const unsigned n = a->n_samples_x, S = n * n;
    const long npix = N * H * W;
    if (!backward) {
        memset(a->diff, 0, sizeof(float) * 3 * npix);
        memset(a->spec, 0, sizeof(float) * 3 * npix);
    } else {
        memset(a->gb_pos_grad, 0, sizeof(float) * 3 * npix);
        memset(a->gb_normal_grad, 0, sizeof(float) * 3 * npix);
        memset(a->gb_kd_grad, 0, sizeof(float) * 3 * npix);
        memset(a->gb_ks_grad, 0, sizeof(float) * 3 * npix);
        memset(a->light_grad, 0, sizeof(float) * 3 * a->light.size[0] * a->light.size[1]);
    }
    long covered = 0;
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : covered) num_threads(n_threads > 0 ? n_threads : 1)
    for (long lin = 0; lin < npix; ++lin) {
        const long z = lin / (H * W), y = (lin / W) % H, x = lin % W;
        /* mask is [N,H,W]: mask[idx.z][idx.y][idx.x] (kernel.cu:470) */
        const float mask = ((const float *)a->mask.data)[z * a->mask.stride[0] + y * a->mask.stride[1] + x * a->mask.stride[2]];
        f3 ro = fetch3v(&a->ro, z, y, x), pos = fetch3v(&a->gb_pos, z, y, x), nrm = fetch3v(&a->gb_normal, z, y, x);
        f3 view_pos = fetch3v(&a->gb_view_pos, z, y, x), kd = fetch3v(&a->gb_kd, z, y, x), ks = fetch3v(&a->gb_ks, z, y, x);
        if (mask <= 0) continue;
        covered++;
        f3 dgrad = mk3(0, 0, 0), sgrad = mk3(0, 0, 0);
        if (backward) {
            dgrad = fetch3v(&a->diff_grad, z, y, x);
            sgrad = fetch3v(&a->spec_grad, z, y, x);
        }
        pixel_out po;
        memset(&po, 0, sizeof(po));
        const float strata_frac = 1.0f / (float)n;
        const float sample_frac = 1.0f / (float)(n * n);
        /* sampling-side copies of the G-buffer (the same values unless the caller froze the samples) */
        const f3 s_pos = fetch3v(&samp->gb_pos, z, y, x), s_nrm = fetch3v(&samp->gb_normal, z, y, x);
        const f3 s_view_pos = fetch3v(&samp->gb_view_pos, z, y, x), s_kd = fetch3v(&samp->gb_kd, z, y, x), s_ks = fetch3v(&samp->gb_ks, z, y, x);
        const float alpha = s_ks.y * s_ks.y;
        f3 wo = safe_normalize(sub3(s_view_pos, s_pos));
        const float metallic = s_ks.z;
        f3 specColor = add3(scale3(mk3(0.04f, 0.04f, 0.04f), 1.0f - metallic), scale3(s_kd, metallic));
        float diffuseWeight = (1.f - metallic) * luminance(s_kd);
        float specularWeight = albedo(specColor, wo, s_nrm);
        float pDiffuse = (diffuseWeight + specularWeight) > 0.f ? diffuseWeight / (diffuseWeight + specularWeight) : 1.f;
        float pSpecular = 1.0f - pDiffuse;

        uint32_t rng = hash_pcg(a->rnd_seed, (uint32_t)lin + a->pixel_index_offset);
        const uint32_t NP = (uint32_t)a->perms.size[0];
        uint32_t lightIdx = rand_pcg(&rng) % NP;
        uint32_t bsdfIdx = rand_pcg(&rng) % NP;
        const int32_t *perms = (const int32_t *)a->perms.data;
        const long ps0 = a->perms.stride[0], ps1 = a->perms.stride[1];
        f3 diffAccum = mk3(0, 0, 0), specAccum = mk3(0, 0, 0);
        for (unsigned i = 0; i < S; ++i) {
            f3 dir, d, s;
            float sx, sy, sz, pdf_light, pdf_bsdf, vis;
            /* light importance sampling (kernel.cu:513-519) */
            unsigned pl = (unsigned)perms[lightIdx * ps0 + i * ps1];
            sx = ((float)(pl % n) + uniform_pcg(&rng)) * strata_frac;
            sy = ((float)(pl / n) + uniform_pcg(&rng)) * strata_frac;
            dir = lightSample(&e, sx, sy, &pdf_light);
            pdf_bsdf = bsdf_pdf(pDiffuse, pSpecular, s_nrm, wo, dir, alpha);
            vis = vis_in ? (float)vis_in[lin * 2 * S + 2 * i] : shadow_test(&e, ro, dir);
            if (vis_out) vis_out[lin * 2 * S + 2 * i] = vis > 0.5f;
            if (dbg) { float *q = dbg + (lin * 2 * S + 2 * i) * 4; q[0] = dir.x; q[1] = dir.y; q[2] = dir.z; q[3] = pdf_light + pdf_bsdf; }
            process_sample(&e, backward, ro, dir, pos, nrm, view_pos, kd, ks, pdf_light + pdf_bsdf, sample_frac, dgrad, sgrad, vis, &d, &s, &po);
            diffAccum = add3(diffAccum, d);
            specAccum = add3(specAccum, s);
            /* BSDF importance sampling (kernel.cu:522-529) */
            unsigned pb = (unsigned)perms[bsdfIdx * ps0 + i * ps1];
            sx = ((float)(pb % n) + uniform_pcg(&rng)) * strata_frac;
            sy = ((float)(pb / n) + uniform_pcg(&rng)) * strata_frac;
            sz = uniform_pcg(&rng);
            dir = bsdf_sample(pDiffuse, pSpecular, s_nrm, wo, mk3(sx, sy, sz), alpha, &pdf_bsdf);
            pdf_light = lightPDF(&e, dir);
            vis = vis_in ? (float)vis_in[lin * 2 * S + 2 * i + 1] : shadow_test(&e, ro, dir);
            if (vis_out) vis_out[lin * 2 * S + 2 * i + 1] = vis > 0.5f;
            if (dbg) { float *q = dbg + (lin * 2 * S + 2 * i + 1) * 4; q[0] = dir.x; q[1] = dir.y; q[2] = dir.z; q[3] = pdf_light + pdf_bsdf; }
            process_sample(&e, backward, ro, dir, pos, nrm, view_pos, kd, ks, pdf_light + pdf_bsdf, sample_frac, dgrad, sgrad, vis, &d, &s, &po);
            diffAccum = add3(diffAccum, d);
            specAccum = add3(specAccum, s);
        }
        if (!backward) {
            float *o = a->diff + 3 * lin;
            o[0] = diffAccum.x; o[1] = diffAccum.y; o[2] = diffAccum.z;
            o = a->spec + 3 * lin;
            o[0] = specAccum.x; o[1] = specAccum.y; o[2] = specAccum.z;
        } else {
            float *o = a->gb_pos_grad + 3 * lin;
            o[0] = po.g_pos.x; o[1] = po.g_pos.y; o[2] = po.g_pos.z;
            o = a->gb_normal_grad + 3 * lin;
            o[0] = po.g_nrm.x; o[1] = po.g_nrm.y; o[2] = po.g_nrm.z;
            o = a->gb_kd_grad + 3 * lin;
            o[0] = po.g_kd.x; o[1] = po.g_kd.y; o[2] = po.g_kd.z;
            o = a->gb_ks_grad + 3 * lin;
            o[0] = po.g_ks.x; o[1] = po.g_ks.y; o[2] = po.g_ks.z;
        }
    }
    free(rec);
    return covered;
}

/* ---------------------------------------------------------------------------------------------
 * bilateral denoiser (denoising.cu:14-130).  col [N,H,W,3], nrm [N,H,W,3], zdz [N,H,W,2] strided. */
static inline f3 fetch3t(const nvdr_tensor *t, long z, long y, long x) { return fetch3v(t, z, y, x); }
static inline void fetch2t(const nvdr_tensor *t, long z, long y, long x, float *a, float *b)
{
    const float *p = elem(t, z, y, x);
    if (t->size[3] == 1) { *a = p[0]; *b = p[0]; } else { *a = p[0]; *b = p[t->stride[3]]; }
}

void oracle_bilateral_fwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz, float sigma, float *out,
                          int n_threads)
{
    const long N = col->size[0], H = col->size[1], W = col->size[2];
    const float variance = sigma * sigma;
    const int rad = (int)(2 * ceil((double)sigma * 2.5) + 1);
    (void)n_threads;
#pragma omp parallel for collapse(2) schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
    for (long z = 0; z < N; ++z)
        for (long y = 0; y < H; ++y)
            for (long x = 0; x < W; ++x) {
                f3 c_nrm = fetch3t(nrm, z, y, x);
                float cz, cdz;
                fetch2t(zdz, z, y, x, &cz, &cdz);
                float accum_w = 0.0f;
                f3 accum = mk3(0, 0, 0);
                for (int fy = -rad; fy <= rad; ++fy)
                    for (int fx = -rad; fx <= rad; ++fx) {
                        long ty = y + fy, tx = x + fx;
                        if (ty < 0 || tx < 0 || ty >= H || tx >= W) continue;
                        f3 t_col = fetch3t(col, z, ty, tx), t_nrm = fetch3t(nrm, z, ty, tx);
                        float tz, tdz;
                        fetch2t(zdz, z, ty, tx, &tz, &tdz);
                        float dist_sqr = (float)(fx * fx + fy * fy);
                        float dist = sqrtf(dist_sqr);
                        float w_xy = expf(-dist_sqr / (2.0f * variance));
                        float w_normal = powf(fminf(fmaxf(dot3(t_nrm, c_nrm), 0.0001f), 1.0f), 128.0f);
                        float w_depth = expf(-(fabsf(tz - cz) / fmaxf(cdz * dist, 0.0001f)));
                        float w = w_xy * w_normal * w_depth;
                        accum = add3(accum, scale3(t_col, w));
                        accum_w += w;
                    }
                float *o = out + ((z * H + y) * W + x) * 4;
                o[0] = accum.x; o[1] = accum.y; o[2] = accum.z;
                o[3] = fmaxf(accum_w, 0.0001f);
            }
}

void oracle_bilateral_bwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz, float sigma,
                          const nvdr_tensor *out_grad, float *col_grad, int n_threads)
{
    const long N = col->size[0], H = col->size[1], W = col->size[2];
    const float variance = sigma * sigma;
    const int rad = (int)(2 * ceil((double)sigma * 2.5) + 1);
    (void)n_threads;
#pragma omp parallel for collapse(2) schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
    for (long z = 0; z < N; ++z)
        for (long y = 0; y < H; ++y)
            for (long x = 0; x < W; ++x) {
                f3 c_nrm = fetch3t(nrm, z, y, x);
                float cz, cdz;
                fetch2t(zdz, z, y, x, &cz, &cdz);
                f3 accum = mk3(0, 0, 0);
                for (int fy = -rad; fy <= rad; ++fy)
                    for (int fx = -rad; fx <= rad; ++fx) {
                        long ty = y + fy, tx = x + fx;
                        if (ty < 0 || tx < 0 || ty >= H || tx >= W) continue;
                        f3 t_nrm = fetch3t(nrm, z, ty, tx);
                        float tz, tdz;
                        fetch2t(zdz, z, ty, tx, &tz, &tdz);
                        float dist_sqr = (float)(fx * fx + fy * fy);
                        float dist = sqrtf(dist_sqr);
                        float w_xy = expf(-dist_sqr / (2.0f * variance));
                        float w_normal = powf(fminf(fmaxf(dot3(t_nrm, c_nrm), 0.0001f), 1.0f), 128.0f);
                        /* transposed weight: the TAP's dz in the denominator (denoising.cu:118) */
                        float w_depth = expf(-(fabsf(tz - cz) / fmaxf(tdz * dist, 0.0001f)));
                        float w = w_xy * w_normal * w_depth;
                        f3 g = fetch3t(out_grad, z, ty, tx); /* first three channels only (denoising.cu:122) */
                        accum = add3(accum, scale3(g, w));
                    }
                float *o = col_grad + ((z * H + y) * W + x) * 3;
                o[0] = accum.x; o[1] = accum.y; o[2] = accum.z;
            }
}

/* ---------------------------------------------------------------------------------------------
 * EnvironmentLight.update_pdf (render/light.py:46-59), sequential float32 sums.
 * base [Hl,Wl,3] contiguous -> pdf [Hl,Wl], cols [Hl,Wl], rows [Hl] */
void oracle_light_update_pdf(const float *base, long Hl, long Wl, float *pdf, float *cols, float *rows)
{
    double total = 0.0;
    for (long y = 0; y < Hl; ++y) {
        /* pixel_grid y coordinate (render/util.py:62-66): (y + 0.5) / H */
        float Y = ((float)y + 0.5f) / (float)Hl;
        float s = sinf(Y * (float)PI_D);
        for (long x = 0; x < Wl; ++x) {
            const float *b = base + (y * Wl + x) * 3;
            float m = fmaxf(b[0], fmaxf(b[1], b[2]));
            pdf[y * Wl + x] = m * s;
            total += (double)pdf[y * Wl + x];
        }
    }
    float tot = (float)total;
    for (long i = 0; i < Hl * Wl; ++i) pdf[i] = pdf[i] / tot;
    for (long y = 0; y < Hl; ++y) {
        float acc = 0.0f;
        for (long x = 0; x < Wl; ++x) {
            acc += pdf[y * Wl + x];
            cols[y * Wl + x] = acc;
        }
    }
    float acc = 0.0f;
    for (long y = 0; y < Hl; ++y) {
        acc += cols[y * Wl + Wl - 1];
        rows[y] = acc;
    }
    for (long y = 0; y < Hl; ++y) {
        float last = cols[y * Wl + Wl - 1];
        float den = last > 0 ? last : 1.0f;
        for (long x = 0; x < Wl; ++x) cols[y * Wl + x] /= den;
    }
    float lastr = rows[Hl - 1];
    float denr = lastr > 0 ? lastr : 1.0f;
    for (long y = 0; y < Hl; ++y) rows[y] /= denr;
}

/* ---------------------------------------------------------------------------------------------
 * CPU walk of the binary tree the GPU built (nvdr_bvh_export): the canonical any-hit traversal of
 * nvdiffrecmc_amd/csrc/bvh.h (visit_node / box_hit / bvh_any_hit2), restated in plain C so that the node-visit and
 * triangle-test counters of the GPU counting kernels can be checked against something that does not run on the GPU
 * ("n measured by a counting build ... must equal the CPU traversal of the same BVH").  The reference has no
 * counterpart (OptiX's traversal is closed: optixTrace, kernel.cu:104-116).
 *   nodes: uint32 [n_nodes, 8]  six words of 16-bit quantised child boxes + two child references (< 0: ~leaf slot)
 *   trirec: float [n_tris, 12]  (v0, e1, e2, ...) in Morton order
 * out_vis[r] = 1 when ray r hits nothing; counts[0] += node visits, counts[1] += triangle tests. */
typedef struct { float nx, ny, nz, ix, iy, iz; } grid_ray;

static inline int w_box_hit(float minx, float miny, float minz, float maxx, float maxy, float maxz, const grid_ray *r,
                            float tmax, float *tnear)
{
    const float x0 = fmaf(minx, r->ix, r->nx), x1 = fmaf(maxx, r->ix, r->nx);
    const float y0 = fmaf(miny, r->iy, r->ny), y1 = fmaf(maxy, r->iy, r->ny);
    const float z0 = fmaf(minz, r->iz, r->nz), z1 = fmaf(maxz, r->iz, r->nz);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
    *tnear = tn;
    return tn <= tf;
}

void oracle_bvh2_walk(const uint32_t *nodes, const float *trirec, long n_tris, const float *g_lo, const float *g_scale,
                      const float *ro, const float *rd, long n_rays, uint8_t *out_vis, long long *counts, int n_threads)
{
    long long n_node = 0, n_tri = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : n_node, n_tri)
#endif
    for (long r = 0; r < n_rays; ++r) {
        const float ox = ro[3 * r], oy = ro[3 * r + 1], oz = ro[3 * r + 2];
        const float dx = rd[3 * r], dy = rd[3 * r + 1], dz = rd[3 * r + 2];
        int occluded = 0;
        float t, u, v, det;
        if (n_tris == 1) {
            const float *q = trirec;
            n_tri++;
            occluded = nvdr_ray_tri(ox, oy, oz, dx, dy, dz, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], &t, &u, &v, &det);
            out_vis[r] = occluded ? 0 : 1;
            continue;
        }
        grid_ray g;
        /* |inv| capped like make_grid_ray (csrc/bvh.h): an axis the ray does not move along must keep culling */
        g.ix = fminf(fmaxf(1.0f / (dx * g_scale[0]), -1.0e30f), 1.0e30f);
        g.iy = fminf(fmaxf(1.0f / (dy * g_scale[1]), -1.0e30f), 1.0e30f);
        g.iz = fminf(fmaxf(1.0f / (dz * g_scale[2]), -1.0e30f), 1.0e30f);
        g.nx = -((ox - g_lo[0]) * g_scale[0] + 2.0f) * g.ix;
        g.ny = -((oy - g_lo[1]) * g_scale[1] + 2.0f) * g.iy;
        g.nz = -((oz - g_lo[2]) * g_scale[2] + 2.0f) * g.iz;
        int stack[128];
        int sp = 0, cur = 0, done = 0;
        while (!done) {
            if (cur >= 0) {
                const uint32_t *w = nodes + 8 * (long)cur;
                float tl, tr;
                n_node++;
                const int hl = w_box_hit((float)(w[0] & 0xffffu), (float)(w[0] >> 16), (float)(w[1] & 0xffffu), (float)(w[1] >> 16),
                                         (float)(w[2] & 0xffffu), (float)(w[2] >> 16), &g, NVDR_RAY_TMAX, &tl);
                const int hr = w_box_hit((float)(w[3] & 0xffffu), (float)(w[3] >> 16), (float)(w[4] & 0xffffu), (float)(w[4] >> 16),
                                         (float)(w[5] & 0xffffu), (float)(w[5] >> 16), &g, NVDR_RAY_TMAX, &tr);
                const int cl = (int)w[6], cr = (int)w[7];
                if (hl && hr) {
                    const int left_first = tl <= tr;
                    stack[sp++] = left_first ? cr : cl;
                    cur = left_first ? cl : cr;
                } else if (hl) {
                    cur = cl;
                } else if (hr) {
                    cur = cr;
                } else if (sp > 0) {
                    cur = stack[--sp];
                } else {
                    done = 1;
                }
            } else {
                const float *q = trirec + 12 * (long)(~cur);
                n_tri++;
                if (nvdr_ray_tri(ox, oy, oz, dx, dy, dz, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], &t, &u, &v, &det)) {
                    occluded = 1;
                    done = 1;
                } else if (sp > 0) {
                    cur = stack[--sp];
                } else {
                    done = 1;
                }
            }
        }
        out_vis[r] = occluded ? 0 : 1;
    }
    counts[0] += n_node;
    counts[1] += n_tri;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* include/nvdr_detmath.h evaluated on the host: op 0 sin, 1 cos, 2 acos, 3 atan2(x, y) -- same
 * numbering as the device hook nvdr_test_detmath. */
void oracle_detmath(int op, const float *x, const float *y, long n, float *out)
{
    for (long i = 0; i < n; ++i) {
        float s, c;
        switch (op) {
        case 0: nvdr_sincosf(x[i], &s, &c); out[i] = s; break;
        case 1: nvdr_sincosf(x[i], &s, &c); out[i] = c; break;
        case 2: out[i] = nvdr_acosf(x[i]); break;
        default: out[i] = nvdr_atan2f(x[i], y[i]); break;
        }
    }
}
