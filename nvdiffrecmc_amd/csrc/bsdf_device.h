// bsdf_device.h -- GGX / Lambert / Frostbite BSDF evaluation, forward and hand-derived backward,
// as gfx950 device functions.  One copy serves both consumers:
//   * the fused env-shade kernel  (the shader's demodulated variant: render/optixutils/c_src/bsdf.h:21-275)
//   * the stand-alone renderutils ops (render/renderutils/c_src/bsdf.cu:17-377)
// The two families share every building block; they differ only in the top-level pbr_bsdf
// (demodulated diffuse + explicit wi  vs.  kd-modulated diffuse + light position).
//
// Arithmetic follows the reference's evaluation order and its float/double promotions (the
// unsuffixed literals in bsdf.h make several sub-expressions fp64); the file is
// built with -ffp-contract=off, so it rounds like the CPU oracle (oracle/nvdr_oracle.c).  Divisions and square roots whose operands are
// bounded by construction go through ieee_arith.h (the same correctly rounded results in 9 instead of 11 / 17 instructions); the bound is
// named at each site.
#pragma once

#include "common.h"
#include "ieee_arith.h"

#define NVDR_SPECULAR_EPSILON 1e-4f
#define NVDR_PI_FLT 3.14159265358979323846f

__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b)
{
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
// (the callers' s: a length that passed `> 0` -- a square root, hence above 2^-75 --, a cosine above NVDR_SPECULAR_EPSILON, or a cosine whose
// quotient is selected away unless it is above it: fwd_pbr_specular)
__device__ __forceinline__ F3 div3(F3 a, float s) { return f3(nvdr_div(a.x, s), nvdr_div(a.y, s), nvdr_div(a.z, s)); }

__device__ __forceinline__ void bwd_dot(F3 a, F3 b, F3 &d_a, F3 &d_b, float d_out)
{
    d_a.x += d_out * b.x; d_a.y += d_out * b.y; d_a.z += d_out * b.z;
    d_b.x += d_out * a.x; d_b.y += d_out * a.y; d_b.z += d_out * a.z;
}
__device__ __forceinline__ void bwd_cross(F3 a, F3 b, F3 &d_a, F3 &d_b, F3 d_out)
{
    d_a.x += d_out.z * b.y - d_out.y * b.z;
    d_a.y += d_out.x * b.z - d_out.z * b.x;
    d_a.z += d_out.y * b.x - d_out.x * b.y;
    d_b.x += d_out.y * a.z - d_out.z * a.y;
    d_b.y += d_out.z * a.x - d_out.x * a.z;
    d_b.z += d_out.x * a.y - d_out.y * a.x;
}
__device__ __forceinline__ F3 safe_normalize(F3 v)
{
    const float l2 = v.x * v.x + v.y * v.y + v.z * v.z;
    // nvdr_sqrt equals sqrtf from 2^-96 upwards (ieee_arith.h); below that -- a vector shorter than 2^-48, e.g. the face normal of a sliver
    // under trained geometry -- v_sqrt_f32 reads a denormal as zero and the +-1 ulp fix-up would hand nvdr_div a denormal denominator (NaN):
    // the rare arm takes the compiler's square root and division, like the reference
    if (__builtin_expect(!(l2 >= 0x1p-96f), 0)) {
        const float l = sqrtf(l2);
        return l > 0.0f ? f3(v.x / l, v.y / l, v.z / l) : f3(0.0f);
    }
    return div3(v, nvdr_sqrt(l2));
}
__device__ __forceinline__ void bwd_safe_normalize(F3 v, F3 &d_v, F3 d_out)
{
    const float l2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float l = l2 >= 0x1p-96f ? nvdr_sqrt(l2) : sqrtf(l2);      // (see safe_normalize)
    if (l > 0.0f) {
        // == (float)(1.0 / (double)(.)) of the reference: a double quotient of floats rounds to the float quotient.  (Plain `/`: l2 * l of a
        // short vector may be denormal.)
        const float fac = 1.0f / (l2 * l);
        d_v.x += (d_out.x * (v.y * v.y + v.z * v.z) - d_out.y * (v.x * v.y) - d_out.z * (v.x * v.z)) * fac;
        d_v.y += (d_out.y * (v.x * v.x + v.z * v.z) - d_out.x * (v.y * v.x) - d_out.z * (v.y * v.z)) * fac;
        d_v.z += (d_out.z * (v.x * v.x + v.y * v.y) - d_out.x * (v.z * v.x) - d_out.y * (v.z * v.y)) * fac;
    }
}
__device__ __forceinline__ float luminance(F3 c) { return dot3(c, f3(0.2126f, 0.7152f, 0.0722f)); }
__device__ __forceinline__ float pow5f(float x) { const float x2 = x * x; return x2 * x2 * x; }

// ---- Lambert -------------------------------------------------------------------------------
__device__ __forceinline__ float fwd_lambert(F3 nrm, F3 wi) { return fmaxf(nvdr_div(dot3(nrm, wi), NVDR_PI_FLT), 0.0f); }
__device__ __forceinline__ void bwd_lambert(F3 nrm, F3 wi, F3 &d_nrm, F3 &d_wi, float d_out)
{
    if (dot3(nrm, wi) > 0.0f) bwd_dot(nrm, wi, d_nrm, d_wi, nvdr_div(d_out, NVDR_PI_FLT));
}

// ---- Fresnel-Schlick (scalar and rgb) --------------------------------------------------------
__device__ __forceinline__ float fwd_fresnel1(float f0, float f90, float cosTheta)
{
    const float c = clampf(cosTheta, NVDR_SPECULAR_EPSILON, 1.0f - NVDR_SPECULAR_EPSILON);
    const float scale = pow5f(1.0f - c);
    return f0 * (1.0f - scale) + f90 * scale;
}
__device__ __forceinline__ void bwd_fresnel1(float f0, float f90, float cosTheta, float &d_f0, float &d_f90,
                                             float &d_cos, float d_out)
{
    const float c = clampf(cosTheta, NVDR_SPECULAR_EPSILON, 1.0f - NVDR_SPECULAR_EPSILON);
    const float scale = pow5f(fmaxf(1.0f - c, 0.0f));
    d_f0 += (float)((double)d_out * (1.0 - (double)scale));
    d_f90 += d_out * scale;
    if (cosTheta >= NVDR_SPECULAR_EPSILON && cosTheta < 1.0f - NVDR_SPECULAR_EPSILON) {
        const float omc = 1.0f - cosTheta;
        d_cos += d_out * (f90 - f0) * -5.0f * ((omc * omc) * (omc * omc));
    }
}
__device__ __forceinline__ F3 fwd_fresnel3(F3 f0, F3 f90, float cosTheta)
{
    const float c = clampf(cosTheta, NVDR_SPECULAR_EPSILON, 1.0f - NVDR_SPECULAR_EPSILON);
    const float scale = pow5f(1.0f - c);
    return f0 * (1.0f - scale) + f90 * scale;
}
__device__ __forceinline__ void bwd_fresnel3(F3 f0, F3 f90, float cosTheta, F3 &d_f0, F3 &d_f90, float &d_cos, F3 d_out)
{
    const float c = clampf(cosTheta, NVDR_SPECULAR_EPSILON, 1.0f - NVDR_SPECULAR_EPSILON);
    const float scale = pow5f(fmaxf(1.0f - c, 0.0f));
    const float oms = (float)(1.0 - (double)scale);
    d_f0 += d_out * oms;
    d_f90 += d_out * scale;
    if (cosTheta >= NVDR_SPECULAR_EPSILON && cosTheta < 1.0f - NVDR_SPECULAR_EPSILON) {
        const float omc = 1.0f - cosTheta;
        const float p4 = (omc * omc) * (omc * omc);
        d_cos += sum3(((d_out * (f90 - f0)) * -5.0f) * p4);
    }
}

// ---- GGX normal distribution -------------------------------------------------------------------
__device__ __forceinline__ float fwd_ndf_ggx(float alphaSqr, float cosTheta)
{
    const float c = clampf(cosTheta, NVDR_SPECULAR_EPSILON, 1.0f - NVDR_SPECULAR_EPSILON);
    const float d = (c * alphaSqr - c) * c + 1.0f;
    return nvdr_div(alphaSqr, d * d * NVDR_PI_FLT);       // d = 1 - c^2 (1 - alphaSqr) >= 1 - (1 - 1e-4)^2 = 2e-4
}
__device__ __forceinline__ void bwd_ndf_ggx(float alphaSqr, float cosTheta, float &d_alphaSqr, float &d_cos, float d_out)
{
    const float c = clampf(cosTheta, NVDR_SPECULAR_EPSILON, 1.0f - NVDR_SPECULAR_EPSILON);
    const float c2 = c * c;
    const float base = (float)(((double)alphaSqr - 1.0) * (double)c2 + 1.0);
    const float cube = base * base * base;
    d_alphaSqr += nvdr_div(d_out * (1.0f - (alphaSqr + 1.0f) * c2), NVDR_PI_FLT * cube);     // base >= 2e-4 as d above: cube >= 8e-12
    if (cosTheta > NVDR_SPECULAR_EPSILON && cosTheta < 1.0f - NVDR_SPECULAR_EPSILON)
        d_cos += nvdr_div(d_out * -(4.0f * (alphaSqr - 1.0f) * alphaSqr * cosTheta), NVDR_PI_FLT * cube);
}

// ---- Smith Lambda / correlated masking -----------------------------------------------------------
__device__ __forceinline__ float fwd_lambda_ggx(float alphaSqr, float cosTheta)
{
    const float c = clampf(cosTheta, NVDR_SPECULAR_EPSILON, 1.0f - NVDR_SPECULAR_EPSILON);
    const float c2 = c * c;
    const float tan2 = (float)nvdr_ddiv(1.0 - (double)c2, (double)c2);        // c2 >= 1e-8
    return 0.5f * (nvdr_sqrt(1.0f + alphaSqr * tan2) - 1.0f);
}
__device__ __forceinline__ void bwd_lambda_ggx(float alphaSqr, float cosTheta, float &d_alphaSqr, float &d_cos, float d_out)
{
    const float c = clampf(cosTheta, NVDR_SPECULAR_EPSILON, 1.0f - NVDR_SPECULAR_EPSILON);
    const float c2 = c * c;
    const float tan2 = (float)nvdr_ddiv(1.0 - (double)c2, (double)c2);
    d_alphaSqr += (float)nvdr_ddiv((double)d_out * (0.25 * (double)tan2), (double)nvdr_sqrt(alphaSqr * tan2 + 1.0f));
    if (cosTheta > NVDR_SPECULAR_EPSILON && cosTheta < 1.0f - NVDR_SPECULAR_EPSILON)
        d_cos += (float)nvdr_ddiv((double)d_out * -(0.5 * (double)alphaSqr),
                                  (double)((c * c * c) * nvdr_sqrt(nvdr_div(alphaSqr, c2) - alphaSqr + 1.0f)));     // c^3 >= 1e-12, the root >= 1
}
__device__ __forceinline__ float fwd_masking_smith(float alphaSqr, float cosI, float cosO)
{
    return nvdr_div(1.0f, 1.0f + fwd_lambda_ggx(alphaSqr, cosI) + fwd_lambda_ggx(alphaSqr, cosO));     // 1 <= the sum <= 1e4
}
__device__ __forceinline__ void bwd_masking_smith(float alphaSqr, float cosI, float cosO, float &d_alphaSqr,
                                                  float &d_cosI, float &d_cosO, float d_out)
{
    const float s = 1.0f + fwd_lambda_ggx(alphaSqr, cosI) + fwd_lambda_ggx(alphaSqr, cosO);
    const float d_l = nvdr_div(-d_out, s * s);
    bwd_lambda_ggx(alphaSqr, cosI, d_alphaSqr, d_cosI, d_l);
    bwd_lambda_ggx(alphaSqr, cosO, d_alphaSqr, d_cosO, d_l);
}

// ---- GGX specular lobe -----------------------------------------------------------------------
__device__ __forceinline__ F3 fwd_pbr_specular(F3 col, F3 nrm, F3 wo, F3 wi, float alpha, float min_roughness)
{
    const float _alpha = clampf(alpha, min_roughness * min_roughness, 1.0f);
    const float alphaSqr = _alpha * _alpha;
    const F3 h = safe_normalize(wo + wi);
    const float woDotN = dot3(wo, nrm), wiDotN = dot3(wi, nrm), woDotH = dot3(wo, h), nDotH = dot3(nrm, h);
    const float D = fwd_ndf_ggx(alphaSqr, nDotH);
    const float G = fwd_masking_smith(alphaSqr, woDotN, wiDotN);
    const F3 F = fwd_fresnel3(col, f3(1.0f), woDotH);
    const F3 w = div3(((F * D) * G) * 0.25f, woDotN);
    const bool front = (woDotN > NVDR_SPECULAR_EPSILON) & (wiDotN > NVDR_SPECULAR_EPSILON);
    return front ? w : f3(0.0f);
}
__device__ __forceinline__ void bwd_pbr_specular(F3 col, F3 nrm, F3 wo, F3 wi, float alpha, float min_roughness,
                                                 F3 &d_col, F3 &d_nrm, F3 &d_wo, F3 &d_wi, float &d_alpha, F3 d_out)
{
    const float _alpha = clampf(alpha, min_roughness * min_roughness, 1.0f);
    const float alphaSqr = _alpha * _alpha;
    const F3 h = safe_normalize(wo + wi);
    const float woDotN = dot3(wo, nrm), wiDotN = dot3(wi, nrm), woDotH = dot3(wo, h), nDotH = dot3(nrm, h);
    const bool front = (woDotN > NVDR_SPECULAR_EPSILON) & (wiDotN > NVDR_SPECULAR_EPSILON);
    if (!front) return;
    const float D = fwd_ndf_ggx(alphaSqr, nDotH);
    const float G = fwd_masking_smith(alphaSqr, woDotN, wiDotN);
    const F3 F = fwd_fresnel3(col, f3(1.0f), woDotH);
    const F3 d_F = div3(((d_out * D) * G) * 0.25f, woDotN);
    const float d_D = sum3(div3(((d_out * F) * G) * 0.25f, woDotN));
    const float d_G = sum3(div3(((d_out * F) * D) * 0.25f, woDotN));
    float d_woDotN = -sum3(div3((((d_out * F) * D) * G) * 0.25f, woDotN * woDotN));
    F3 d_f90 = f3(0.0f);
    float d_woDotH = 0.0f, d_wiDotN = 0.0f, d_nDotH = 0.0f, d_alphaSqr = 0.0f;
    bwd_fresnel3(col, f3(1.0f), woDotH, d_col, d_f90, d_woDotH, d_F);
    bwd_masking_smith(alphaSqr, woDotN, wiDotN, d_alphaSqr, d_woDotN, d_wiDotN, d_G);
    bwd_ndf_ggx(alphaSqr, nDotH, d_alphaSqr, d_nDotH, d_D);
    F3 d_h = f3(0.0f);
    bwd_dot(nrm, h, d_nrm, d_h, d_nDotH);
    bwd_dot(wo, h, d_wo, d_h, d_woDotH);
    bwd_dot(wi, nrm, d_wi, d_nrm, d_wiDotN);
    bwd_dot(wo, nrm, d_wo, d_nrm, d_woDotN);
    F3 d_h_un = f3(0.0f);
    bwd_safe_normalize(wo + wi, d_h_un, d_h);
    d_wo += d_h_un;
    d_wi += d_h_un;
    if (alpha > min_roughness * min_roughness) d_alpha += d_alphaSqr * 2 * alpha;
}

// ---- the shader's PBR BSDF: demodulated grey diffuse + specular, explicit wi (optixutils bsdf.h:222-275)
__device__ __forceinline__ void fwd_pbr_bsdf_shader(F3 kd, F3 arm, F3 pos, F3 nrm, F3 view_pos, F3 wi,
                                                    float min_roughness, F3 &diffuse, F3 &specular)
{
    const F3 wo = safe_normalize(view_pos - pos);
    const float alpha = arm.y * arm.y;
    const F3 spec_col = (f3(0.04f) * (1.0f - arm.z) + kd * arm.z) * (float)(1.0 - (double)arm.x);
    diffuse = f3(fwd_lambert(nrm, wi));
    specular = fwd_pbr_specular(spec_col, nrm, wo, wi, alpha, min_roughness);
}
__device__ __forceinline__ void bwd_pbr_bsdf_shader(F3 kd, F3 arm, F3 pos, F3 nrm, F3 view_pos, F3 wi,
                                                    float min_roughness, F3 &d_kd, F3 &d_arm, F3 &d_pos, F3 &d_nrm,
                                                    F3 d_diffuse, F3 d_specular)
{
    const F3 _wo = view_pos - pos;
    const F3 wo = safe_normalize(_wo);
    const float alpha = arm.y * arm.y;
    const F3 spec_col = (f3(0.04f) * (1.0f - arm.z) + kd * arm.z) * (float)(1.0 - (double)arm.x);
    float d_alpha = 0.0f;
    F3 d_wi = f3(0.0f), d_spec_col = f3(0.0f), d_wo = f3(0.0f);
    bwd_pbr_specular(spec_col, nrm, wo, wi, alpha, min_roughness, d_spec_col, d_nrm, d_wo, d_wi, d_alpha, d_specular);
    bwd_lambert(nrm, wi, d_nrm, d_wi, sum3(d_diffuse));
    d_kd -= (d_spec_col * (arm.x - 1.0f)) * arm.z;
    d_arm.x += sum3(d_spec_col * ((f3(0.04f) - kd) * arm.z - f3(0.04f)));
    d_arm.z -= sum3((d_spec_col * (kd - f3(0.04f))) * (arm.x - 1.0f));
    d_arm.y += d_alpha * 2 * arm.y;
    F3 d__wo = f3(0.0f);
    bwd_safe_normalize(_wo, d__wo, d_wo);
    d_pos -= d__wo; // the view-position and wi gradients are computed and dropped by the reference (kernel.cu:429,439)
}

// ---- Frostbite diffuse (renderutils bsdf.cu:74-150) ---------------------------------------------
__device__ __forceinline__ float fwd_frostbite(F3 nrm, F3 wi, F3 wo, float linearRoughness)
{
    const float wiDotN = dot3(wi, nrm), woDotN = dot3(wo, nrm);
    if (wiDotN > 0.0f && woDotN > 0.0f) {
        const F3 h = safe_normalize(wo + wi);
        const float wiDotH = dot3(wi, h);
        const float energyBias = 0.5f * linearRoughness;
        const float energyFactor = 1.0f - (0.51f / 1.51f) * linearRoughness;
        const float f90 = energyBias + 2.f * wiDotH * wiDotH * linearRoughness;
        const float wiScatter = fwd_fresnel1(1.f, f90, wiDotN);
        const float woScatter = fwd_fresnel1(1.f, f90, woDotN);
        return wiScatter * woScatter * energyFactor;
    }
    return 0.0f;
}
__device__ __forceinline__ void bwd_frostbite(F3 nrm, F3 wi, F3 wo, float linearRoughness, F3 &d_nrm, F3 &d_wi, F3 &d_wo,
                                              float &d_linearRoughness, float d_out)
{
    const float wiDotN = dot3(wi, nrm), woDotN = dot3(wo, nrm);
    if (wiDotN > 0.0f && woDotN > 0.0f) {
        const F3 h = safe_normalize(wo + wi);
        const float wiDotH = dot3(wi, h);
        const float energyBias = 0.5f * linearRoughness;
        const float energyFactor = 1.0f - (0.51f / 1.51f) * linearRoughness;
        const float f90 = energyBias + 2.f * wiDotH * wiDotH * linearRoughness;
        const float f0 = 1.f;
        const float wiScatter = fwd_fresnel1(f0, f90, wiDotN);
        const float woScatter = fwd_fresnel1(f0, f90, woDotN);
        const float d_wiScatter = d_out * woScatter * energyFactor;
        const float d_woScatter = d_out * wiScatter * energyFactor;
        const float d_energyFactor = d_out * wiScatter * woScatter;
        float d_woDotN = 0.0f, d_f0 = 0.0f, d_f90 = 0.0f;
        bwd_fresnel1(f0, f90, woDotN, d_f0, d_f90, d_woDotN, d_woScatter);
        float d_wiDotN = 0.0f;
        bwd_fresnel1(f0, f90, wiDotN, d_f0, d_f90, d_wiDotN, d_wiScatter);
        const float d_energyBias = d_f90;
        const float d_wiDotH = d_f90 * 4 * wiDotH * linearRoughness;
        d_linearRoughness += d_f90 * 2 * wiDotH * wiDotH;
        d_linearRoughness -= (0.51f / 1.51f) * d_energyFactor;
        d_linearRoughness += (float)(0.5 * (double)d_energyBias);
        F3 d_h = f3(0.0f);
        bwd_dot(wi, h, d_wi, d_h, d_wiDotH);
        F3 d_wo_wi = f3(0.0f);
        bwd_safe_normalize(wo + wi, d_wo_wi, d_h);
        d_wi += d_wo_wi;
        d_wo += d_wo_wi;
        bwd_dot(wo, nrm, d_wo, d_nrm, d_woDotN);
        bwd_dot(wi, nrm, d_wi, d_nrm, d_wiDotN);
    }
}

// ---- renderutils pbr_bsdf: kd-modulated diffuse (lambert / frostbite) + specular, light POSITION (bsdf.cu:300-377)
__device__ __forceinline__ F3 fwd_pbr_bsdf_ru(F3 kd, F3 arm, F3 pos, F3 nrm, F3 view_pos, F3 light_pos,
                                              float min_roughness, int BSDF)
{
    const F3 wo = safe_normalize(view_pos - pos);
    const F3 wi = safe_normalize(light_pos - pos);
    const float alpha = arm.y * arm.y;
    const F3 spec_col = (f3(0.04f * (1.0f - arm.z)) + kd * arm.z) * (float)(1.0 - (double)arm.x);
    const F3 diff_col = kd * (1.0f - arm.z);
    const float diff = BSDF == 0 ? fwd_lambert(nrm, wi) : fwd_frostbite(nrm, wi, wo, arm.y);
    return diff_col * diff + fwd_pbr_specular(spec_col, nrm, wo, wi, alpha, min_roughness);
}
__device__ __forceinline__ void bwd_pbr_bsdf_ru(F3 kd, F3 arm, F3 pos, F3 nrm, F3 view_pos, F3 light_pos,
                                                float min_roughness, int BSDF, F3 &d_kd, F3 &d_arm, F3 &d_pos, F3 &d_nrm,
                                                F3 &d_view_pos, F3 &d_light_pos, F3 d_out)
{
    const F3 _wi = light_pos - pos, _wo = view_pos - pos;
    const F3 wi = safe_normalize(_wi), wo = safe_normalize(_wo);
    const float alpha = arm.y * arm.y;
    const F3 spec_col = (f3(0.04f * (1.0f - arm.z)) + kd * arm.z) * (float)(1.0 - (double)arm.x);
    const F3 diff_col = kd * (1.0f - arm.z);
    const float diff = BSDF == 0 ? fwd_lambert(nrm, wi) : fwd_frostbite(nrm, wi, wo, arm.y);
    float d_alpha = 0.0f;
    F3 d_spec_col = f3(0.0f), d_wi = f3(0.0f), d_wo = f3(0.0f);
    bwd_pbr_specular(spec_col, nrm, wo, wi, alpha, min_roughness, d_spec_col, d_nrm, d_wo, d_wi, d_alpha, d_out);
    const float d_diff = sum3(diff_col * d_out);
    if (BSDF == 0)
        bwd_lambert(nrm, wi, d_nrm, d_wi, d_diff);
    else
        bwd_frostbite(nrm, wi, wo, arm.y, d_nrm, d_wi, d_wo, d_arm.y, d_diff);
    const F3 d_diff_col = d_out * diff;
    d_kd += d_diff_col * (1.0f - arm.z);
    d_arm.z -= sum3(d_diff_col * kd);
    d_kd -= (d_spec_col * (arm.x - 1.0f)) * arm.z;
    d_arm.x += sum3(d_spec_col * ((f3(0.04f) - kd) * arm.z - f3(0.04f)));
    d_arm.z -= sum3((d_spec_col * (kd - f3(0.04f))) * (arm.x - 1.0f));
    d_arm.y += d_alpha * 2 * arm.y;
    F3 d__wi = f3(0.0f);
    bwd_safe_normalize(_wi, d__wi, d_wi);
    d_light_pos += d__wi;
    d_pos -= d__wi;
    F3 d__wo = f3(0.0f);
    bwd_safe_normalize(_wo, d__wo, d_wo);
    d_view_pos += d__wo;
    d_pos -= d__wo;
}
