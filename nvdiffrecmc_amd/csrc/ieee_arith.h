// ieee_arith.h -- correctly rounded fp32 / fp64 division and square root WITHOUT the range-scaling steps of the compiler's expansion.
//
// gfx950 has no division or square-root instruction that rounds correctly; `a / b` and sqrtf(x) compile to a Newton iteration around
// v_rcp_f32 / v_sqrt_f32 that is bracketed by steps which only matter at the ends of the exponent range:
//     a / b      v_div_scale x 2, v_rcp, fma, fma, mul, fma, fma, fma, v_div_fmas, v_div_fixup                    11 VALU
//     sqrtf(x)   compare + multiply + select (pre-scale below 2^-96), v_sqrt, +-1 ulp, 2 fma, 2 compare + select,
//                multiply + select (un-scale), class test + select                                                17 VALU
//     (double) a / b    the same bracket in fp64                                                                   11 VALU (fp64)
// The shading kernels spend a quarter of their vector instructions there (a normalisation alone is a square root and three divisions),
// and the numerical contract fixes the ROUNDING of these operations, not the way it is reached.  The functions below run the same
// iteration on the unscaled operands:
//     nvdr_div   v_rcp, fma, fma, mul, fma, fma, fma, fma, v_div_fixup    9 VALU; the reciprocal's refinement (3 of them) is shared by
//                                                                         the quotients of one denominator (div3: 21 instead of 33)
//     nvdr_sqrt  v_sqrt, +-1 ulp, 2 fma, 2 compare + select               9 VALU
//     nvdr_ddiv  v_rcp_f64, 4 fma, mul, fma, fma, v_div_fixup_f64         9 VALU
//     nvdr_dsqrt v_rsq_f64, 2 mul, 7 fma, class test + select (2)          13 VALU instead of 20
// v_div_scale returns its operand unchanged -- and v_div_fmas is a plain fma -- unless (ISA, V_DIV_SCALE_F32) the denominator is
// denormal or above 2^126, the numerator is below 2^-103 (biased exponent <= 23), the exponents differ by 96 or more, or the quotient is
// denormal: outside those cases the instruction sequence below IS the compiler's, so the result is the correctly rounded quotient bit
// for bit.  v_div_fixup is kept: zeros, infinities and NaNs give exactly what `/` gives.  What differs: a numerator below 2^-103 or a
// denormal quotient may be one ulp off (values of 1e-31 and less); a DENORMAL denominator or a quotient that overflows gives NaN where
// `/` gives a huge number or infinity.  Hence the rule of use: nvdr_div only where the denominator is bounded away from the denormals
// by construction (a clamped cosine, a length nvdr_sqrt returned for a square of 2^-96 or more -- i.e. >= 2^-48; sqrtf of any positive float is above 2^-75, but nvdr_sqrt of one below 2^-96 may be a denormal: safe_normalize (bsdf_device.h) sends those to sqrtf and `/` --, 1 + x with
// x >= 0, an integer, pi) or where the quotient is selected away when it is not (fwd_pbr_specular); the sites whose denominator is an
// unclamped cosine keep `/`.
// nvdr_sqrt equals sqrtf for EVERY float (all 2^32 inputs compared on the device: nvdr_test_arith, tests/test_gpu_arith.py) except
// positive inputs below 2^-96 -- the unscaled residuals lose bits, and v_sqrt_f32 reads a denormal as zero -- and negative denormals
// (-0 instead of NaN); the arguments here are 1 + x, squared lengths of unit-scale vectors and differences of numbers of order one
// (zero, or above 2^-25).
// nvdr_ddiv is used on operands that are floats promoted to double or products of two of them: their exponents lie within +-300, far
// inside the range where v_div_scale_f64 does anything, so it equals `/` for every such input (zeros / infinities through the fix-up).
// nvdr_dsqrt likewise: sqrt(double) pre-scales arguments below 2^-767 only; zero and infinity keep their select.
// NVDR_PLAIN_ARITH=1 compiles them to the plain operators (A/B and the cross-check of the self-test).
#pragma once

#include <hip/hip_runtime.h>

#ifndef NVDR_PLAIN_ARITH
#define NVDR_PLAIN_ARITH 0
#endif

__device__ __forceinline__ float nvdr_div(float a, float b)
{
#if NVDR_PLAIN_ARITH
    return a / b;
#else
    float r = __builtin_amdgcn_rcpf(b);
    const float nb = -b;
    const float e0 = __builtin_fmaf(nb, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = a * r;
    const float e1 = __builtin_fmaf(nb, q, a);
    q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(nb, q, a);
    q = __builtin_fmaf(e2, r, q);
    return __builtin_amdgcn_div_fixupf(q, b, a);
#endif
}

__device__ __forceinline__ float nvdr_sqrt(float x)
{
#if NVDR_PLAIN_ARITH
    return sqrtf(x);
#else
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
    const float e_dn = __builtin_fmaf(-s_dn, s, x), e_up = __builtin_fmaf(-s_up, s, x);
    s = e_dn <= 0.0f ? s_dn : s;
    s = e_up > 0.0f ? s_up : s;
    return s;
#endif
}

__device__ __forceinline__ double nvdr_ddiv(double a, double b)
{
#if NVDR_PLAIN_ARITH
    return a / b;
#else
    double r = __builtin_amdgcn_rcp(b);
    const double nb = -b;
    double e = __builtin_fma(nb, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(nb, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    const double e1 = __builtin_fma(nb, q, a);
    return __builtin_amdgcn_div_fixup(__builtin_fma(e1, r, q), b, a);
#endif
}

__device__ __forceinline__ double nvdr_dsqrt(double x)
{
#if NVDR_PLAIN_ARITH
    return sqrt(x);
#else
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;       // +-0 and +inf are their own roots
#endif
}
