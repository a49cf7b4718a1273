// cuda_shim.h -- just enough CUDA/OptiX device vocabulary to compile the REFERENCE's device
// sources (render/optixutils/c_src/envsampling/kernel.cu, bsdf.h, math_utils.h, denoising.cu)
// with g++ for the CPU, unmodified and in place.  Test infrastructure only (oracle/_ref).
//
// What is emulated and how:
//   * __device__/__global__/__constant__ vanish; float2/3/4, uint3, make_* are plain structs;
//   * CUDA's mixed-precision min/max overloads (float,double)->double are provided because the
//     shader relies on them (kernel.cu:74,366,387);
//   * sincos(float-or-double, float*, float*) resolves to the float overload as in CUDA;
//   * the launch index and the trace payload are thread_local so the driver can run pixels under OpenMP;
//   * optixTrace is a brute-force any-hit loop with the predicate of include/nvdr_raytri.h
//     (OptiX itself is closed source -- this is the one part of the reference that cannot be compiled);
//   * atomicAdd(float*) is an OpenMP atomic.
#pragma once
#define __CUDACC__ 1
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sstream>
#include <stdexcept>
#include <string>

// The shader is compiled by NVRTC (optix_wrapper.cpp:74-126) where no <math.h> is in scope, so M_PI
// is NOT predefined and bsdf.h:14-16 supplies its FLOAT definition; keep it that way here.
#undef M_PI

#define __device__
#define __host__
#define __global__
#define __constant__
#define __inline__ inline
#define __forceinline__ inline
#define __restrict__

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint3 { unsigned int x, y, z; };
struct dim3 { unsigned int x = 1, y = 1, z = 1; };
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { uint3 r = {x, y, z}; return r; }

// CUDA math overload set (crt/math_functions.hpp)
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline double min(float a, double b) { return fmin((double)a, b); }
static inline double max(float a, double b) { return fmax((double)a, b); }
static inline double min(double a, float b) { return fmin(a, (double)b); }
static inline double max(double a, float b) { return fmax(a, (double)b); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float max(float a, int b) { return fmaxf(a, (float)b); }
static inline float max(int a, float b) { return fmaxf((float)a, b); }
#ifdef NVDR_REF_DETMATH
// Variant build: route the shader's transcendental calls through include/nvdr_detmath.h (the
// functions our oracle and the HIP kernels use) so that every DISCRETE decision of the reference
// code (texel, lobe, visibility) can be compared bit for bit with oracle/nvdr_oracle.c.
#include "nvdr_detmath.h"
static inline float ref_dm_sin(float x) { float s, c; nvdr_sincosf(x, &s, &c); return s; }
static inline float ref_dm_cos(float x) { float s, c; nvdr_sincosf(x, &s, &c); return c; }
static inline void sincos(float x, float *s, float *c) { nvdr_sincosf(x, s, c); }
static inline void sincos(double x, float *s, float *c) { nvdr_sincosf((float)x, s, c); }
#define sinf(x) ref_dm_sin(x)
#define cosf(x) ref_dm_cos(x)
#define sin(x) ref_dm_sin(x)
#define cos(x) ref_dm_cos(x)
#define acosf(x) nvdr_acosf(x)
#define atan2f(y, x) nvdr_atan2f(y, x)
#else
static inline void sincos(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
static inline void sincos(double x, float *s, float *c) { sincos((float)x, s, c); }
#endif

static inline float atomicAdd(float *p, float v)
{
    float old;
#pragma omp atomic capture
    { old = *p; *p += v; }
    return old;
}

// launch geometry (thread-local: one "thread" of the launch at a time per host thread)
extern thread_local uint3 ref_launch_index;
extern uint3 ref_launch_dim;
extern thread_local uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
