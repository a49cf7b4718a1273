// common.h -- shared host/device helpers for libnvdr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <cstring>
#include <string>

#include "nvdr_hip.h"
#include "nvdr_detmath.h"

// ---------------------------------------------------------------------------------------------
// Error handling.  The reference's CUDA_CHECK/OPTIX_CHECK build a message and drop it
// (render/optixutils/c_src/common.h:37-61); here every failure is recorded and returned.

void nvdr_set_error(const char *fmt, ...);

#define NVDR_HIP_TRY(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            nvdr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (int)_e;                                                                     \
        }                                                                                       \
    } while (0)

#define NVDR_REQUIRE(cond, ...)                                                                 \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            nvdr_set_error(__VA_ARGS__);                                                        \
            return -1;                                                                          \
        }                                                                                       \
    } while (0)

#define NVDR_LAUNCH_CHECK() NVDR_HIP_TRY(hipGetLastError())

// Experiment switches (NVDR_DEBUG, NVDR_PBLOCKS, NVDR_LG_MODE, ...) are read ONLY when NVDR_TUNING=1 is set as well: a stray
// variable in a production environment changes nothing (it is named once on stderr).  core.hip.
const char *nvdr_tuning_env(const char *name);

// roctx range for the lifetime of a scope (core.hip; a no-op unless NVDR_ROCTX=1)
void nvdr_range_push(const char *name);
void nvdr_range_pop(void);
struct NvdrRange {
    explicit NvdrRange(const char *name) { nvdr_range_push(name); }
    ~NvdrRange() { nvdr_range_pop(); }
    NvdrRange(const NvdrRange &) = delete;
    NvdrRange &operator=(const NvdrRange &) = delete;
};

// ---------------------------------------------------------------------------------------------
// Strided / broadcast views (semantics of fetch3 in optixutils/c_src/common.h:13-27 and
// Tensor::nhwcIndex in renderutils/c_src/tensor.h:31): a dim of size 1 is read with index 0.

struct View4 {
    const float *p;
    int64_t s0, s1, s2, s3; // element strides, already zeroed for broadcast dims
    int c;                  // size of the channel (last) dim
};

static inline View4 make_view4(const nvdr_tensor &t)
{
    View4 v;
    v.p = (const float *)t.data;
    v.s0 = t.size[0] == 1 ? 0 : t.stride[0];
    v.s1 = t.size[1] == 1 ? 0 : t.stride[1];
    v.s2 = t.size[2] == 1 ? 0 : t.stride[2];
    v.s3 = t.size[3] == 1 ? 0 : t.stride[3];
    v.c = (int)t.size[3];
    return v;
}

struct F3 {
    float x, y, z;
};

__device__ __forceinline__ F3 f3(float x, float y, float z)
{
    F3 r;
    r.x = x; r.y = y; r.z = z;
    return r;
}
__device__ __forceinline__ F3 f3(float a) { return f3(a, a, a); }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float b) { return f3(a.x * b, a.y * b, a.z * b); }
__device__ __forceinline__ F3 operator*(float b, F3 a) { return f3(a.x * b, a.y * b, a.z * b); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 &operator+=(F3 &a, F3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
__device__ __forceinline__ F3 &operator-=(F3 &a, F3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
__device__ __forceinline__ float sum3(F3 a) { return a.x + a.y + a.z; }

// channel-broadcasting fetch of a 3-vector at (n,h,w); a 1-channel tensor replicates its value
__device__ __forceinline__ F3 fetch3(const View4 &v, int64_t n, int64_t h, int64_t w)
{
    const float *q = v.p + n * v.s0 + h * v.s1 + w * v.s2;
    return f3(q[0], q[v.s3], q[2 * v.s3]);
}
__device__ __forceinline__ float fetch1(const View4 &v, int64_t n, int64_t h, int64_t w)
{
    return v.p[n * v.s0 + h * v.s1 + w * v.s2];
}

static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }
static inline unsigned div_up(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }
