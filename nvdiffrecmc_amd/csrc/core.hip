// core.hip -- error state, version, roctx ranges, and the detmath device test hook.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"
#include "ieee_arith.h"

static thread_local char g_err[1024] = "";

void nvdr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *nvdr_last_error(void) { return g_err; }

const char *nvdr_tuning_env(const char *name)
{
    const char *v = getenv(name);
    if (!v) return nullptr;
    const char *t = getenv("NVDR_TUNING");
    if (t && atoi(t) > 0) return v;
    fprintf(stderr, "[nvdr] %s is set but ignored: experiment switches need NVDR_TUNING=1\n", name);
    return nullptr;
}

// ---------------------------------------------------------------------------------------------
// roctx ranges around the entry points and their stages (bvh build, gen, trace, shade, light gradient, filter, optimiser): the
// reference has no tracing at all (train.py:416,481-492 time the iteration on the host).  Off unless NVDR_ROCTX=1; the marker
// library is looked up at the first range so that nothing links against the profiler (`rocprofv3 --marker-trace --kernel-trace`
// shows the ranges on the host timeline beside the kernels they enqueue -- they time the ENQUEUE, the launches are asynchronous).
static int (*g_range_push)(const char *) = nullptr;
static int (*g_range_pop)(void) = nullptr;
static int g_range_state = 0;           // 0 not looked up yet, 1 on, -1 off

bool nvdr_range_enabled(void)
{
    if (g_range_state == 0) {
        const char *e = getenv("NVDR_ROCTX");
        int st = -1;
        if (e && atoi(e) > 0) {
            void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                g_range_push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                g_range_pop = (int (*)(void))dlsym(h, "roctxRangePop");
                if (g_range_push && g_range_pop) st = 1;
            }
            if (st < 0) fprintf(stderr, "nvdr: NVDR_ROCTX=1 but no roctx library could be loaded (%s)\n", dlerror());
        }
        g_range_state = st;
    }
    return g_range_state > 0;
}
void nvdr_range_push(const char *name) { if (nvdr_range_enabled()) g_range_push(name); }
void nvdr_range_pop(void) { if (nvdr_range_enabled()) g_range_pop(); }
extern "C" int nvdr_version(void) { return 101; }

extern "C" size_t nvdr_abi_sizeof(int which)
{
    switch (which) {
    case 0: return sizeof(nvdr_adam_tensor);
    case 1: return sizeof(nvdr_env_shade_args);
    case 2: return sizeof(nvdr_texture_args);
    case 3: return sizeof(nvdr_interpolate_bwd_args);
    case 4: return sizeof(nvdr_tensor);
    case 5: return sizeof(nvdr_gbuffer_args);
    case 6: return sizeof(nvdr_mesh_args);
    case 7: return sizeof(nvdr_bvh_info);
    default: return 0;
    }
}

// ---------------------------------------------------------------------------------------------
// ieee_arith.h against the compiler's own expansions, on the device (nvdr_test_arith).  Counters (uint64, zeroed by the entry):
//   [0] nvdr_sqrt != sqrtf among ALL floats that are zero, NaN, infinite, >= 2^-96 or <= -2^-126         (must be 0)
//   [1] nvdr_sqrt != sqrtf among the positive floats below 2^-96 and the negative denormals (outside its documented domain; reported)
//   [2] nvdr_div  != `/`   for pairs inside its domain: random pairs + every mantissa of the denominator   (must be 0)
//   [3] nvdr_div  != `/`   for {+-0, +-inf, NaN, +-1, +-3, +-max, +-2^-100} over {+-0, +-inf, NaN, +-1, +-3}   (must be 0)
//   [4] nvdr_ddiv != `/`   on doubles made of floats: (1 - f1) / f2, f1 * f2 / f3, f1 / (f2 * f2 * pi)      (must be 0)
//   [5] nvdr_dsqrt != sqrt on 1 - v for every float v in [0, 1) and on the random floats promoted to double (must be 0)
//   [6] how many comparisons ran
__device__ __forceinline__ unsigned arith_hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b); }
// a float with a random mantissa and sign and an exponent in [-range, range]
__device__ __forceinline__ float arith_float(unsigned h, int range)
{
    const unsigned e = 127u + (unsigned)((int)(arith_hash(h ^ 0x9e3779b9u) % (unsigned)(2 * range + 1)) - range);
    return __uint_as_float((h & 0x807fffffu) | (e << 23));
}

__global__ void __launch_bounds__(256) arith_selftest_kernel(unsigned long long *out, int part)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad0 = 0, bad1 = 0, n = 0;
    if (part == 0) {                                    // every float through the square root
        for (unsigned long long i = gid; i < (1ull << 32); i += stride) {
            const float x = __uint_as_float((unsigned)i);
            const bool ok = same_bits(nvdr_sqrt(x), sqrtf(x));
            const bool below = (x > 0.0f && x < 1.262177448e-29f) || (x < 0.0f && x > -1.175494351e-38f);      // below 2^-96, or a negative denormal
            if (!ok) { if (below) ++bad1; else ++bad0; }
            ++n;
        }
        atomicAdd(out + 0, bad0); atomicAdd(out + 1, bad1);
    } else if (part == 1) {                             // divisions inside the domain
        for (unsigned long long i = gid; i < (1ull << 30); i += stride) {
            const unsigned h = arith_hash((unsigned)i), g = arith_hash((unsigned)i ^ 0x68bc21ebu);
            const float a = arith_float(h, 30), b = arith_float(g, 60);
            if (!same_bits(nvdr_div(a, b), a / b)) ++bad0;
            // every mantissa of the denominator (i & 2^23 - 1), exponent from the upper bits, against a random numerator
            const float b2 = __uint_as_float(((unsigned)i & 0x7fffffu) | ((127u - 20u + (((unsigned)i >> 23) & 31u)) << 23));
            if (!same_bits(nvdr_div(a, b2), a / b2)) ++bad0;
            if (!same_bits(nvdr_div(1.0f, b2), 1.0f / b2)) ++bad0;
            n += 3;
        }
        atomicAdd(out + 2, bad0);
    } else if (part == 2) {                             // special values: every numerator over the denominators +-0, +-inf, NaN, +-1, +-3
        const float sp[] = {0.0f, -0.0f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1.0f, -1.0f, 3.0f, -3.0f,
                            3.402823466e38f, -3.402823466e38f, 7.888609052e-31f, -7.888609052e-31f};
        const int K = (int)(sizeof(sp) / sizeof(sp[0])), KB = 9;
        if (gid < (unsigned long long)(K * KB)) {
            const float a = sp[gid / KB], b = sp[gid % KB];
            if (!same_bits(nvdr_div(a, b), a / b)) ++bad0;
            ++n;
        }
        atomicAdd(out + 3, bad0);
    } else if (part == 3) {                             // fp64 division on doubles made of floats
        for (unsigned long long i = gid; i < (1ull << 28); i += stride) {
            const unsigned h = arith_hash((unsigned)i);
            const float f1 = arith_float(h, 40), f2 = arith_float(arith_hash(h + 1u), 40), f3 = arith_float(arith_hash(h + 2u), 40);
            const float c2 = __uint_as_float(0x3f800000u - 1u - (h >> 3));            // a float in (2^-64 .. 1): the c2 of fwd_lambda_ggx and beyond
            if (!same_bits(nvdr_ddiv(1.0 - (double)c2, (double)c2), (1.0 - (double)c2) / (double)c2)) ++bad0;
            if (!same_bits(nvdr_ddiv((double)f1 * (double)f2, (double)f3), (double)f1 * (double)f2 / (double)f3)) ++bad0;
            const double den = (double)(f2 * f2) * 3.14159265358979323846;
            if (!same_bits(nvdr_ddiv((double)f1, den), (double)f1 / den)) ++bad0;
            n += 3;
        }
        atomicAdd(out + 4, bad0);
    } else {                                            // fp64 square root
        for (unsigned long long i = gid; i < 0x3f800000ull; i += stride) {      // every float in [0, 1)
            const double x = 1.0 - (double)__uint_as_float((unsigned)i);
            if (!same_bits(nvdr_dsqrt(x), sqrt(x))) ++bad0;
            const double y = (double)arith_float(arith_hash((unsigned)i), 120);
            if (!same_bits(nvdr_dsqrt(y), sqrt(y))) ++bad0;
            n += 2;
        }
        if (gid == 0) {
            const double sp[] = {0.0, -0.0, (double)__builtin_inff(), -1.0, (double)__builtin_nanf("")};
            for (double x : sp) { if (!same_bits(nvdr_dsqrt(x), sqrt(x))) ++bad0; ++n; }
        }
        atomicAdd(out + 5, bad0);
    }
    atomicAdd(out + 6, n);
}

extern "C" int nvdr_test_arith(unsigned long long *counters, void *stream)
{
    NVDR_REQUIRE(counters != nullptr, "nvdr_test_arith: counters is NULL (7 uint64 in device memory)");
    hipStream_t st = (hipStream_t)stream;
    NVDR_HIP_TRY(hipMemsetAsync(counters, 0, 7 * sizeof(unsigned long long), st));
    for (int part = 0; part < 5; ++part) {
        arith_selftest_kernel<<<part == 2 ? 1 : 8192, 256, 0, st>>>(counters, part);
        NVDR_LAUNCH_CHECK();
    }
    return 0;
}

__global__ void detmath_kernel(int op, const float *__restrict__ x, const float *__restrict__ y, int64_t n,
                               float *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c, r;
    switch (op) {
    case 0: nvdr_sincosf(x[i], &s, &c); r = s; break;
    case 1: nvdr_sincosf(x[i], &s, &c); r = c; break;
    case 2: r = nvdr_acosf(x[i]); break;
    default: r = nvdr_atan2f(x[i], y[i]); break;
    }
    out[i] = r;
}

extern "C" int nvdr_test_detmath(int op, const float *x, const float *y, int64_t n, float *out, void *stream)
{
    NVDR_REQUIRE(op >= 0 && op <= 3, "nvdr_test_detmath: bad op %d", op);
    if (n <= 0) return 0;
    detmath_kernel<<<div_up(n, 256), 256, 0, (hipStream_t)stream>>>(op, x, y, n, out);
    NVDR_LAUNCH_CHECK();
    return 0;
}
