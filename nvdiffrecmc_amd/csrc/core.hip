// core.hip -- error state, version, roctx ranges, and the detmath device test hook.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

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
