// core.hip -- error state, version, and the detmath device test hook.
#include <stdarg.h>

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
extern "C" int nvdr_version(void) { return 100; }

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
