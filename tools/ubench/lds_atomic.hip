// lds_atomic.hip -- how fast are LDS atomics on gfx950?  One workgroup of 1024 threads per CU, 96 KB of accumulators, every lane
// issues ITERS atomic adds to pseudo-random (or deliberately colliding) addresses.  Not product code: a measurement behind the
// light-gradient gather (csrc/env_shade.hip).   build: hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>

#define WORDS (24 * 1024)       // 96 KB
template <int OP, int PATTERN>
__global__ void __launch_bounds__(1024) k(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float acc[];
    for (int i = threadIdx.x; i < WORDS; i += 1024) acc[i] = 0.0f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    for (int it = 0; it < iters; ++it) {
        s = s * 747796405u + 2891336453u;
        unsigned a;
        if (PATTERN == 0) a = (s >> 8) % WORDS;                               // random words
        else if (PATTERN == 1) a = ((s >> 8) % (WORDS / 3)) * 3;              // random texels (three consecutive floats, as the gather)
        else if (PATTERN == 2) a = (threadIdx.x + it * 64) % WORDS;           // consecutive: conflict free
        else a = ((threadIdx.x >> 3) * 3 + it * 96) % WORDS;                  // 8 lanes per address
        if (OP == 0) {
            __hip_atomic_fetch_add(acc + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (OP == 1) {
            __hip_atomic_fetch_add((unsigned *)acc + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (OP == 2) {                                                   // three floats of one texel
            a = a < WORDS - 2 ? a : 0;
            __hip_atomic_fetch_add(acc + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(acc + a + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(acc + a + 2, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (OP == 3) {                                                   // plain read-modify-write (not atomic): the LDS access cost alone
            acc[a] += 1.0f;
        } else if (OP == 4) {                                                   // packed: one 64-bit integer atomic
            __hip_atomic_fetch_add((unsigned long long *)acc + (a >> 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (OP == 5) {                                                   // round 6: fp64 (ds_add_f64) -- does it run at the integer rate?
            __hip_atomic_fetch_add((double *)acc + (a >> 1), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (OP == 6) {                                                   // round 6: the product's compare-and-swap loop on the bit pattern
            unsigned *w = (unsigned *)acc + a;
            unsigned o = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (!__hip_atomic_compare_exchange_strong(w, &o, __float_as_uint(__uint_as_float(o) + 1.0f), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {}
        }
    }
    __syncthreads();
    float t = 0.0f;
    for (int i = threadIdx.x; i < WORDS; i += 1024) t += acc[i];
    out[blockIdx.x * 1024 + threadIdx.x] = t;
}

template <int OP, int PATTERN>
static void run(const char *name, int per_iter)
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipFuncSetAttribute((const void *)k<OP, PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, WORDS * 4);
    float *out;
    hipMalloc(&out, sizeof(float) * cus * 1024);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k<OP, PATTERN><<<cus, 1024, WORDS * 4>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double lane_ops = (double)cus * 1024 * iters * per_iter;
    printf("%-52s %7.3f ms  %7.1f G lane-ops/s  = %.2f lane-ops per CU per ns (%s)\n", name, ms, lane_ops / ms / 1e6, lane_ops / ms / 1e6 / cus, hipGetErrorString(hipGetLastError()));
    hipFree(out);
}

int main()
{
    run<0, 0>("ds_add_f32, random words", 1);
    run<0, 1>("ds_add_f32, random texel base", 1);
    run<2, 1>("3 x ds_add_f32 on a random texel (rgb)", 3);
    run<0, 2>("ds_add_f32, consecutive words (no conflicts)", 1);
    run<0, 3>("ds_add_f32, 8 lanes per address", 1);
    run<1, 0>("ds_add_u32, random words", 1);
    run<1, 2>("ds_add_u32, consecutive words", 1);
    run<4, 0>("ds_add_u64, random", 1);
    run<5, 0>("ds_add_f64, random", 1);
    run<5, 2>("ds_add_f64, consecutive", 1);
    run<5, 3>("ds_add_f64, 8 lanes per address", 1);
    run<6, 0>("compare-and-swap loop (f32 add), random words", 1);
    run<6, 3>("compare-and-swap loop (f32 add), 8 lanes per address", 1);
    run<3, 0>("plain read + add + write, random words", 1);
    run<3, 2>("plain read + add + write, consecutive words", 1);
    return 0;
}
