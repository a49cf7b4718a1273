// valu_rate.hip -- issue rate of the VALU instruction classes the traversal kernel is made of, on gfx950.
// Not product code: a measurement behind the "valu" roofline of bench.py (what is the peak a wave64 VALU stream can reach?).
// Each test issues N independent instructions per iteration from W waves per SIMD and reports shader cycles per
// wave-instruction per SIMD (s_memtime ticks = shader cycles).   build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define UNROLL 8
#define OPS(body) for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { body } }

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float *out, unsigned long long *cyc, int iters)
{
    float x[UNROLL];
    unsigned q[UNROLL];
    const float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f;
    for (int u = 0; u < UNROLL; ++u) { x[u] = threadIdx.x * 0.001f + u; q[u] = threadIdx.x * 2654435761u + u; }
    const unsigned sel = 0x03020504u ^ (threadIdx.x & 1);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (OP == 0) OPS(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(a), "v"(b));)
    if (OP == 1) OPS(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(q[u]) : "v"(q[(u + 1) % UNROLL]), "v"(sel));)
    if (OP == 2) OPS(asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(x[u]) : "v"(q[u]));)
    if (OP == 3) OPS(asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[u]) : "v"(a));)
    if (OP == 4) OPS(asm volatile("v_and_b32 %0, %0, %1" : "+v"(q[u]) : "v"(sel));)
    if (OP == 5) OPS(asm volatile("v_cmp_le_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[u]) : "v"(a), "v"(b) : "vcc");)
    if (OP == 6) OPS(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&x[u & ~1]) : "v"(*(const double *)&x[(u & ~1)]));)
    if (OP == 7) OPS(asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(q[u]));)
    if (OP == 8) OPS(asm volatile("v_rcp_f32 %0, %0" : "+v"(x[u]));)
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int u = 0; u < UNROLL; ++u) s += x[u] + (float)q[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

template <int OP>
static void run(const char *name, int per_instr, int blocks_per_cu)
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 20000;
    float *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, sizeof(float) * cus * blocks_per_cu * 256);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0);
        rate_kernel<OP><<<cus * blocks_per_cu, 256>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double waves = (double)cus * blocks_per_cu * 4;
    const double instr_per_wave = (double)iters * UNROLL * per_instr;
    const double waves_per_simd = blocks_per_cu;            // 4 waves per block, one per SIMD
    const double cyc_per_instr = (double)h / waves / instr_per_wave / waves_per_simd;   // per SIMD: W waves interleave
    printf("%-28s waves/SIMD %d  %.3f cycles per wave-instruction per SIMD  (%.2f ms, %.1f G wave-instr/s, clock ~%.0f MHz)\n", name,
           blocks_per_cu, cyc_per_instr, ms, waves * instr_per_wave / ms / 1e6, (double)h / waves / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 8}) {
        run<0>("v_fma_f32", 1, w);
        run<1>("v_perm_b32", 1, w);
        run<2>("v_cvt_f32_u32", 1, w);
        run<3>("v_max_f32", 1, w);
        run<4>("v_and_b32", 1, w);
        run<5>("v_cmp_le_f32 + v_cndmask", 2, w);
        run<6>("v_pk_fma_f32", 1, w);
        run<7>("v_lshrrev_b32", 1, w);
        run<8>("v_rcp_f32", 1, w);
    }
    return 0;
}
