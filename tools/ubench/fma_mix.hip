// fma_mix.hip -- does gfx950 execute v_fma_mix_f32 (f16 operand consumed directly by an f32 fma), and what does the slab arithmetic of
// the oct walk cost with it?  Two ways to turn two plane BYTES of a node into two plane distances t = q * a + b:
//   A (the kernel today)   v_cvt_f32_ubyte0/1 + 2 x v_fma_f32                                  = 4 VALU per byte pair
//   B                      v_perm_b32 -> packed halves (1024 + q0, 1024 + q1) [0x64 in the high byte of an f16 is 1024.0, the
//                          byte in the low mantissa bits adds q exactly] + 2 x v_fma_mix_f32 with b' = b - 1024 a  = 3 VALU
// Not product code: a measurement.   build: hipcc --offload-arch=gfx950 -O3 fma_mix.hip -o fma_mix
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void check_kernel(const unsigned *q, const float *a, const float *b, float *ta, float *tb, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned w = q[i];
    const float A = a[i], B = b[i];
    const float q0 = (float)(w & 0xffu), q1 = (float)((w >> 8) & 0xffu);
    ta[2 * i] = fmaf(q0, A, B);
    ta[2 * i + 1] = fmaf(q1, A, B);
    const unsigned p = __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u);   // bytes: [0x64, w.b1, 0x64, w.b0]
    const float B2 = fmaf(-1024.0f, A, B);
    float t0, t1;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(p), "v"(A), "v"(B2));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(p), "v"(A), "v"(B2));
    tb[2 * i] = t0;
    tb[2 * i + 1] = t1;
}

#define UNROLL 8
template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float *out, unsigned long long *cyc, int iters)
{
    float x[UNROLL], y[UNROLL];
    unsigned q[UNROLL];
    const float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f;
    for (int u = 0; u < UNROLL; ++u) { x[u] = threadIdx.x * 0.001f + u; y[u] = x[u]; q[u] = threadIdx.x * 2654435761u + u; }
    const unsigned sel = 0x04010400u;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) {          // cvt, cvt, fma, fma
                float c0, c1;
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(c0) : "v"(q[u]));
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(c1) : "v"(q[u]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[u]) : "v"(c0), "v"(a));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y[u]) : "v"(c1), "v"(a));
            } else if (OP == 1) {   // perm, fma_mix, fma_mix
                unsigned p;
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p) : "v"(0x64646464u), "v"(q[u]), "v"(sel));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[u]) : "v"(p), "v"(a));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(y[u]) : "v"(p), "v"(a));
            } else if (OP == 2) {   // v_cvt_f32_ubyte0 alone
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(x[u]) : "v"(q[u]));
            } else if (OP == 3) {   // v_fma_mix_f32 alone
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[u]) : "v"(q[u]), "v"(a));
            } else if (OP == 4) {   // v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(a), "v"(b));
            } else if (OP == 5) {   // v_alignbit_b32
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(q[u]) : "v"(x[u]));
            } else if (OP == 6) {   // v_ldexp_f32
                asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[u]) : "v"(q[u] & 1u));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = b;
    for (int u = 0; u < UNROLL; ++u) s += x[u] + y[u] + (float)q[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

template <int OP>
static void run(const char *name, int per_iter, int blocks_per_cu)
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
    const double units = (double)iters * UNROLL;
    printf("%-34s waves/SIMD %d  %6.2f cycles per unit per SIMD (%d VALU per unit)  %.2f ms  %.1f G units/s  clock ~%.0f MHz\n", name, blocks_per_cu,
           (double)h / waves / units / blocks_per_cu, per_iter, ms, waves * units / ms / 1e6, (double)h / waves / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}

int main()
{
    const int n = 1 << 20;
    unsigned *hq = (unsigned *)malloc(4 * n);
    float *ha = (float *)malloc(4 * n), *hb = (float *)malloc(4 * n), *hta = (float *)malloc(8 * n), *htb = (float *)malloc(8 * n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        hq[i] = (unsigned)rand();
        // a = inv * 2^e: wide dynamic range of either sign; b of the size of the products it cancels against (org * inv + noi)
        const float mag = expf(((float)rand() / RAND_MAX) * 30.0f - 12.0f);
        ha[i] = (rand() & 1) ? mag : -mag;
        hb[i] = -ha[i] * (((float)rand() / RAND_MAX) * 400.0f - 100.0f);
    }
    unsigned *q; float *a, *b, *ta, *tb;
    hipMalloc(&q, 4 * n); hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n); hipMalloc(&ta, 8 * n); hipMalloc(&tb, 8 * n);
    hipMemcpy(q, hq, 4 * n, hipMemcpyHostToDevice); hipMemcpy(a, ha, 4 * n, hipMemcpyHostToDevice); hipMemcpy(b, hb, 4 * n, hipMemcpyHostToDevice);
    check_kernel<<<n / 256, 256>>>(q, a, b, ta, tb, n);
    hipMemcpy(hta, ta, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(htb, tb, 8 * n, hipMemcpyDeviceToHost);
    printf("check: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    double worst = 0.0;     // |t_B - t_A| in units of |a| = in quantisation steps of the plane coordinate
    int bad = 0;
    for (int i = 0; i < 2 * n; ++i) {
        const double d = fabs((double)htb[i] - (double)hta[i]) / fabs((double)ha[i / 2]);
        if (d > worst) worst = d;
        if (!(d < 1e-2)) bad++;
    }
    printf("v_fma_mix_f32 path vs cvt + fma: worst difference %.3e quantisation steps (of 255 per node extent), %d of %d beyond 1e-2\n", worst, bad, 2 * n);
    for (int w : {2, 8}) {
        run<0>("A: 2 cvt_ubyte + 2 fma", 4, w);
        run<1>("B: perm + 2 fma_mix", 3, w);
        run<2>("v_cvt_f32_ubyte0", 1, w);
        run<3>("v_fma_mix_f32", 1, w);
        run<4>("v_max3_f32", 1, w);
        run<5>("v_alignbit_b32", 1, w);
        run<6>("v_ldexp_f32", 1, w);
    }
    return 0;
}
