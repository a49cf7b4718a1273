// l1_lookup.hip -- what does a divergent 64-byte node fetch cost at the vector L1 of gfx950?
// Not product code: a measurement behind DESIGN.md section 5 ("the L1 lookup rate, not its hit rate, sets the latency of the
// traversal step").  Every pattern moves the SAME 4 KB per wave and "step" (64 nodes of 64 bytes, as a step of the wide walk does for
// 64 rays) out of an array of 64-byte nodes, with independent addresses (an LCG per lane, nothing depends on the loaded data):
//   own4   each lane loads the 4 x 16 B of ITS OWN node with 4 instructions      (the traversal kernel: 64 distinct lines per instruction)
//   quad4  4 instructions, in each the 4 lanes of a quad load the 4 pieces of ONE node  (16 distinct lines per instruction)
//   own1   each lane loads only the first 16 B of its node, 1 instruction        (64 distinct lines, a quarter of the bytes)
//   same4  all 64 lanes load the 4 pieces of the same node, 4 instructions       (1 line per instruction: the broadcast floor)
// Reported: shader cycles per wave and step at W waves per SIMD, and lines looked up per CU and cycle.
// build: hipcc --offload-arch=gfx950 -O3 l1_lookup.hip -o l1_lookup ; run: ./l1_lookup [nodes]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int PATTERN>
__global__ void __launch_bounds__(256) lookup_kernel(const uint4 *__restrict__ nodes, unsigned n_nodes, int iters, unsigned *out,
                                                     unsigned long long *cyc)
{
    const unsigned lane = threadIdx.x & 63u;
    // (quad4: the four lanes of a quad run the same address stream, so they agree on the node without talking to each other)
    unsigned state = (blockIdx.x * 256u + (PATTERN == 1 ? (threadIdx.x & ~3u) : threadIdx.x)) * 2654435761u + 12345u;
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            state = state * 747796405u + 2891336453u;
            unsigned node = __umulhi(state, n_nodes);
            unsigned piece = (unsigned)k;
            if (PATTERN == 1) piece = lane & 3u;     // the quad's lanes split the pieces of their common node
            if (PATTERN == 3) node = (unsigned)__builtin_amdgcn_readfirstlane((int)node);
            if (PATTERN == 2 && k > 0) { q[k] = make_uint4(0, 0, 0, 0); continue; }
            q[k] = nodes[4u * node + piece];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= q[k].x ^ q[k].y ^ q[k].z ^ q[k].w;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256u + threadIdx.x] = acc;
    if (lane == 0) atomicAdd(cyc, t1 - t0);
}

template <int PATTERN>
static void run(const char *name, const uint4 *nodes, unsigned n_nodes, int blocks_per_cu, double lines_per_step)
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 4000;
    unsigned *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, sizeof(unsigned) * cus * blocks_per_cu * 256);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0);
        lookup_kernel<PATTERN><<<cus * blocks_per_cu, 256>>>(nodes, n_nodes, iters, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double waves = (double)cus * blocks_per_cu * 4;
    const double cyc_per_step = (double)h / waves / iters;                     // per wave
    const double steps_per_cu_cycle = blocks_per_cu * 4.0 / cyc_per_step;         // all waves of a CU together
    printf("%-6s %2d waves/SIMD  %8.1f cycles per wave-step  %6.3f lines looked up per CU and cycle  %6.1f B per CU and cycle  (%.2f ms)\n", name,
           blocks_per_cu, cyc_per_step, steps_per_cu_cycle * lines_per_step, steps_per_cu_cycle * (PATTERN == 2 ? 1024.0 : 4096.0), ms);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char **argv)
{
    const unsigned sizes[3] = {argc > 1 ? (unsigned)atoi(argv[1]) : 10687u, 128u, 687000u};   // bob's wide nodes (684 KB), 8 KB, 44 MB
    for (int s = 0; s < 3; ++s) {
        const unsigned n = sizes[s];
        uint4 *nodes;
        hipMalloc(&nodes, (size_t)n * 64);
        hipMemset(nodes, 1, (size_t)n * 64);
        printf("== %u nodes of 64 B (%.1f KB)\n", n, n * 64.0 / 1024.0);
        for (int w = 8; w >= 2; w /= 4) {
            run<0>("own4", nodes, n, w, 256.0);
            run<1>("quad4", nodes, n, w, 64.0);
            run<2>("own1", nodes, n, w, 64.0);
            run<3>("same4", nodes, n, w, 4.0);
        }
        hipFree(nodes);
    }
    return 0;
}
