// gather64.hip -- what do the memory-side counters report for RANDOM 64-BYTE GATHERS on gfx950?
// Not product code: the calibration behind `large_mesh.hbm` of bench.py (the x2 on FETCH_SIZE is calibrated for wide
// coalesced streams only, MI355X_MICROARCH.md "HBM").  The traversal kernel's misses are divergent fetches of 64-byte nodes; this
// kernel issues exactly that -- every lane loads the 4 x 16 B of its own pseudo-random 64-byte record -- over an array that fits the
// 256 MB Infinity Cache (42 MB: the large benchmark mesh's tree + triangles) and one that does not (1 GB), in two address patterns:
//   single  one record per lane and step                                   -> known: records x 64 B requested
//   pair    records 2j and 2j + 1 (the two halves of ONE 128-byte line), the second after the first has arrived
//           -> if an L2 miss fills 128 B, the second half hits L2: TCC_MISS ~ records / 2
// Run under rocprofv3 (tools/sessions/r04_s29.sh): one pass with --pmc FETCH_SIZE, one with --pmc TCC_REQ_sum TCC_MISS_sum; the
// kernel names carry the pattern and the array size.  Printed here: records fetched and the time of each launch.
// build: hipcc --offload-arch=gfx950 -O3 gather64.hip -o gather64 ; run: ./gather64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int PAIR, int BIG>
__global__ void __launch_bounds__(256) gather64_kernel(const uint4 *__restrict__ recs, unsigned n_recs, int iters, unsigned *out)
{
    unsigned state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        state = state * 747796405u + 2891336453u;
        unsigned r = __umulhi(state, PAIR ? n_recs / 2u : n_recs) * (PAIR ? 2u : 1u);
        uint4 q0 = recs[4u * r], q1 = recs[4u * r + 1u], q2 = recs[4u * r + 2u], q3 = recs[4u * r + 3u];
        acc ^= q0.x ^ q1.y ^ q2.z ^ q3.w;
        if (PAIR) {
            r += 1u + (acc & 0u);                      // the other half of the same 128-byte line, AFTER the first half has arrived
            q0 = recs[4u * r]; q1 = recs[4u * r + 1u]; q2 = recs[4u * r + 2u]; q3 = recs[4u * r + 3u];
            acc ^= q0.x ^ q1.y ^ q2.z ^ q3.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;             // keeps the loads alive
}

template <int PAIR, int BIG>
static void run(const uint4 *d, unsigned n_recs, unsigned *out, const char *what)
{
    const int blocks = 256 * 8, iters = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    gather64_kernel<PAIR, BIG><<<blocks, 256>>>(d, n_recs, 4, out);      // warm
    hipEventRecord(a);
    gather64_kernel<PAIR, BIG><<<blocks, 256>>>(d, n_recs, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double recs = (double)blocks * 256 * iters * (PAIR ? 2 : 1);
    printf("%-28s array %6.0f MB: %.0f records of 64 B = %.3f GB requested in %.3f ms (%.2f G records/s, %.2f TB/s requested)\n", what,
           n_recs * 64.0 / 1e6, recs, recs * 64 / 1e9, ms, recs / ms / 1e6, recs * 64 / ms / 1e9);
}

int main()
{
    const size_t big = 1ull << 30, small = 42ull << 20;
    uint4 *d;
    unsigned *out;
    hipMalloc(&d, big);
    hipMalloc(&out, 64);
    hipMemset(d, 1, big);
    hipDeviceSynchronize();
    run<0, 0>(d, (unsigned)(small / 64), out, "single, 42 MB (fits MALL)");
    run<1, 0>(d, (unsigned)(small / 64), out, "pair,   42 MB (fits MALL)");
    run<0, 1>(d, (unsigned)(big / 64), out, "single, 1 GB");
    run<1, 1>(d, (unsigned)(big / 64), out, "pair,   1 GB");
    return 0;
}
