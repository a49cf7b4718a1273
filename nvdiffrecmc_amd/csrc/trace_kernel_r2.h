// trace_kernel_r2.h -- the ROUND-2 shadow-ray kernel (four-slot wide nodes, the triangle tested by the lane that reaches the leaf),
// kept as traversal variant 0 beside the round-3 kernel of trace_kernel.h (variant 1, the default) so that the two can be timed
// in ONE process on the same rays (nvdr_ctx_set_trace_variant; tools/ab_inproc.py) and checked against each other bit for bit.
#pragma once

#include "trace_kernel.h"

#ifndef NVDR_LEAF_MIN
#define NVDR_LEAF_MIN 8
#endif
#ifndef NVDR_TRACE_ALIGN
#define NVDR_TRACE_ALIGN 8
#endif
#ifndef NVDR_TRACE_PAD
#define NVDR_TRACE_PAD 10
#endif
#define NVDR_TRAV_EMPTY 0x7ffffff0           // child reference of an unused slot of a wide node

// one slot of a wide node: box (x, y, z words) + child reference (w).  The (lo, hi) pair of every axis is ordered
// (near, far) for this ray by one byte permute with a per-ray selector, which replaces the min/max pair of the
// generic slab test: 3 perm + 6 cvt + 6 fma + 4 instead of 6 cvt + 6 fma + 6 min/max + 4.
__device__ __forceinline__ bool slot_hit(const uint4 &q, const GridRay &r, float tmax, float &tnear)
{
    const unsigned X = __builtin_amdgcn_perm(q.x, q.y, r.px), Y = __builtin_amdgcn_perm(q.x, q.z, r.py),
                   Z = __builtin_amdgcn_perm(q.y, q.z, r.pz);
    const float nx = fmaf(lo16(X), r.ix, r.nx), fx = fmaf(hi16(X), r.ix, r.nx);
    const float ny = fmaf(lo16(Y), r.iy, r.ny), fy = fmaf(hi16(Y), r.iy, r.ny);
    const float nz = fmaf(lo16(Z), r.iz, r.nz), fz = fmaf(hi16(Z), r.iz, r.nz);
    const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    const float tf = fminf(fminf(fx, fy), fminf(fz, tmax));
    tnear = tn;
    return (tn <= tf) & ((int)q.w != NVDR_TRAV_EMPTY);
}

// COUNT: the counting build (box / triangle tests, per-wave clocks)
template <bool COUNT>
__device__ __forceinline__ void env_trace_body_r2(const TraceLaunch &a, int *smem)
{
    const BvhView &bvh = a.bvh;
    const float4 *__restrict__ rays = a.rays;
    const float4 *__restrict__ pix_origin = a.pix_origin;
    const uint32_t *__restrict__ live = a.live;
    uint8_t *__restrict__ vis = a.vis;
    unsigned long long *counters = a.counters;
    const unsigned rays_per_pixel = a.rays_per_pixel;
    const TravStack stack = make_stack(smem, a.spill, bvh.stack_max, bvh.overflow);
    const int lane = threadIdx.x & 63;
    const unsigned total = *a.ray_count;
    const unsigned wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    ChunkDealer dealer;
    dealer.init(a.queues, total, wid, gridDim.x * (blockDim.x >> 6));
    unsigned next = 0, end = 0;                             // wave-uniform list positions of the claimed chunk
    bool more = total > 0;
    unsigned n_box = 0, n_tri = 0, n_ray = 0;
    const bool single = bvh.n_tris == 1;
    const unsigned long long t_begin = COUNT ? wall_clock64() : 0ull;
    const unsigned long long c_begin = COUNT ? (unsigned long long)__builtin_readcyclecounter() : 0ull;

    int ray = -1, cur = 0, sp = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    GridRay g;
    g.nx = g.ny = g.nz = g.ix = g.iy = g.iz = 0.0f;
    g.px = g.py = g.pz = 0u;
    // The three arms of an iteration -- refill, leaf step, node step -- are gated by WAVE-UNIFORM lane counts so that
    // the two expensive rare ones are never issued for a handful of lanes:
    //   refill : when >= NVDR_REFILL_MIN lanes are idle (or nobody can step) and the range still has rays;
    //   leaf   : when >= NVDR_LEAF_MIN lanes are parked on a leaf, or no lane has a node to visit;
    //   node   : whenever some lane has one.
    // Measured (same GPU session, bob 512^2 x 64 spp): ungated 1.36 ms, (16, 8) 1.23-1.31 ms, one-arm-per-iteration
    // (16, 16) 1.30 ms, (32, 16) 1.68 ms.  The loop has ONE back edge (refill falls through into the step): with a
    // `continue` after the refill the compiler kept two copies of the ray state and moved ~27 registers per iteration.
    // The loop is placed at a fixed offset from a 256-byte boundary so that edits elsewhere cannot move it relative to the
    // instruction-cache lines.
    asm volatile(".p2align %0" ::"n"(NVDR_TRACE_ALIGN));
    asm volatile(".rept %0\n s_nop 0\n .endr" ::"n"(NVDR_TRACE_PAD));
    while (true) {
        const unsigned long long idle = __ballot(ray < 0);
        const int n_idle = __popcll(idle);
        if (n_idle >= NVDR_REFILL_MIN && next >= end && more) {
            more = dealer.claim(lane, next, end);
            if (!more) next = end = 0u;
        }
        if (next < end && n_idle >= NVDR_REFILL_MIN) {
            // refill every idle lane from the wave's chunk (no atomics: the cursor is wave-uniform)
            const unsigned take = next + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
            if (ray < 0 && take < end) {
                const unsigned slot = live[take];
                ray = (int)slot;
                if (COUNT) n_ray++;
                const float4 rd = rays[slot];
                const float4 ro = pix_origin[slot / rays_per_pixel];
                ox = ro.x; oy = ro.y; oz = ro.z;
                dx = rd.x; dy = rd.y; dz = rd.z;
                g = make_grid_ray(bvh.info, ox, oy, oz, dx, dy, dz);
                cur = single ? ~0 : 0;
                sp = 0;
            }
            next += (unsigned)n_idle;
        } else if (n_idle == 64) {
            if (!more) break;
        }
        const unsigned long long on_leaf = __ballot(ray >= 0 && cur < 0);
        const int n_leaf = __popcll(on_leaf);
        const int n_node = __popcll(__ballot(ray >= 0 && cur >= 0));
        const int POP = NVDR_TRAV_DONE, HIT = NVDR_TRAV_DONE - 1, WAIT = NVDR_TRAV_DONE - 2;
        const bool leaf_turn = n_leaf >= NVDR_LEAF_MIN || n_node == 0;   // parked leaves are tested in batches
        const bool node_turn = n_node > 0;
        int nxt = WAIT;                                 // next node / leaf, or one of the markers
        if (leaf_turn && ray >= 0 && cur < 0) {
            if (COUNT) n_tri++;
            nxt = tri_any_hit(bvh.tris, ~cur, ox, oy, oz, dx, dy, dz) ? HIT : POP;
        }
        const int popv = stack.peek(sp);                // value a pop would return (unused when sp == 0)
        if (node_turn && ray >= 0 && cur >= 0) {
            // one step = the four grandchildren of `cur` (bvh.h "wide"): test all, continue with ONE hit slot, push the other
            // hits.  Any-hit needs no order at all, and ordering does not pay here: continuing with the FIRST hit slot instead
            // of the nearest one (4 selects + min3 + 3 compares + 3 selects less per step) also visits 3 % FEWER boxes on the
            // benchmark's shadow rays (44.5 vs 46.0 per ray) -- measured -5 % (8 views), -8 % (one view), -9 % (684 k
            // triangles) in interleaved in-process A/B runs (profiles/r02_ab_traversal_variants.md).
            const uint4 *w4 = bvh.wide + 4 * (int64_t)cur;
            const uint4 q0 = w4[0], q1 = w4[1], q2 = w4[2], q3 = w4[3];
            float t0, t1, t2, t3;
            const bool h0 = slot_hit(q0, g, NVDR_RAY_TMAX, t0), h1 = slot_hit(q1, g, NVDR_RAY_TMAX, t1);
            const bool h2 = slot_hit(q2, g, NVDR_RAY_TMAX, t2), h3 = slot_hit(q3, g, NVDR_RAY_TMAX, t3);
            (void)t0; (void)t1; (void)t2; (void)t3;
            const int c0 = (int)q0.w, c1 = (int)q1.w, c2 = (int)q2.w, c3 = (int)q3.w;
            if (COUNT) n_box += (c0 != NVDR_TRAV_EMPTY) + (c1 != NVDR_TRAV_EMPTY) + (c2 != NVDR_TRAV_EMPTY) + (c3 != NVDR_TRAV_EMPTY);
            // continue with the first hit slot; a later hit slot is pushed iff an earlier one was hit (slot 0 is never pushed)
            nxt = h0 ? c0 : h1 ? c1 : h2 ? c2 : h3 ? c3 : POP;
            const bool b01 = h0 | h1, b012 = b01 | h2;
            if (h1 & h0) sp = stack.push(sp, c1);
            if (h2 & b01) sp = stack.push(sp, c2);
            if (h3 & b012) sp = stack.push(sp, c3);
            // Measured and dropped (same A/B runs): preferring an internal node (+6..11 %) or a leaf (+9..10 %) over the first hit
            // slot; unconditional LDS writes at the running depth + one rare spill branch (0.70 vs 0.67 ms); a wave-uniform
            // "nobody leaves the LDS part of the stack" fast path (+-0 %); reading the stack top only in lanes that pop (+-1 %);
            // leaf batches of 12 / 16 instead of 8 (+-1 %); refill thresholds 8 / 24 (+5 % / +-0 %); 6 waves per SIMD (+3..7 %);
            // a 16-entry LDS stack (+-1 %).  Also measured and dropped (session Y): TWO rays per lane, software-pipelined so that the
            // node fetch of one ray is in flight while the other ray's node is tested (112 VGPRs, 4 waves per SIMD, two LDS stacks
            // per lane; bit-exact on the first run) -- +30 % (one view) / +37 % (8 views): eight hardware-interleaved waves hide
            // the fetch better than four waves that interleave two rays in software.
        }
        bool finished = false;
        if (nxt != WAIT) {
            const bool pop = nxt == POP;
            finished = (nxt == HIT) | (pop & (sp == 0));
            sp -= (pop & (sp > 0)) ? 1 : 0;
            cur = pop ? popv : nxt;
        }
        if (finished) {
            vis[ray] = nxt == HIT ? 0 : 1;
            ray = -1;
        }
    }
    if (COUNT) {
        for (int o = 32; o >= 1; o >>= 1) {
            n_box += __shfl_xor(n_box, o);
            n_tri += __shfl_xor(n_tri, o);
            n_ray += __shfl_xor(n_ray, o);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], (unsigned long long)n_box);
            atomicAdd(&counters[1], (unsigned long long)n_tri);
            atomicAdd(&counters[2], (unsigned long long)n_ray);
            // load balance: sum and maximum of the per-wave busy time (100 MHz ticks), wave count
            const unsigned long long dt = wall_clock64() - t_begin;
            atomicAdd(&counters[3], dt);
            atomicMax(&counters[4], dt);
            atomicAdd(&counters[5], 1ull);
            // shader-clock cycles spent (sum over waves; / counters[3] = cycles per 100 MHz tick, i.e. the clock the waves
            // actually ran at) and the set of XCDs that ran waves
            atomicAdd(&counters[6], (unsigned long long)__builtin_readcyclecounter() - c_begin);
            atomicOr(&counters[7], 1ull << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u));
            counters[8 + 2 * wid] = t_begin;                 // per-wave begin / end ticks (wid < 8192); the XCD it ran on in the top byte
            counters[9 + 2 * wid] = (t_begin + dt) | ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) << 56);
        }
    }
}
