// trace_kernel.h -- stage 2 of env-shade: persistent-wavefront any-hit traversal of the ray stream.
//
// Replaces optixTrace inside __raygen__rg (render/optixutils/c_src/envsampling/kernel.cu:101-118).
#pragma once

#include "bvh.h"

#ifndef NVDR_REFILL_MIN
#define NVDR_REFILL_MIN 16
#endif
#ifndef NVDR_LEAF_MIN
#define NVDR_LEAF_MIN 8
#endif
// Chunks of NVDR_TRACE_QCHUNK rays are CLAIMED from device counters, one 128-B line each, instead of being dealt
// round-robin: all waves then work inside a moving window of the list (neighbouring pixels -> the same subtrees stay in
// L2) and a wave that drew cheap rays simply claims more.  One counter serialises at ~70 ns per claim (round 1: 1.4-2.5 ms
// with a single counter); 64 of them see < 1 M claims/s each.
// Measured (round 2, same GPU session, 8-view launch, 43 M live rays): static round-robin chunks 4.30 ms, 64 queues x 256
// rays 3.75 ms (-13 %), 128 x 256 4.06 ms, 64 x 128 4.07 ms.
#define NVDR_TRACE_QUEUES 64
#ifndef NVDR_TRACE_QCHUNK
#define NVDR_TRACE_QCHUNK 256
#endif
#ifndef NVDR_TRACE_ALIGN
#define NVDR_TRACE_ALIGN 8
#endif
#ifndef NVDR_TRACE_PAD
#define NVDR_TRACE_PAD 10
#endif
#ifndef NVDR_TRACE_OCC
#define NVDR_TRACE_OCC 8       // waves per SIMD the kernel is compiled for (= resident workgroups per CU of the persistent grid)
#endif

struct TraceLaunch {
    BvhView bvh;
    const float4 *rays;            // stream slot -> (dir.xyz, pdf sum)
    const float4 *pix_origin;      // compacted pixel -> shadow-ray origin
    const uint32_t *live;          // the stream slots to traverse
    const unsigned *ray_count;     // their number (device counter)
    unsigned rays_per_pixel;
    uint8_t *vis;                  // stream slot -> 1 = unoccluded
    int *spill;                    // HBM part of the traversal stacks (bvh.h)
    unsigned long long *counters;  // counting build only (nvdr_hip.h NVDR_COUNTERS_*)
    unsigned *queues;              // [256][32] chunk counters, zeroed before every launch; the last line holds diagnostics
};

// Chunk dealing: wave w uses counter w % 64 and receives the chunks q, q + 64, q + 128 ... of the list, so the whole chip works
// inside ONE moving window of the list.  Every counter must be served by somebody: chunk c sits on counter c % 64 and the
// launcher starts at least min(chunks, 2048) workgroups of 4 waves, so counter c % 64 < waves.
// (Measured and dropped, interleaved in-process A/B, profiles/r02_ab_traversal_variants.md: one contiguous eighth of the list per
// XCD with stealing -- each L2 caching another region of the tree -- is 4-7 % SLOWER on bob and +-2 % on 684 k triangles; static
// dealing without atomics (wave w walks the chunks w, w + waves, ...) is within 1 % of the claiming on 684 k triangles.)
struct ChunkDealer {
    unsigned *queue;               // this wave's counter
    unsigned n_chunks, total, sub;

    __device__ __forceinline__ void init(unsigned *queues, unsigned total_, unsigned wid)
    {
        total = total_;
        n_chunks = (total_ + NVDR_TRACE_QCHUNK - 1u) / NVDR_TRACE_QCHUNK;
        sub = wid % NVDR_TRACE_QUEUES;
        queue = queues + sub * 32u;
    }
    // wave-uniform: claims the next chunk for the whole wave; false = the list is used up
    __device__ __forceinline__ bool claim(int lane, unsigned &next, unsigned &end)
    {
        unsigned j = 0;
        if (lane == 0) j = atomicAdd(queue, 1u);
        j = (unsigned)__builtin_amdgcn_readfirstlane((int)j);
        const unsigned c = j * NVDR_TRACE_QUEUES + sub;
        if (c >= n_chunks) return false;
        next = c * NVDR_TRACE_QCHUNK;
        end = min(next + NVDR_TRACE_QCHUNK, total);
        return true;
    }
};

// COUNT: the counting build (box / triangle tests, per-wave clocks)
template <bool COUNT>
__device__ __forceinline__ void env_trace_body(const TraceLaunch &a, int *smem)
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
    dealer.init(a.queues, total, wid);
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
