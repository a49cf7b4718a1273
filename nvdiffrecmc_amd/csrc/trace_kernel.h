// trace_kernel.h -- stage 2 of env-shade: persistent-wavefront any-hit traversal of the ray stream.
//
// Replaces optixTrace inside __raygen__rg (render/optixutils/c_src/envsampling/kernel.cu:101-118: tmin 0, tmax 1e16, terminate on
// the first hit, no culling).
//
// Round 3 design (round 2 walked four-slot nodes and tested a leaf in the lane that reached it):
//   * EIGHT-WIDE COMPRESSED NODES (bvh.h "oct"): one dependent 64-byte fetch covers three levels of the binary tree; the children
//     of a node are contiguous, so the traversal state is a (first child, 8-bit hit mask) GROUP, and the stack holds one 8-byte
//     group per visited node that still has unvisited internal children -- not one entry per child;
//   * DEFERRED, COMPACTED TRIANGLE TESTS: a lane that hits a leaf box does not test the triangle itself (round 2: ~10 of 64
//     lanes active in that arm, a third of the kernel's VALU work).  It appends (lane, triangle) to a per-wavefront LDS queue and
//     keeps walking as if the triangle had been missed -- any-hit visibility does not depend on the order of the tests.  When 64
//     entries have gathered the WHOLE wavefront tests one triangle per lane (the owner's ray is fetched with ds_bpermute), and a
//     hit ends the owner's walk.  A ray whose walk ends with tests still queued is undecided ("draining") until the queue is
//     flushed; lanes are only refilled right after a complete flush, so a queue entry never outlives the ray it belongs to.
//     (Measured and dropped, round 3, session 25: refilling as soon as 4 / 8 / 16 lanes are DECIDED, without a flush -- entries tagged
//     with their owner's generation, a FIFO queue, a per-lane count of the tests run so that a finished walk knows when its last
//     verdict is in -- is bit-identical and 11-15 % SLOWER at every threshold: the extra permute, LDS counters and votes per iteration
//     cost more than the lanes that idle until 16 are free.)
//   (Round 6, profiles/r06_ab_trace_split_coop_dup.md: (a) SPLIT WALKS in the drain exist as a build of their own for small launches -- see the split round
//   below.  Measured and dropped: (b) A QUAD-COOPERATIVE node fetch -- the four lanes of a quad load the four
//   quarters of ONE node per instruction and transpose inside the quad with DPP quad permutes, a quarter of the cache lines per request: 22-25 % slower; the
//   vector-memory path charges per lane-request, not per line.  (c) Doubling the node's four requests costs +14 ... +22 % on bob, +40 ... +45 % on 684 k
//   triangles: the kernel is sensitive to the number of requests, most where the tree spills the L2.)
#pragma once

#include "bvh.h"

#ifndef NVDR_REFILL_MIN
#define NVDR_REFILL_MIN 16      // refill (and flush) when this many lanes have no node to visit
#endif
// Chunks of NVDR_TRACE_QCHUNK rays are CLAIMED from device counters, one 128-B line each, instead of being dealt
// round-robin: all waves then work inside a moving window of the list (neighbouring pixels -> the same subtrees stay in
// L2) and a wave that drew cheap rays simply claims more.  One counter serialises at ~70 ns per claim (round 1: 1.4-2.5 ms
// with a single counter); 64 of them see < 1 M claims/s each.
// Measured (round 2, same GPU session, 8-view launch, 43 M live rays): static round-robin chunks 4.30 ms, 64 queues x 256
// rays 3.75 ms (-13 %), 128 x 256 4.06 ms, 64 x 128 4.07 ms.
#define NVDR_TRACE_QUEUES 64
#define NVDR_TRACE_QCHUNK 256    // (ChunkDealer::init holds its log2)
#ifndef NVDR_TRACE_OCC
#define NVDR_TRACE_OCC 8       // waves per SIMD the kernel is compiled for (= resident workgroups per CU of the persistent grid)
#endif
#ifndef NVDR_TRACE_STEALS
#define NVDR_TRACE_STEALS 0    // counters of other wavefronts a wavefront tries once its own is used up (A/B only: measured a loss, see ChunkDealer::retarget)
#endif
// Split walks in the drain (env_trace_body<.., SPLIT = true>; the launcher picks that build for SMALL launches, env_shade.hip launch_trace)
#ifndef NVDR_TRACE_SPLIT_FREE
#define NVDR_TRACE_SPLIT_FREE 8    // ... when at least this many lanes have no node to visit
#endif
#ifndef NVDR_TRACE_SPLIT_EVERY
#define NVDR_TRACE_SPLIT_EVERY 1   // ... every so many iterations of the drain (power of two)
#endif
#ifndef NVDR_TRACE_SPLIT_BELOW
#define NVDR_TRACE_SPLIT_BELOW 3000u   // rays per wavefront of the context's last traversal launch below which the split build is launched
#endif
#define NVDR_LEAFQ_CAP 128     // entries of a wavefront's triangle-test queue (< 64 before an append round, <= 64 appended per round)

struct TraceLaunch {
    BvhView bvh;
    const float4 *rays;            // stream slot -> (dir.xyz, pdf sum)
    const float4 *pix_origin;      // compacted pixel -> shadow-ray origin
    const uint32_t *live;          // the stream slots to traverse: NVDR_LIVE_SEGS dense segments
    const unsigned *ray_count;     // the chunk's block of counters: NVDR_LIVE_SEGS segment lengths, the segment capacity (device memory)
    unsigned rays_per_pixel;
    uint8_t *vis;                  // stream slot -> 1 = unoccluded
    int *spill;                    // HBM part of the traversal stacks (bvh.h)
    unsigned long long *counters;  // counting build only (nvdr_hip.h NVDR_COUNTERS_*)
    unsigned *queues;              // [256][32] chunk counters, zeroed before every launch; the last line holds diagnostics
    unsigned top_nodes;            // entries of the treetop table (bvh.h) this launch keeps in LDS; 0: none
};

// LDS of one workgroup of the traversal kernel: per wavefront the oct stack (NVDR_OSTACK_LDS x 64 lanes x 8 B) and the
// triangle-test queue (NVDR_LEAFQ_CAP x 8 B)
// ... and, per workgroup, the treetop table (bvh.h NVDR_TRACE_TOP_NODES: 64 bytes per node)
#define NVDR_TRACE_LDS_PER_WAVE (NVDR_OSTACK_LDS * 64 * 8 + NVDR_LEAFQ_CAP * 8)
#define NVDR_TRACE_LDS_BYTES(threads, top_nodes) ((size_t)((threads) / 64) * NVDR_TRACE_LDS_PER_WAVE + (size_t)(top_nodes) * 64)

// The live-ray list is NVDR_LIVE_SEGS SEGMENTS, each behind its own length counter (round 5): the generation kernel appends per wavefront,
// ~400 entries per claim, and same-address atomics retire one every ~12 ns on this part -- 109 k claims on ONE counter were 1.3 ms of
// serialised time, exactly the kernel's duration once its set-up no longer hid it (profiles/r05_ab_live_segments.md).  Generation wavefront w
// appends to segment w % SEGS; segment s lives at live[s * seg_cap ...] and is dense.  A chunk's block of counters
// (nvdr_ctx::chunk_counts, NVDR_LIVE_WORDS words per chunk of the ray stream): word 32 s = the length of segment s -- ONE COUNTER PER
// 128-BYTE LINE: sixteen counters in one line serialise like one, and worse (7.8 instead of 1.3 ms) -- and word 32 SEGS = seg_cap.
#define NVDR_LIVE_SEGS 16
#define NVDR_LIVE_WORDS ((NVDR_LIVE_SEGS + 1) * 32)
#define NVDR_LIVE_SUBS (NVDR_TRACE_QUEUES / NVDR_LIVE_SEGS)        // dealing counters per segment

// Chunk dealing: counter q = t * SEGS + s hands out the chunks t, t + SUBS, t + 2 SUBS ... of segment s; wave w serves counter w % 64 --
// the launcher starts at least 64 wavefronts, so every counter is served by somebody.  The waves of the chip thus work inside ONE moving window of every segment (the segments hold interleaved pixels:
// neighbouring pixels -> the same subtrees stay in L2); the segments are equally long to a fraction of a percent -- the generation
// wavefronts take the pixels round-robin -- so nobody steals.  (Letting a wavefront that has used its counter up walk on through all 64 --
// two dependent loads per counter -- made the kernel 10 % slower at eight views and 22 % at one: every wavefront paid that walk at the end.)
// (Measured and dropped, interleaved in-process A/B, profiles/r02_ab_traversal_variants.md: one contiguous eighth of the list per
// XCD with stealing -- each L2 caching another region of the tree -- is 4-7 % SLOWER on bob and +-2 % on 684 k triangles; static
// dealing without atomics (wave w walks the chunks w, w + waves, ...) is within 1 % of the claiming on 684 k triangles.)
struct ChunkDealer {
    // (kept small: these are wave-uniform, but every one of them the compiler cannot hold in a scalar register costs the kernel a vector
    // register it does not have -- a dealer with ten fields spilled nine dwords and made the traversal 10 % slower)
    unsigned *qp;                  // the wave's counter
    unsigned first, end_abs;       // list position of that counter's first chunk, end of its segment
    unsigned shift;                // log2 of the chunk size

    // Chunk size: NVDR_TRACE_QCHUNK rays, but a small launch (one view: 5 M rays over 8192 wavefronts = 2.6 chunks of 256 each) is
    // cut finer -- at least ~8 claims per wavefront, 64 rays at the least -- so that the wavefronts run out of work together.
    // Returns the number of live rays of the launch.
    __device__ __forceinline__ unsigned init(unsigned *queues, const unsigned *counts, unsigned wid, unsigned n_waves)
    {
        unsigned total = 0u;
#pragma unroll
        for (int k = 0; k < NVDR_LIVE_SEGS; ++k) total += counts[k * 32];
        shift = 8u;                                                     // log2(NVDR_TRACE_QCHUNK)
#ifndef NVDR_TRACE_COARSE_CHUNKS              // (A/B variants only)
        while (shift > 6u && (total >> shift) < 8u * n_waves) --shift;
#endif
        const unsigned q = wid % NVDR_TRACE_QUEUES;
        qp = queues + q * 32u;
        const unsigned seg = q % NVDR_LIVE_SEGS, t = q / NVDR_LIVE_SEGS;
        const unsigned base = seg * counts[NVDR_LIVE_SEGS * 32];
        first = base + (t << shift);
        end_abs = base + counts[seg * 32u];
        return total;
    }
    // Point the dealer at counter q (round 6: STEALING, an A/B switch -- NVDR_TRACE_STEALS -- that stays off).  The counting launch's per-wavefront
    // clocks show the wavefronts of a one-view launch ending between 40 % and 100 % of the kernel's span, grouped by XCD (47-58 % against 70-82 %,
    // profiles/r06_trace_phase_cycles.md "who ends when"), which looked like counters running dry at different times.  They do not: letting a used-up
    // wavefront try 4 / 12 / 32 / 63 other counters, one per refill round, costs +0.7 ... +17 % at one view and +1 ... +2 % at eight (session 13) -- every
    // counter is used up at about the same time, and what follows is each wavefront's own DRAIN: its last rays, a few of which walk 100+ node steps.
    __device__ __forceinline__ void retarget(unsigned *queues, const unsigned *counts, unsigned q)
    {
        qp = queues + q * 32u;
        const unsigned seg = q % NVDR_LIVE_SEGS, t = q / NVDR_LIVE_SEGS;
        const unsigned base = seg * counts[NVDR_LIVE_SEGS * 32];
        first = base + (t << shift);
        end_abs = base + counts[seg * 32u];
    }
    // wave-uniform: claims the next chunk for the whole wave; false = the wave's counter is used up
    __device__ __forceinline__ bool claim(int lane, unsigned &next, unsigned &end)
    {
        unsigned j = 0u;
        if (lane == 0) j = atomicAdd(qp, 1u);
        j = (unsigned)__builtin_amdgcn_readfirstlane((int)j);
        // chunk j of this counter = chunk j * SUBS + t of the segment
        const unsigned pos = first + (j << (shift + 2u));           // (SUBS = 4; j stays far below 2^22: no wrap)
        if (j >= 0x00400000u || pos >= end_abs) return false;
        next = pos;
        end = min(pos + (1u << shift), end_abs);
        return true;
    }
};
static_assert(NVDR_LIVE_SUBS == 4, "ChunkDealer::claim shifts by log2(NVDR_LIVE_SUBS) = 2");

// 8-byte entries (low word, high word) in explicit address spaces: ds_write_b64 / global_store_dwordx2, no flat accesses
typedef __attribute__((address_space(3))) unsigned long long lds_pair_t;
typedef __attribute__((address_space(1))) unsigned long long glb_pair_t;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4_t;           // (a plain vector: HIP's uint4 class does not live in address spaces)
typedef __attribute__((address_space(3))) u32x4_t lds_uint4_t;
__device__ __forceinline__ uint4 lds_load4(const lds_uint4_t *p) { const u32x4_t v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ unsigned long long pack2(unsigned lo, unsigned hi) { return (unsigned long long)lo | ((unsigned long long)hi << 32); }

// the oct walk's stack of one lane: entry k at LDS word pair (k * 64 + lane), deeper entries in the HBM spill columns
struct OctStack {
    lds_pair_t *lds;
    glb_pair_t *glb;
    int gstride, smax;
    int *ovf;
    __device__ __forceinline__ int push(int sp, unsigned long long v) const
    {
        if (sp >= smax) {    // unreachable unless the depth bound is wrong (bvh.h): the group is dropped, the context flagged
            atomicOr(ovf, 1);
            return sp;
        }
        if (sp < NVDR_OSTACK_LDS) lds[sp * 64] = v;
        else glb[(int64_t)(sp - NVDR_OSTACK_LDS) * gstride] = v;
        return sp + 1;
    }
    __device__ __forceinline__ unsigned long long pop(int sp) const      // entry at depth sp (the caller has already decremented)
    {
        if (sp < NVDR_OSTACK_LDS) return lds[sp * 64];
        return glb[(int64_t)(min(sp, smax - 1) - NVDR_OSTACK_LDS) * gstride];
    }
};

__device__ __forceinline__ float ubyte_f32(unsigned w, int k) { return (float)((w >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyteK

// A ray prepared for the oct walk: plane distance t(p) = p * inv + noi in grid space (as bvh.h make_grid_ray; the reciprocal is
// v_rcp_f32 here: 1 ulp, i.e. < 0.01 grid cells on a plane coordinate, inside the one-cell slack of every box; |inv| capped for the
// reason given there).
struct OctRay {
    float ix, iy, iz, nx, ny, nz;
};

// COUNT: the counting build (box / triangle tests, node steps, per-wave clocks).
// PH: the PHASE-CLOCK builds (round 6; counting launches run them behind the counting kernel, on the same rays): shader-clock cycles of the
// wavefront loop by phase, read at wave-uniform points only (PH = 1: eight scalar accumulators, no vector register, no per-lane counter -- the
// counting build's 17 spilled dwords would drown what is being measured) or, PH = 2, additionally inside the node step with a wait for the
// node behind its four loads (three more vector registers, and the production kernel's overlap of the fetch with the ray's frame set-up is
// gone: this build only splits the node step of PH = 1 into fetch / box arithmetic / stack).
// TOP: the build that keeps the treetop table in LDS (bvh.h, the treetop table; round 6, measured no gain: launched only when the context's table size is
// set, NVDR_TRACE_TOP_NODES -- and by counting launches, whose build is this one so that it can report the steps the table serves).
template <bool COUNT, int PH = 0, bool SPLIT = false, bool TOP = false>
__device__ __forceinline__ void env_trace_body(const TraceLaunch &a, int *smem)
{
    const BvhView &bvh = a.bvh;
    const float4 *__restrict__ rays = a.rays;
    const float4 *__restrict__ pix_origin = a.pix_origin;
    const uint32_t *__restrict__ live = a.live;
    const uint4 *__restrict__ oct = bvh.oct;
    const float4 *__restrict__ tris8 = bvh.tris8;
    uint8_t *__restrict__ vis = a.vis;
    unsigned long long *counters = a.counters;
    const unsigned rays_per_pixel = a.rays_per_pixel;
    // slot -> pixel: a shift when the pixel's 2 n^2 slots are a power of two (n_samples_x = 8, 16, ...), the division otherwise (wave-uniform)
    const bool rpp_pow2 = (rays_per_pixel & (rays_per_pixel - 1u)) == 0u;
    const unsigned rpp_shift = (unsigned)__builtin_ctz(rays_per_pixel | 0x80000000u);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *wbase = (char *)smem + wave * NVDR_TRACE_LDS_PER_WAVE;
    OctStack stack;
    stack.lds = (lds_pair_t *)wbase + lane;
    stack.gstride = blockDim.x;
    stack.smax = bvh.oct_stack_max;
    stack.ovf = bvh.overflow;
    stack.glb = (glb_pair_t *)a.spill + (int64_t)blockIdx.x * blockDim.x * max(bvh.oct_stack_max - NVDR_OSTACK_LDS, 0) + threadIdx.x;
    lds_pair_t *leafq = (lds_pair_t *)(wbase + NVDR_OSTACK_LDS * 64 * 8);      // (triangle, owner lane) entries, used as a stack
    // the treetop table, behind the wavefronts' regions: a straight copy of what the build left behind oct[] (bvh.hip bvh_oct_top_kernel)
    lds_uint4_t *top = (lds_uint4_t *)((char *)smem + (blockDim.x >> 6) * NVDR_TRACE_LDS_PER_WAVE);
    if (TOP) {
        for (unsigned i = threadIdx.x; i < a.top_nodes * 4u; i += blockDim.x) { const uint4 v = bvh.oct_top[i]; u32x4_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; top[i] = w; }
        __syncthreads();
    }
    // the grid transform, wave-uniform (scalar registers)
    const BvhDeviceInfo *__restrict__ info = bvh.info;
    const float gsx = info->g_scale[0], gsy = info->g_scale[1], gsz = info->g_scale[2];
    const float glx = info->g_lo[0], gly = info->g_lo[1], glz = info->g_lo[2];

    const unsigned wid = blockIdx.x * (blockDim.x >> 6) + wave;
    ChunkDealer dealer;
    const unsigned total = dealer.init(a.queues, a.ray_count, wid, gridDim.x * (blockDim.x >> 6));
    // (host-mapped word: the launcher's hint for the NEXT launch -- which build to start, env_shade.hip launch_trace; written here, in front of the loop,
    // so that nothing of it stays live across the loop)
    if (!COUNT && !PH && wid == 0u && lane == 0) bvh.overflow[1] = (int)min(total, 0x7fffffffu);
    unsigned next = 0, end = 0;                             // wave-uniform list positions of the claimed chunk
    unsigned steals = 0u;                                   // wave-uniform: other wavefronts' counters tried so far
    bool more = total > 0;
    unsigned n_box = 0, n_tri = 0, n_ray = 0, n_step = 0, n_batch = 0, n_top = 0;
    // phase-clock builds: shader-clock cycles of this wavefront by phase of the loop (wave-uniform; nvdr_hip.h NVDR_COUNTERS_PHASES)
    // (32-bit: a launch lasts a few million cycles, differences are taken modulo 2^32; eight scalar registers instead of sixteen)
    unsigned ph_refill = 0u, ph_fetch = 0u, ph_box = 0u, ph_stack = 0u, ph_queue = 0u, ph_batch = 0u, ph_iters = 0u, ph_steps = 0u;
    auto clk = [&]() -> unsigned { return PH ? (unsigned)__builtin_readcyclecounter() : 0u; };
    const unsigned long long t_begin = COUNT ? wall_clock64() : 0ull;
    const unsigned long long c_begin = (COUNT || PH) ? (unsigned long long)__builtin_readcyclecounter() : 0ull;

    // per-lane walk state.  ray < 0: idle.  ray >= 0 and (gbits | sp) == 0: the walk is over, queued tests decide ("draining").
    int ray = -1, sp = 0;
    unsigned gbase = 0, gbits = 0;                          // current group: first child, hit bits of the children still to visit
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    OctRay g;
    g.ix = g.iy = g.iz = g.nx = g.ny = g.nz = 0.0f;
    unsigned q_count = 0;                                   // wave-uniform fill of the triangle-test queue
    bool was_split = false;                                 // wave-uniform: some walk of this wavefront has been split (drain mode)
#ifdef NVDR_TRACE_TOUCH
    unsigned touch = 0u;
#endif
#ifdef NVDR_TRACE_PRELIVE
    unsigned pre = 0u, pre_base = 0u;                       // list entry pre_base + lane, fetched ahead (pre_base wave-uniform)
#endif

    // One batch of triangle tests: the top n entries of the queue, one per lane.  Returns nothing; a hit ends the owner's walk
    // (visibility 0) whatever the owner is doing.  Entries of rays that have ended meanwhile test against the owner's stale
    // registers and can only "kill" an idle lane: harmless, because lanes are refilled only when the queue is empty.
    auto test_batch = [&](unsigned n) {
        const unsigned tb0 = clk();
        __builtin_amdgcn_wave_barrier();                    // the entries were written by other lanes of this wavefront
        const unsigned first = q_count - n;
        const bool valid = (unsigned)lane < n;
        unsigned long long e = pack2(0u, (unsigned)lane);
        if (valid) e = leafq[first + lane];
        const unsigned e_tri = (unsigned)e, e_own = (unsigned)(e >> 32);
        const int src = (int)(e_own << 2);
        // the owner's ray (all lanes execute the permutes: a source lane must be active to be read)
        const float rox = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ox)));
        const float roy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(oy)));
        const float roz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(oz)));
        const float rdx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dx)));
        const float rdy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dy)));
        const float rdz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dz)));
        bool hit = false;
        if (valid) hit = tri_any_hit(tris8, (int)e_tri, rox, roy, roz, rdx, rdy, rdz);
        if (COUNT) { n_tri += valid ? 1u : 0u; n_batch += lane == 0 ? 1u : 0u; }
        // owners of the hits, gathered on the scalar unit (hits are rare: one per occluded ray)
        unsigned long long hm = __ballot(hit), kill = 0ull;
        bool same_ray = false;                              // (split walks: this lane holds a copy of a ray another lane just found occluded)
        while (hm) {
            const int l = __builtin_ctzll(hm);
            hm &= hm - 1ull;
            const unsigned own = (unsigned)__builtin_amdgcn_readlane((int)e_own, l);
            kill |= 1ull << own;
            if (SPLIT && was_split) same_ray = same_ray || ray == __builtin_amdgcn_readlane(ray, (int)own);
        }
        q_count = first;
        if ((((kill >> lane) & 1ull) || same_ray) && ray >= 0) {
            vis[ray] = 0;
            ray = -1;
            sp = 0;
            gbits = 0u;
        }
        if (PH) ph_batch += clk() - tb0;
    };

    unsigned iters = 0;
    while (true) {
        // every iteration retires work, so the loop ends by itself; the bound (far above any real launch: 2^23 iterations of one
        // wavefront, seconds; the 8-view benchmark launch takes ~750) only turns a bug into an error report instead of a hung GPU
        if (++iters > (1u << 23)) {
            if (lane == 0) atomicOr(bvh.overflow, 4);
            break;
        }
        // ---- refill: when enough lanes have no node to visit, decide the undecided rays and hand out new ones
        const unsigned tr0 = clk(), bt0 = ph_batch;
        const unsigned long long busy = __ballot(ray >= 0 && (gbits | (unsigned)sp) != 0u);
        const int n_free = 64 - __popcll(busy);
        if (n_free >= NVDR_REFILL_MIN && next >= end && more) {
            more = dealer.claim(lane, next, end);
#ifdef NVDR_TRACE_PRELIVE
            if (more) { pre = live[min(next + (unsigned)lane, end - 1u)]; pre_base = next; }
#endif
            if (!more) {
                next = end = 0u;
                if (steals < NVDR_TRACE_STEALS) {           // this counter is used up: the next refill round asks another one
                    ++steals;
                    dealer.retarget(a.queues, a.ray_count, (wid + 17u * steals) % NVDR_TRACE_QUEUES);
                    more = true;
                }
            }
        }
        if ((n_free >= NVDR_REFILL_MIN && next < end) || busy == 0ull) {
            while (q_count > 0u) test_batch(min(q_count, 64u));
            if (ray >= 0 && (gbits | (unsigned)sp) == 0u) {         // walk over, every queued test missed: unoccluded
                if (!SPLIT) vis[ray] = 1;                // (split walks: the byte was written at the fetch, a copy must not write it)
                ray = -1;
            }
            if (next < end) {
                const unsigned long long idle = __ballot(ray < 0);
                const unsigned take = next + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
#ifdef NVDR_TRACE_PRELIVE
                // (A/B only, measured +2 ... +3.6 %, profiles/r06_ab_trace_prelive.md: the list entries of this refill were fetched behind the previous
                // one -- entry pre_base + l waits in lane l -- so the chain list entry -> ray -> set-up starts one memory round trip later; entries
                // beyond the 64 fetched ones are read directly)
                const unsigned rel = take - pre_base;
                const unsigned pre_slot = (unsigned)__builtin_amdgcn_ds_bpermute((int)((rel & 63u) << 2), (int)pre);
#endif
                if (ray < 0 && take < end) {
#ifdef NVDR_TRACE_PRELIVE
                    const unsigned slot = rel < 64u ? pre_slot : live[take];
#else
                    const unsigned slot = live[take];
#endif
                    ray = (int)slot;
                    if (SPLIT) vis[slot] = 1;             // unoccluded until some lane that holds the ray finds a hit
                    if (COUNT) n_ray++;
                    const float4 rd = rays[slot];
                    const float4 ro = pix_origin[rpp_pow2 ? slot >> rpp_shift : slot / rays_per_pixel];
                    ox = ro.x; oy = ro.y; oz = ro.z;
                    dx = rd.x; dy = rd.y; dz = rd.z;
                    g.ix = fminf(fmaxf(__builtin_amdgcn_rcpf(dx * gsx), -1.0e30f), 1.0e30f);
                    g.iy = fminf(fmaxf(__builtin_amdgcn_rcpf(dy * gsy), -1.0e30f), 1.0e30f);
                    g.iz = fminf(fmaxf(__builtin_amdgcn_rcpf(dz * gsz), -1.0e30f), 1.0e30f);
                    g.nx = -((ox - glx) * gsx + 2.0f) * g.ix;
                    g.ny = -((oy - gly) * gsy + 2.0f) * g.iy;
                    g.nz = -((oz - glz) * gsz + 2.0f) * g.iz;
                    gbase = (TOP && a.top_nodes) ? NVDR_OCT_TOP_FLAG : 0u;    // the root is "child 0 of group 0" (of the treetop table, if there is one)
                    gbits = 1u;
                    sp = 0;
                }
                next += (unsigned)__popcll(idle);
#ifdef NVDR_TRACE_PRELIVE
                if (next < end) { pre = live[min(next + (unsigned)lane, end - 1u)]; pre_base = next; }
#endif
            } else if (__ballot(ray >= 0) == 0ull) {
                if (!more) break;
                continue;                                   // nothing to do until the next claim
            }
        }

        // ---- drain (the SPLIT build): no ray left to claim and lanes without a node to visit -- every lane with two or more pending items hands its
        // OLDEST one to an idle lane, which takes a copy of the ray (15 ds_bpermute) and walks that group with a stack of its own.  An any-hit walk may
        // visit its pending groups in any order and by any lane; the oldest group is the one nearest the root, i.e. the largest share of what is left
        // (handing over the NEWEST entry -- the smallest subtree -- gained nothing, session 6).  The stack is a bag: the newest entry moves into the slot
        // of the one given away.  A hit found by any copy ends all lanes that hold the ray; the "unoccluded" byte is written when a ray is FETCHED
        // and only ever overwritten (a copy that ends without a hit must not write: another may have found one).
        // Measured (session 14, in-process A/B): -4 ... -5 % per one-view launch of bob (700 rays per wavefront), -6 % on 684 k triangles (1 800 per
        // wavefront), but +3 ... +6 % on launches of 5 000 rays per wavefront (the block's registers and tests are paid in every iteration, the drain
        // is a tenth of such a launch): a build of its own, launched for small launches only.
        if (SPLIT && !more && next >= end && n_free >= NVDR_TRACE_SPLIT_FREE && (iters & (NVDR_TRACE_SPLIT_EVERY - 1u)) == 0u &&
            __ballot(ray >= 0 && (sp > 0 || (gbits & (gbits - 1u)) != 0u)) != 0ull) {
            while (q_count > 0u) test_batch(min(q_count, 64u));      // the queue's entries refer to the lanes as they are now
            if (ray >= 0 && (gbits | (unsigned)sp) == 0u) ray = -1;
            const bool donor = ray >= 0 && (sp > 0 || (gbits & (gbits - 1u)) != 0u);
            const unsigned long long dm = __ballot(donor), idle = __ballot(ray < 0);
            const unsigned np = min((unsigned)__popcll(dm), (unsigned)__popcll(idle));      // pairs: the k-th idle lane takes from the k-th donor
            if (np != 0u) {
                const unsigned drank = (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dm, 0u));
                const unsigned irank = (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                unsigned gift_base = 0u, gift_bits = 0u;
                if (donor && drank < np) {
                    if (sp > 0) {                           // the oldest waiting group (entry 0); the newest takes its slot
                        const unsigned long long bottom = stack.pop(0);
                        gift_base = (unsigned)bottom;
                        gift_bits = (unsigned)(bottom >> 32);
                        sp--;
                        if (sp > 0) stack.lds[0] = stack.pop(sp);
                    } else {                                // the further children of the current group
                        const unsigned keep = gbits & (0u - gbits);
                        gift_base = gbase;
                        gift_bits = gbits ^ keep;
                        gbits = keep;
                    }
                    leafq[drank] = pack2((unsigned)lane, 0u);       // (the queue is empty: its LDS serves as the pairing table)
                }
                __builtin_amdgcn_wave_barrier();
                const bool taker = ray < 0 && irank < np;
                const int src = taker ? (int)((unsigned)leafq[irank] << 2) : (lane << 2);
                __builtin_amdgcn_wave_barrier();
                // (all lanes execute the permutes: a source lane must be active to be read.  A lane that takes nothing reads itself, so the
                // ray state is permuted in place, one register at a time)
                auto perm = [&](float &x) { x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(x))); };
                const unsigned c_base = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)gift_base), c_bits = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)gift_bits);
                ray = __builtin_amdgcn_ds_bpermute(src, ray);
                perm(ox); perm(oy); perm(oz); perm(dx); perm(dy); perm(dz);
                perm(g.ix); perm(g.iy); perm(g.iz); perm(g.nx); perm(g.ny); perm(g.nz);
                if (taker) { gbase = c_base; gbits = c_bits; sp = 0; }
                was_split = true;
            }
        }
        // ---- node step of every lane that has one
        unsigned leaf_bits = 0u, leaf_base = 0u;
        unsigned tn0 = 0u;
        unsigned dt_fetch = 0u, dt_box = 0u, dt_stack = 0u;     // (per lane: written under divergent control; one active lane's copy is read back below)
        if (PH) {
            tn0 = clk();
            ph_refill += (tn0 - tr0) - (ph_batch - bt0);      // (the flush of the refill is accounted as triangle batches)
            ph_iters++;
            if (__ballot(ray >= 0 && (gbits | (unsigned)sp) != 0u) != 0ull) ph_steps++;
        }
        if (ray >= 0 && (gbits | (unsigned)sp) != 0u) {
            if (gbits == 0u) {
                sp--;
                const unsigned long long top = stack.pop(sp);
                gbase = (unsigned)top;
                gbits = (unsigned)(top >> 32);
            }
            const int k = __builtin_ctz(gbits);
            gbits &= gbits - 1u;
            const unsigned at = gbase + (unsigned)k;
            const uint4 *nd = oct + 4 * (int64_t)at;
            uint4 h, p1, p2, p3;
            // a node of the treetop comes out of LDS: four ds_read_b128 instead of four vector-memory requests (the rays of a wavefront start at the
            // same nodes: mostly broadcast reads)
            if (TOP && (at & NVDR_OCT_TOP_FLAG)) {
                const lds_uint4_t *tn = top + 4u * (at & (NVDR_OCT_TOP_FLAG - 1u));
                h = lds_load4(tn); p1 = lds_load4(tn + 1); p2 = lds_load4(tn + 2); p3 = lds_load4(tn + 3);
            } else {
                h = nd[0]; p1 = nd[1]; p2 = nd[2]; p3 = nd[3];
            }
#ifdef NVDR_TRACE_DUP_FETCH
            // (A/B only: the same four 16-byte requests once more, results thrown away -- twice the L1 look-ups and returned lines per node
            // step with nothing else changed: what a kernel bound by the L1's request rate slows down by, a latency-bound one barely notices)
            if (!TOP || !(at & NVDR_OCT_TOP_FLAG)) {
                const volatile uint4 *ndv = (const volatile uint4 *)nd;
#pragma unroll
                for (int q = 0; q < NVDR_TRACE_DUP_FETCH; ++q) {
                    const uint4 d = const_cast<const uint4 &>(ndv[q]);
                    asm volatile("" :: "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
                }
            }
#endif
            if (COUNT) { n_step++; n_top += (TOP && (at & NVDR_OCT_TOP_FLAG)) ? 1u : 0u; }
            if (PH == 2) {
                // (this build waits for the node here, so that the fetch has a phase of its own; the production kernel lets the
                // compiler place the wait: the ray's set-up of the node frame overlaps part of it)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                dt_fetch = (unsigned)(clk() - tn0);
            }
            // the node's frame: plane = org + q * 2^e  ->  t = q * (inv * 2^e) + (org * inv + noi)
            const float ax = __builtin_ldexpf(g.ix, (int)((h.y >> 16) & 15u)), bx = fmaf((float)(h.x & 0xffffu), g.ix, g.nx);
            const float ay = __builtin_ldexpf(g.iy, (int)((h.y >> 20) & 15u)), by = fmaf((float)(h.x >> 16), g.iy, g.ny);
            const float az = __builtin_ldexpf(g.iz, (int)((h.y >> 24) & 15u)), bz = fmaf((float)(h.y & 0xffffu), g.iz, g.nz);
            // (near, far) byte planes of each axis for this ray's direction: lo.x = p1.xy, lo.y = p1.zw, lo.z = p2.xy, hi.x = p2.zw, hi.y = p3.xy, hi.z = p3.zw
            const bool sx = g.ix < 0.0f, sy = g.iy < 0.0f, sz = g.iz < 0.0f;
            const unsigned nx0 = sx ? p2.z : p1.x, nx1 = sx ? p2.w : p1.y, fx0 = sx ? p1.x : p2.z, fx1 = sx ? p1.y : p2.w;
            const unsigned ny0 = sy ? p3.x : p1.z, ny1 = sy ? p3.y : p1.w, fy0 = sy ? p1.z : p3.x, fy1 = sy ? p1.w : p3.y;
            const unsigned nz0 = sz ? p3.z : p2.x, nz1 = sz ? p3.w : p2.y, fz0 = sz ? p2.x : p3.z, fz1 = sz ? p2.y : p3.w;
            unsigned miss = 0u;                             // bit j = slot j is missed, shifted in from slot 7 down to slot 0
#pragma unroll
            for (int j = 7; j >= 0; --j) {
                const unsigned wnx = j < 4 ? nx0 : nx1, wny = j < 4 ? ny0 : ny1, wnz = j < 4 ? nz0 : nz1;
                const unsigned wfx = j < 4 ? fx0 : fx1, wfy = j < 4 ? fy0 : fy1, wfz = j < 4 ? fz0 : fz1;
                const float tnx = fmaf(ubyte_f32(wnx, j & 3), ax, bx), tfx = fmaf(ubyte_f32(wfx, j & 3), ax, bx);
                const float tny = fmaf(ubyte_f32(wny, j & 3), ay, by), tfy = fmaf(ubyte_f32(wfy, j & 3), ay, by);
                const float tnz = fmaf(ubyte_f32(wnz, j & 3), az, bz), tfz = fmaf(ubyte_f32(wfz, j & 3), az, bz);
                const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
                const float tf = fminf(fminf(tfx, tfy), tfz);
                miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn), 31u);     // (miss << 1) | sign(tf - tn)
            }
            // (Measured and dropped, round 3, session 19: the (near, far) distances of an axis as ONE v_pk_fma_f32 -- 24 instead of 48 fma
            // instructions per node step, bit-identical -- is 5 % SLOWER (2.577 vs 2.455 ms per 8-view launch, 8.68 vs 8.35 ms on 684 k
            // triangles): packed fp32 issues at 2.8 cycles against 1.85 for v_fma_f32 at this occupancy and needs its operands in aligned
            // register pairs, which cost copies and three more spilled dwords.)
            // (Measured and dropped, round 3: bytes -> distances through v_perm_b32 + v_fma_mix_f32 -- two bytes and the constant
            // 0x64 make two f16 values 1024 + q, which an f32 fma consumes directly: 3 VALU per byte pair instead of 4, exact to 8e-5
            // quantisation steps -- is NOT faster: v_fma_mix_f32 and v_perm_b32 issue at 2.6-2.7 cycles against 2.1 / 1.9 for the
            // conversion and the fma; 2.501 vs 2.488 ms per 8-view launch.  tools/ubench/fma_mix.hip, profiles/r03_fma_mix_ubench.txt.)
            const unsigned n_int = h.z >> 28, n_leaf = h.w >> 28;
            const unsigned hits = ~miss & ((1u << (n_int + n_leaf)) - 1u);
            if (COUNT) n_box += n_int + n_leaf;
            if (PH == 2) {
                asm volatile("" :: "v"(hits));                 // (the box arithmetic ends here)
                dt_box = (unsigned)(clk() - tn0) - dt_fetch;
            }
            const unsigned hi = hits & ((1u << n_int) - 1u);
            leaf_bits = hits >> n_int;
            leaf_base = h.w & (NVDR_OCT_MAX_INDEX - 1u);
            if (hi != 0u) {
                // descend into the new group; the old one waits on the stack if it still has children to visit
                if (gbits != 0u) sp = stack.push(sp, pack2(gbase, gbits));
                gbase = h.z & (NVDR_OCT_MAX_INDEX - 1u);
                gbits = hi;
            }
            if (PH == 2) {
                asm volatile("" :: "v"(gbits), "v"(sp));
                dt_stack = (unsigned)(clk() - tn0) - dt_fetch - dt_box;
            }
#ifdef NVDR_TRACE_TOUCH
            // (A/B only, measured +3 ... +19 %, profiles/r06_ab_trace_treetop.md: the node this lane visits NEXT is known here -- the first child of the group it just entered, or the next sibling -- while
            // the leaf-queue rounds, a triangle batch and the refill votes still lie between this point and its fetch: one 4-byte request for its
            // line now, never consumed, so that the fetch finds it in the L1)
            asm volatile("" :: "v"(touch));
            if (gbits != 0u && !(TOP && (gbase & NVDR_OCT_TOP_FLAG))) touch = ((const unsigned *)(oct + 4 * (int64_t)(gbase + (unsigned)__builtin_ctz(gbits))))[NVDR_TRACE_TOUCH - 1];
#endif
        }
        unsigned tq0 = 0u, bq0 = 0u;
        if (PH) {
            tq0 = clk();
            bq0 = ph_batch;
            if (PH == 2) {
                const unsigned long long am = __ballot(dt_fetch != 0u);
                if (am != 0ull) {
                    const int l = __builtin_ctzll(am);
                    ph_fetch += (unsigned)__builtin_amdgcn_readlane((int)dt_fetch, l);
                    ph_box += (unsigned)__builtin_amdgcn_readlane((int)dt_box, l);
                    ph_stack += (unsigned)__builtin_amdgcn_readlane((int)dt_stack, l);
                }
            } else {
                ph_fetch += tq0 - tn0;                         // PH = 1: the whole node step (fetch + box arithmetic + stack) in one figure
            }
        }

        // ---- queue the hit leaves: one entry per lane and round, ranks by ballot; a full queue is tested at once
        while (true) {
            const unsigned long long has = __ballot(leaf_bits != 0u);
            if (has == 0ull) break;
            if (leaf_bits != 0u) {
                const unsigned pos = q_count + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(has >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)has, 0u));
                const int j = __builtin_ctz(leaf_bits);
                leaf_bits &= leaf_bits - 1u;
                leafq[pos] = pack2(leaf_base + (unsigned)j, (unsigned)lane);
            }
            q_count += (unsigned)__popcll(has);
            if (q_count >= 64u) {
                test_batch(64u);
                if (ray < 0) leaf_bits = 0u;                // the owner was just found occluded: its other leaves do not matter
            }
        }
        if (PH) ph_queue += (clk() - tq0) - (ph_batch - bq0);
    }
    if (PH) {
        if (lane == 0) {
            unsigned long long *pc = counters + NVDR_COUNTERS_PHASES + (PH == 2 ? 8 : 0);
            const unsigned long long total = (unsigned long long)__builtin_readcyclecounter() - c_begin;
            atomicAdd(&pc[0], (unsigned long long)ph_refill);
            atomicAdd(&pc[1], (unsigned long long)ph_fetch);            // PH = 1: the whole node step
            if (PH == 2) {
                atomicAdd(&pc[2], (unsigned long long)ph_box);
                atomicAdd(&pc[3], (unsigned long long)ph_stack);
                atomicAdd(&pc[4], (unsigned long long)ph_queue);
                atomicAdd(&pc[5], (unsigned long long)ph_batch);
                atomicAdd(&pc[6], total);
                atomicAdd(&pc[7], (unsigned long long)ph_iters);
            } else {
                atomicAdd(&pc[2], (unsigned long long)ph_queue);
                atomicAdd(&pc[3], (unsigned long long)ph_batch);
                atomicAdd(&pc[4], (unsigned long long)ph_iters);
                atomicAdd(&pc[5], (unsigned long long)ph_steps);
                atomicAdd(&pc[6], total);
                atomicAdd(&pc[7], 1ull);
            }
        }
    }
    if (COUNT) {
        for (int o = 32; o >= 1; o >>= 1) {
            n_box += __shfl_xor(n_box, o);
            n_tri += __shfl_xor(n_tri, o);
            n_ray += __shfl_xor(n_ray, o);
            n_step += __shfl_xor(n_step, o);
            n_batch += __shfl_xor(n_batch, o);
            n_top += __shfl_xor(n_top, o);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], (unsigned long long)n_box);
            atomicAdd(&counters[1], (unsigned long long)n_tri);
            atomicAdd(&counters[2], (unsigned long long)n_ray);
            atomicAdd(&counters[NVDR_COUNTERS_BVH2 + 3], (unsigned long long)n_step);
            atomicAdd(&counters[NVDR_COUNTERS_BVH2 + 4], (unsigned long long)n_batch);
            atomicAdd(&counters[NVDR_COUNTERS_BVH2 + 5], (unsigned long long)n_tri);
            atomicAdd(&counters[NVDR_COUNTERS_BVH2 + 6], (unsigned long long)n_top);
            // load balance: sum and maximum of the per-wave busy time (100 MHz ticks), wave count
            const unsigned long long dt = wall_clock64() - t_begin;
            atomicAdd(&counters[3], dt);
            atomicMax(&counters[4], dt);
            atomicAdd(&counters[5], 1ull);
            // shader-clock cycles spent (sum over waves; / counters[3] = cycles per 100 MHz tick, i.e. the clock the waves
            // actually ran at) and the set of XCDs that ran waves
            atomicAdd(&counters[6], (unsigned long long)__builtin_readcyclecounter() - c_begin);
            atomicOr(&counters[7], 1ull << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u));
            if (wid < 8192u) {
                counters[8 + 2 * wid] = t_begin;                 // per-wave begin / end ticks; the XCD it ran on in the top byte
                counters[9 + 2 * wid] = (t_begin + dt) | ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) << 56);
            }
        }
    }
}
