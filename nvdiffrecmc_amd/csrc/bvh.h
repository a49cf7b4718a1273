// bvh.h -- device BVH layout, context object, and the per-lane traversal routines.
//
// Replaces the OptiX GAS + optixTrace of the reference (optixAccelBuild at
// render/optixutils/c_src/torch_bindings.cpp:97-110, shadow_test at envsampling/kernel.cu:101-118).
//
// Layout in HBM (see DESIGN.md "Data layout"):
//   nodes[n]  : 2 x uint4 = 32 B per internal node, two nodes per 64-B line.  The boxes of BOTH children
//       live in the parent, quantised to a 16-bit grid spanned over the (padded) scene AABB:
//         w0 = lmin.x | lmin.y << 16   w1 = lmin.z | lmax.x << 16   w2 = lmax.y | lmax.z << 16
//         w3 = rmin.x | rmin.y << 16   w4 = rmin.z | rmax.x << 16   w5 = rmax.y | rmax.z << 16
//         w6 = left child              w7 = right child
//       child >= 0: internal node index; child < 0: ~child = leaf slot (one triangle per leaf).
//       Minima are rounded down and maxima up (plus one grid cell of slack), so a quantised box always
//       contains the exact one: culling stays conservative and visibility stays bit-identical to the
//       brute-force oracle.  WHY 32 B: a divergent traversal step is bound by the vector-L1 lookup rate
//       (one 16-B piece per lane per lookup), not by bytes: measured on MI355X, re-touching the same
//       lines twice as often costs +65 %.  Two lookups per node instead of four is the lever.
//       The ray is transformed into grid space once (o' = (o - g_lo) * g_scale, d' = d * g_scale), which
//       leaves the ray parameter t unchanged, so the slab test runs directly on the decoded integers.
//   oct[m]    : 4 x uint4 = 64 B per EIGHT-WIDE node (what the shadow-ray traversal of env-shade walks; csrc/trace_kernel.h).  Collapsed
//       from nodes[] after the fit (bvh.hip: the SAH-optimal treelets chosen by the dynamic programme of bvh_fit_kernel, resolved
//       into per-node slot budgets by pointer jumping, counted, prefix-summed and emitted -- bvh_oct_budget / count / emit kernels;
//       the layout is a function of the tree alone).  Slots are
//       ordered internal children first (n_int of them, stored CONTIGUOUSLY in oct[] from child_base: no per-child index), then
//       leaves (n_leaf triangles, stored contiguously in tris8[] from tri_base), then empty slots.  Child boxes are re-quantised
//       to 8 bits in the NODE'S OWN frame -- origin = the node's lower corner on the 16-bit grid, one power-of-two cell size
//       per axis -- rounded outward, so they contain the 16-bit boxes, which contain the exact ones:
//         w0.x = org.x | org.y << 16      w0.y = org.z | ex << 16 | ey << 20 | ez << 24       (plane = org + q << e)
//         w0.z = child_base | n_int << 28 w0.w = tri_base | n_leaf << 28
//         w1 = lo.x[0..7], lo.y[0..7]     w2 = lo.z[0..7], hi.x[0..7]      w3 = hi.y[0..7], hi.z[0..7]   (one byte per slot)
//       8 B per child instead of 16 B (four-slot nodes of round 2) and ~n/5 nodes instead of n: bob 0.15 MB instead of 0.68 MB,
//       684 k triangles 9 MB instead of 44 MB; one dependent 64-B fetch per THREE tree levels.
//   tris8[k]  : the triangle records in the order the oct nodes refer to them (a node's leaf children are adjacent).
//   tris[k]   : 3 x float4 = 48 B per triangle in Morton order, world space, full precision:
//       (v0.xyz, e1.x) (e1.yz, e2.xy) (e2.z, orig_index_bits, 0, 0) -- the hit predicate itself
//       (include/nvdr_raytri.h) never sees quantised data.
#pragma once

#include "common.h"
#include "nvdr_raytri.h"

// Traversal stacks.  The BINARY walk (test hooks, closest hit, counting) keeps the first NVDR_STACK_LDS entries of every lane in
// LDS (entry k of lane l at word k*64 + l: one bank per lane, conflict-free); deeper entries spill to a per-lane column of an
// HBM scratch buffer owned by the context.  The OCT walk (shadow rays) keeps (child group, remaining hit bits) PAIRS: one 8-byte
// entry per visited node that still has unvisited internal children, NVDR_OSTACK_LDS of them in LDS (ds_write_b64 / ds_read_b64,
// conflict-free at 8 B per lane), the rest in the same spill buffer.
//
// WHY THE STACKS CANNOT OVERFLOW.  A Karras node is identified by the common-prefix length delta of its key range and
// delta strictly grows from parent to child.  Distinct 30-bit keys give delta = clz(a ^ b) in [2, 31] (30 values), equal
// keys give delta = 32 + clz(i ^ j) with i, j < n (at most ceil(log2 n) values): a root-to-leaf path holds at most
// h_max = 30 + ceil(log2 n) <= 60 internal nodes whatever the mesh (100 k triangles on one centroid: a balanced tree over
// the index bits).  The binary walk holds <= h_max entries.  Every level of the oct tree descends at least one binary level, and
// the oct walk pushes at most one entry per level of its path, so it holds <= h_max entries too.  (Round 5: the bottom subtrees of up to
// 64 leaves are rebuilt by agglomerative clustering, bvh.hip TREELETS; a rebuilt treelet is committed only if it has at most
// NVDR_TREELET_CAP levels, so a path holds <= h_max + NVDR_TREELET_CAP internal nodes.)  nvdr_bvh_build sizes the spill
// columns from this bound (nvdr_stack_bound); a push beyond it -- unreachable unless the bound is wrong -- raises the context's
// overflow flag (host-mapped memory), which every later call on the context and nvdr_ctx_check() turn into an error instead
// of a silently wrong visibility (tests/test_gpu_bvh.py feeds degenerate meshes).
#ifndef NVDR_STACK_LDS
#define NVDR_STACK_LDS 12
#endif
#define NVDR_STACK_MAX 104
#define NVDR_TREELET_CAP 16                  // levels of a rebuilt treelet (bvh.hip); deeper ones keep their Karras subtree
#define NVDR_QUERY_BLOCK 256                 // threads per workgroup of every traversal kernel
#ifndef NVDR_QUERY_MAX_BLOCKS
#define NVDR_QUERY_MAX_BLOCKS 2048           // persistent / grid-stride launches never exceed this
#endif
#define NVDR_PROF_RING 512
#define NVDR_MAX_CHUNKS 1024                 // chunks of the env-shade ray stream one launch may be cut into
#define NVDR_TRAV_DONE 0x7fffffff            // traversal marker: nothing left (never a valid node / leaf id)
#define NVDR_OSTACK_LDS 6                    // oct walk: (child group, hit bits) entries per lane kept in LDS (8 B each); deeper ones spill
#define NVDR_OCT_MAX_INDEX (1 << 28)         // child_base / tri_base share their word with a 4-bit count
// The TREETOP TABLE of the eight-wide tree (round 6; OFF by default -- measured no gain): the first K nodes in breadth-first order, copied
// behind oct[] by bvh_oct_top_kernel with their child groups renumbered -- bit 27 of child_base says "this group lives in the table, at this
// entry".  The TOP build of the shadow-ray kernel keeps the table in LDS (trace_kernel.h): every ray starts there, and a node step inside it
// costs four LDS reads instead of four vector-memory requests.  K = NVDR_TRACE_TOP_NODES (tuning switch, read when the context is created;
// 64 entries = 4 KB is what eight resident workgroups per CU leave free).  (Triangle count limit 2^27 so that bit 27 of a node index is free.)
#ifndef NVDR_TRACE_TOP_MAX
#define NVDR_TRACE_TOP_MAX 1024         // capacity of the table
#endif
#define NVDR_OCT_TOP_FLAG (1u << 27)
#define NVDR_GRID_MAX 65531.0f               // usable span of the 16-bit box grid (2 cells of slack on both ends)

struct BvhDeviceInfo {
    int bounds[6];          // vertex AABB as order-preserving ints (min xyz, max xyz)
    int height;             // tree height in internal nodes
    int root;               // root node index (Karras numbering: 0)
    unsigned int pix_count; // env-shade: number of covered pixels appended to the work list
    float pad;              // world-space padding applied to every leaf box
    float g_lo[3];          // quantisation grid: world -> grid is (x - g_lo) * g_scale + 2
    float g_scale[3];
    unsigned int ray_count; // (unused since the chunked ray stream: the per-chunk counters live in nvdr_ctx::chunk_counts)
};

// upper bound of the traversal stack depth (binary and oct walk alike) for a tree over n triangles (see the note above)
static inline int nvdr_stack_bound(int64_t n_tris)
{
    int lg = 0;
    while ((1ll << lg) < n_tris) ++lg;
    const int h_max = 30 + lg + NVDR_TREELET_CAP;      // (+ the levels a rebuilt treelet may add below a Karras node, bvh.hip)
    return h_max < NVDR_STACK_MAX ? h_max : NVDR_STACK_MAX;
}

// optional caller-provided device allocator (nvdr_ctx_set_allocator): the Python shim hands over torch's caching allocator, so the
// context's scratch (ray stream, stack spill, BVH buffers) shows up in torch's memory accounting and is recycled by it
typedef void *(*nvdr_alloc_fn)(size_t bytes, int device, void *stream, void *user);
typedef void (*nvdr_free_fn)(void *ptr, void *user);

struct nvdr_ctx {
    int device = 0;
    nvdr_alloc_fn alloc_fn = nullptr;
    nvdr_free_fn free_fn = nullptr;
    void *alloc_user = nullptr;
    int64_t n_allocs = 0;          // live allocations made through ctx_malloc
    int n_cus = 256;
    int64_t cap_tris = 0;
    int64_t n_tris = 0;
    int64_t n_verts = 0;
    uint4 *nodes = nullptr;        // [2 * cap]
    uint4 *oct = nullptr;          // [4 * cap] eight-wide nodes collapsed from nodes[] (bvh_oct_emit_kernel)
    uint4 *oct_top = nullptr;      // [4 * NVDR_TRACE_TOP_MAX] the treetop table (inside the oct allocation, behind its cap nodes)
    int trace_top = 0;             // entries of the table the build fills and the shadow-ray kernel keeps in LDS (0: none, the default; NVDR_TRACE_TOP_NODES)
    float4 *tris8 = nullptr;       // [3 * cap] triangle records in oct-leaf order
    int *oct_task = nullptr;       // [cap] the wide roots: binary nodes that root an eight-wide node (bvh_oct_budget_kernel)
    unsigned long long *oct_jump = nullptr;   // [cap] (ancestor, budget map) per binary node: the budget resolution's pointer-jumping state
    unsigned *oct_wslot = nullptr; // [cap] per wide root: (parent wide root << 3) | position among its internal slots
    unsigned long long *oct_cnt = nullptr, *oct_scan = nullptr;   // [cap] (internal, leaf) slot counts per binary node and their exclusive prefix sums
    unsigned *oct_ctl = nullptr;   // build counters, one 128-byte line each: wide roots, oct nodes, nodes written, triangles placed
    float4 *tris = nullptr;        // [3 * cap]
    uint32_t *keys[2] = {nullptr, nullptr};
    uint32_t *vals[2] = {nullptr, nullptr};
    uint2 *up = nullptr;           // [2T]: .x = (parent << 2) | (side << 1) | local of the internal nodes [0,T-1), then of the leaves [T, 2T); .y = collapse-DP record of an internal node
    int *flags = nullptr;          // [T] arrival counters of the bottom-up pass
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    BvhDeviceInfo *dinfo = nullptr;
    float *bounds_part = nullptr;  // [BVH_BOUNDS_BLOCKS][6] per-workgroup partial AABBs of bvh_bounds_kernel, then its ticket word
    unsigned *bounds_ticket = nullptr;
    int *spill = nullptr;          // [NVDR_QUERY_MAX_BLOCKS][stack_max - NVDR_STACK_LDS][NVDR_QUERY_BLOCK]
    int stack_max = 0;             // entries per lane the current spill allocation supports (LDS part included)
    int oct_stack_max = 0;         // the same for the oct walk's (group, bits) entries
    int spill_cap = 0;             // stack_max the spill buffer was allocated for (grow-only)
    int *ovf_host = nullptr;       // host-mapped overflow flag (a push beyond stack_max sets it)
    int *ovf_dev = nullptr;        // its device address
    unsigned debug = 0;            // NVDR_DEBUG, read ONCE when the context is created
    bool per_cu_user = false;      // NVDR_PBLOCKS was given: no launch-size rule on top of it
    int per_cu[3] = {10, 6, 6};     // workgroups per CU of the sample-generation, forward- and backward-shading kernels (NVDR_PBLOCKS="g,f,b")
    // BVH builds run on the context's own side stream, overlapped with whatever the caller enqueues next that does not need the
    // tree (pixel compaction and sample generation of env-shade: ~0.35-2.3 ms against a 0.25 ms build); consumers wait on `ev_built`
    // (ctx_wait_built).  The reference builds on stream 0 while everything else runs on torch's stream (torch_bindings.cpp:99).
    bool async_build = true;       // NVDR_ASYNC_BUILD=0: build on the caller's stream
    hipStream_t build_stream = nullptr;
    hipEvent_t ev_inputs = nullptr, ev_built = nullptr;
    float *in_verts = nullptr;     // the build's own copy of the caller's geometry (taken on the caller's stream: the caller may
    int32_t *in_tris = nullptr;    // overwrite or free its tensors as soon as nvdr_bvh_build has returned)
    int64_t in_verts_cap = 0, in_tris_cap = 0;
    bool build_deferred = false;   // nvdr_ctx_set_build_mode(2): the build's launches wait for the first consumer of the tree
    bool queued = false;           // a prepared build whose kernels have not been launched yet (ctx_launch_build)
    const float *queued_verts = nullptr; const int32_t *queued_tris = nullptr;
    int queued_rebuild = 0; int64_t queued_n_verts = 0, queued_n_tris = 0;
    bool built_pending = false;    // a build is (possibly) still in flight on build_stream
    hipStream_t built_waited = nullptr;   // the caller stream that already waits on ev_built ...
    bool built_waited_valid = false;      // ... if any (the default stream's handle IS the null pointer)
    // env-shade scratch
    int *pix_list = nullptr;       // [N*H*W] compacted indices of the covered pixels of the whole launch
    int64_t pix_cap = 0;
    unsigned *chunk_counts = nullptr; // [NVDR_MAX_CHUNKS][544] per chunk of the ray stream, one counter per 128-byte line: the lengths of the live list's segments, their capacity (trace_kernel.h)
    int64_t stream_cap_pixels = 0; // pixels one chunk of the ray stream holds (x 2S rays)
    int64_t stream_budget = 8192ll << 20;   // bytes the ray stream may take (nvdr_ctx_set_stream_budget); 2.8 % of the HBM of an MI355X
    // ray stream of the three-stage env-shade (csrc/env_shade.hip)
    float4 *rays = nullptr;
    int *texel = nullptr;
    uint8_t *vis = nullptr;
    uint32_t *live = nullptr;      // stream slots of the rays that need traversal (dead samples left out)
    unsigned *queues = nullptr;    // [256][32] chunk counters of the traversal kernel, one 128-B line each (NVDR_TRACE_QUEUES)
    float4 *pix_origin = nullptr;
    float4 *pix_grad = nullptr;    // [2 * stream_cap_pixels] backward: the incoming (diffuse, specular) gradients of a compacted pixel, packed in front of stage 3
    float4 *pix_setup = nullptr;   // [4 * stream_cap_pixels] per compacted pixel: (pos, nrm.x) (nrm.yz, view_pos.xy) (view_pos.z, kd) (ks, -), written by stage 1
    size_t stream_cap_rays = 0;    // slots of texel / vis
    size_t stream_cap_live = 0;    // entries of live (its segments together, trace_kernel.h)
    size_t stream_cap_total = 0;   // slots of rays: the chunk's own + the spare blocks of the light-gradient records
    uint16_t *lg_tags = nullptr;   // (band, fill) of every block of 128 slots of `rays` (0xFFFF: no records)
    size_t lg_tags_cap = 0;
    uint16_t *cdf_guide = nullptr; // guide tables of the light's CDF inversion (env_shade.hip), rebuilt per launch
    size_t guide_cap = 0;
    float *dp_cost = nullptr;      // [2 * cap][8] hand-off records of bvh_fit_kernel: collapse-DP table (7) + height of the two children of every binary node
    float oct_c_leaf = 0.45f;      // cost of a triangle test relative to a node step in the collapse DP
    int shade_queue = 3;           // shading kernels that queue the live light samples (NVDR_SHADE_QUEUE): S == 64 across pixels, bit 0 backward, bit 1 forward; S > 64 inside the pixel, bit 0 backward, bit 2 forward
    int trace_split_mode = 2;      // which traversal build a launch starts: 0 plain, 1 split walks in the drain, 2 by the last launch's rays per wavefront (NVDR_TRACE_SPLIT_MODE)
    bool refit_keep_collapse = true;   // a refit keeps the eight-wide collapse of the last full build (NVDR_REFIT_KEEP_COLLAPSE=0: every refit runs the dynamic programme again)
    bool collapse_valid = false;       // the collapse data of the current topology (slot splits, wide roots, prefix sums) is in the context
    int lg_f64 = -1;               // gather with fp64 accumulators + ds_add_f64: -1 for the launches whose workgroups are dealt to the bands (small ones), 0 never, 1 wherever the probe then fits 16 bands (NVDR_LG_F64)
    int lg_mode = -1;              // gather work split: -1 by launch size, 0 all bands per workgroup, 1 one set of workgroups per band (NVDR_LG_MODE)
    bool lg_tags_dirty = true;     // the array may hold tags nobody consumed (fresh allocation, a backward pass without gather)
    uint64_t stream_id = 0;        // id of the ray stream currently held in rays/texel/pix_origin/pix_list
    uint64_t stream_seq = 0;
    float *lg_part = nullptr;      // light-gradient partials: [chunks][Hl*Wl*3] of the band gather, or 8 per-XCD copies
    size_t lg_cap = 0;
    // optional per-stage timing ring (nvdr_ctx_set_profiling)
    bool profiling = false;
    hipEvent_t prof_ev[NVDR_PROF_RING][4] = {};   // one record per (launch, chunk of the ray stream)
    int prof_kind[NVDR_PROF_RING] = {};           // bit 0: backward, bit 1: first chunk of its launch
    int64_t prof_n = 0;
};

struct BvhView {
    const uint4 *nodes;
    const uint4 *oct;
    const uint4 *oct_top;
    const float4 *tris8;
    int oct_stack_max;
    const float4 *tris;
    const BvhDeviceInfo *info;
    int n_tris;
    int stack_max;      // entries per lane (LDS + spill) the context's spill buffer holds
    int *overflow;      // host-mapped flag, set by a push beyond stack_max
};

int ctx_check_overflow(nvdr_ctx *c, const char *who);   // bvh.hip
int ctx_launch_build(nvdr_ctx *c, hipStream_t stream);  // bvh.hip: the kernels of a prepared (possibly deferred) build
int ctx_wait_built(nvdr_ctx *c, hipStream_t stream);    // bvh.hip: make `stream` wait for the context's last BVH build

// every device buffer the context owns beyond its few fixed control words goes through these two
static inline hipError_t ctx_malloc_raw(nvdr_ctx *c, void **p, size_t bytes, hipStream_t stream)
{
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (c->alloc_fn) {
        *p = c->alloc_fn(bytes, c->device, (void *)stream, c->alloc_user);
        if (!*p) return hipErrorOutOfMemory;
    } else {
        const hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) return e;
    }
    c->n_allocs++;
    return hipSuccess;
}
template <typename T>
static inline hipError_t ctx_malloc(nvdr_ctx *c, T **p, size_t bytes, hipStream_t stream = nullptr)
{
    return ctx_malloc_raw(c, (void **)p, bytes, stream);
}
template <typename T>
static inline void ctx_free(nvdr_ctx *c, T *&p)
{
    if (!p) return;
    if (c->free_fn) c->free_fn((void *)p, c->alloc_user);
    else (void)hipFree((void *)p);
    c->n_allocs--;
    p = nullptr;
}

static inline BvhView bvh_view(const nvdr_ctx *c)
{
    BvhView v;
    v.nodes = c->nodes;
    v.oct = c->oct;
    v.oct_top = c->oct_top;
    v.tris8 = c->tris8;
    v.oct_stack_max = c->oct_stack_max;
    v.tris = c->tris;
    v.info = c->dinfo;
    v.n_tris = (int)c->n_tris;
    v.stack_max = c->stack_max;
    v.overflow = c->ovf_dev;
    return v;
}

// ---------------------------------------------------------------------------------------------
// device side

// explicit address spaces: keeps the compiler from emitting flat (generic) accesses with aperture checks
typedef __attribute__((address_space(3))) int lds_int_t;
typedef __attribute__((address_space(1))) int glb_int_t;

struct TravStack {
    lds_int_t *lds; // this lane's LDS column (stride 64 words)
    glb_int_t *glb; // this lane's HBM spill column
    int gstride;    // spill stride between two levels of a lane = threads of the workgroup
    int smax;       // entries this lane may hold (LDS + spill); see the note at NVDR_STACK_MAX
    int *ovf;       // host-mapped overflow flag of the context
    // returns the new depth.  Beyond the bound the entry is NOT stored and the depth does not grow (that subtree is skipped:
    // the walk stays finite -- an earlier version kept counting and re-popped one entry over and over, which hung the GPU in
    // the overflow test) and the context's flag is raised: never silently, the NEXT call on the context and nvdr_ctx_check report it
    __device__ __forceinline__ int push(int sp, int v) const
    {
        if (sp < NVDR_STACK_LDS) { lds[sp * 64] = v; return sp + 1; }
        if (sp < smax) { glb[(int64_t)(sp - NVDR_STACK_LDS) * gstride] = v; return sp + 1; }
        atomicOr(ovf, 1);
        return sp;
    }
    // peek returns what a pop at depth sp would yield without changing anything (branch-free loops read it every step)
    __device__ __forceinline__ int peek(int sp) const
    {
        int v = lds[max(min(sp - 1, NVDR_STACK_LDS - 1), 0) * 64];
        if (sp - 1 >= NVDR_STACK_LDS) v = glb[(int64_t)(min(sp - 1, smax - 1) - NVDR_STACK_LDS) * gstride];
        return v;
    }
    __device__ __forceinline__ int pop(int sp) const
    {
        return sp < NVDR_STACK_LDS ? lds[sp * 64]
                                   : glb[(int64_t)(min(sp, smax - 1) - NVDR_STACK_LDS) * gstride];
    }
};

// the lane's stack view inside a kernel: `smem` is the dynamic LDS base (blockDim.x * NVDR_STACK_LDS ints)
__device__ __forceinline__ TravStack make_stack(int *smem, int *spill, int stack_max, int *overflow)
{
    TravStack s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s.lds = (lds_int_t *)smem + wave * NVDR_STACK_LDS * 64 + lane;
    // Every workgroup keeps its spill columns together (level k of lane l at block_base + k * blockDim + l): a lane's levels
    // are 1 KB apart (the first layout strided them by the whole launch: 2 MB, one large page per level).
    s.gstride = blockDim.x;
    s.smax = stack_max;
    s.ovf = overflow;
    s.glb = (glb_int_t *)spill + (int64_t)blockIdx.x * blockDim.x * max(stack_max - NVDR_STACK_LDS, 0) + threadIdx.x;
    return s;
}
#define NVDR_STACK_LDS_BYTES(threads) ((size_t)(threads) * NVDR_STACK_LDS * sizeof(int))

__device__ __forceinline__ bool tri_any_hit(const float4 *__restrict__ tris, int slot, float ox, float oy, float oz,
                                            float dx, float dy, float dz)
{
    const float4 a = tris[3 * slot + 0], b = tris[3 * slot + 1], c = tris[3 * slot + 2];
    float t, u, v, det;
    return nvdr_ray_tri(ox, oy, oz, dx, dy, dz, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, &t, &u, &v, &det) != 0;
}

// A ray in grid space (for the box tests only): t(b) = b * inv + noi with noi = -o * inv, one fused op per plane.
struct GridRay {
    float nx, ny, nz, ix, iy, iz;
    unsigned px, py, pz;    // v_perm_b32 selectors that order a slot's (lo, hi) pair of an axis as (near, far) for this ray
};
__device__ __forceinline__ GridRay make_grid_ray(const BvhDeviceInfo *__restrict__ info, float ox, float oy, float oz,
                                                 float dx, float dy, float dz)
{
    GridRay g;
    const float sx = info->g_scale[0], sy = info->g_scale[1], sz = info->g_scale[2];
    // An axis the ray does not move along has 1 / (d * s) = +-inf, and in the fused form below b * inf + (-o * inf) is NaN for EVERY
    // plane (grid coordinates are positive), so fminf / fmaxf would drop that axis from every box test: correct, but an exactly
    // axis-parallel ray -- two such axes; a light sample at the pole of the probe is (0, 1, -0) once in ~1e8 samples -- would cull
    // along ONE axis only and visit a large part of the tree: hundreds of thousands of steps for one lane on a 684 k-triangle
    // mesh, 0.3-0.4 s during which the whole launch waits for it (the "slow mode" / "stalls" of rounds 1-2,
    // profiles/r02_slow_mode.md).  With |inv| capped at 1e30 the planes of such an axis evaluate to (b - o) * 1e30, i.e. to
    // -huge / +huge when the origin lies between them (no constraint, as it must be) and to +-huge on the same side when it does
    // not (culled): still conservative -- o is at least 0.99 cells inside the padded box of any triangle the exact predicate can
    // hit, the rounding of b * 1e30 - o * 1e30 is worth 0.004 cells, and a ray with |inv| >= 1e30 moves < 1e-14 cells over the
    // whole parameter range [0, 1e16] anyway.
    // (NVDR_INV_CAP=0 builds the uncapped form again: tools/axis_ray_probe.py compares the two)
#ifndef NVDR_INV_CAP
#define NVDR_INV_CAP 1
#endif
    const float INV_MAX = NVDR_INV_CAP ? 1.0e30f : __builtin_inff();
    g.ix = fminf(fmaxf(1.0f / (dx * sx), -INV_MAX), INV_MAX);
    g.iy = fminf(fmaxf(1.0f / (dy * sy), -INV_MAX), INV_MAX);
    g.iz = fminf(fmaxf(1.0f / (dz * sz), -INV_MAX), INV_MAX);
    g.nx = -((ox - info->g_lo[0]) * sx + 2.0f) * g.ix;
    g.ny = -((oy - info->g_lo[1]) * sy + 2.0f) * g.iy;
    g.nz = -((oz - info->g_lo[2]) * sz + 2.0f) * g.iz;
    // slot words: x = lo.x | lo.y << 16, y = lo.z | hi.x << 16, z = hi.y | hi.z << 16.  v_perm_b32(S0, S1, sel) picks each
    // result byte from S1 (selector 0-3) or S0 (4-7):  X = perm(x, y): lo.x = bytes 4,5 / hi.x = bytes 2,3;
    // Y = perm(x, z): lo.y = 6,7 / hi.y = 0,1;  Z = perm(y, z): lo.z = 4,5 / hi.z = 2,3.  Result = near | far << 16.
    g.px = g.ix >= 0.0f ? 0x03020504u : 0x05040302u;
    g.py = g.iy >= 0.0f ? 0x01000706u : 0x07060100u;
    g.pz = g.iz >= 0.0f ? 0x03020504u : 0x05040302u;
    return g;
}

// Slab test of one child box against the ray interval [0, tmax].  The traversal is VALU-issue bound, so every plane
// distance is ONE fma (pairs of them pack into v_pk_fma_f32).  An axis-parallel ray gives inv = +-inf and NaN plane
// distances; fminf/fmaxf drop NaNs, i.e. that axis simply stops culling: conservative (make_grid_ray caps |inv| so that this
// only happens for NaN inputs).  The fma rounds differently
// from (b - o) * inv by < 0.01 grid cells, far inside the one-cell slack every box carries.
__device__ __forceinline__ bool box_hit(float minx, float miny, float minz, float maxx, float maxy, float maxz,
                                        const GridRay &r, float tmax, float &tnear)
{
    const float x0 = fmaf(minx, r.ix, r.nx), x1 = fmaf(maxx, r.ix, r.nx);
    const float y0 = fmaf(miny, r.iy, r.ny), y1 = fmaf(maxy, r.iy, r.ny);
    const float z0 = fmaf(minz, r.iz, r.nz), z1 = fmaf(maxz, r.iz, r.nz);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
    tnear = tn;
    return tn <= tf;
}

__device__ __forceinline__ float lo16(unsigned w) { return (float)(w & 0xffffu); }
__device__ __forceinline__ float hi16(unsigned w) { return (float)(w >> 16); }

// one traversal step's worth of node data: fetch (two 16-B lookups), decode, test both children
struct NodeHit {
    bool hl, hr;
    float tl, tr;
    int cl, cr;
};
__device__ __forceinline__ NodeHit visit_node(const uint4 *__restrict__ nodes, int cur, const GridRay &r, float tmax)
{
    const uint4 a = nodes[2 * cur + 0], b = nodes[2 * cur + 1];
    NodeHit h;
    h.hl = box_hit(lo16(a.x), hi16(a.x), lo16(a.y), hi16(a.y), lo16(a.z), hi16(a.z), r, tmax, h.tl);
    h.hr = box_hit(lo16(a.w), hi16(a.w), lo16(b.x), hi16(b.x), lo16(b.y), hi16(b.y), r, tmax, h.tr);
    h.cl = (int)b.z;
    h.cr = (int)b.w;
    return h;
}

// ---------------------------------------------------------------------------------------------
// Any-hit traversal of up to TWO rays per lane that share an origin (a light-sampled and a BSDF-sampled
// shadow ray of one stratum; `todo` bit r = ray r exists).  Returns a mask: bit r = ray r is OCCLUDED.
// One loop for both rays: a lane whose first ray ends restarts at the root with its second ray instead of
// idling until the slowest lane of the wave is done.  The kernel is VALU-issue bound (rocprofv3: VALU busy
// ~90 % at ~40 % active lanes), so the loop body is kept short: node step and triangle step are the two arms
// of one branch.  (A "while-while" variant that parks leaves and runs the triangle test once per batch of
// node steps was measured 1.7x SLOWER: the extra divergence of the nested loops costs more than it saves.)
template <bool COUNT>
__device__ __forceinline__ unsigned bvh_any_hit2(const BvhView &bvh, float ox, float oy, float oz, float ax, float ay,
                                                 float az, float bx, float by, float bz, unsigned todo,
                                                 const TravStack &stack, unsigned &n_box, unsigned &n_tri)
{
    unsigned occluded = 0;
    if (bvh.n_tris == 1) {
        if (COUNT) n_tri += (todo & 1u) + ((todo >> 1) & 1u);
        if ((todo & 1u) && tri_any_hit(bvh.tris, 0, ox, oy, oz, ax, ay, az)) occluded |= 1u;
        if ((todo & 2u) && tri_any_hit(bvh.tris, 0, ox, oy, oz, bx, by, bz)) occluded |= 2u;
        return occluded;
    }
    int ray = (todo & 1u) ? 0 : ((todo & 2u) ? 1 : 2);
    float dx = ray == 0 ? ax : bx, dy = ray == 0 ? ay : by, dz = ray == 0 ? az : bz;
    GridRay g = make_grid_ray(bvh.info, ox, oy, oz, dx, dy, dz);
    int sp = 0, cur = 0;
    // `cur` >= 0: internal node, < 0: a leaf (~cur = triangle slot), NVDR_TRAV_DONE: nothing left.  Leaves travel
    // through the same variable / stack as nodes, so the loop body holds ONE copy of the triangle test.
    while (ray < 2) {
        bool finished = false;
        if (cur >= 0) {
            const NodeHit h = visit_node(bvh.nodes, cur, g, NVDR_RAY_TMAX);
            if (COUNT) n_box += 2;
            if (h.hl && h.hr) {
                const bool left_first = h.tl <= h.tr;
                sp = stack.push(sp, left_first ? h.cr : h.cl);
                cur = left_first ? h.cl : h.cr;
            } else if (h.hl) {
                cur = h.cl;
            } else if (h.hr) {
                cur = h.cr;
            } else if (sp > 0) {
                sp--;
                cur = stack.pop(sp);
            } else {
                finished = true;
            }
        } else {
            if (COUNT) n_tri++;
            if (tri_any_hit(bvh.tris, ~cur, ox, oy, oz, dx, dy, dz)) {
                occluded |= 1u << ray;
                finished = true;
            } else if (sp > 0) {
                sp--;
                cur = stack.pop(sp);
            } else {
                finished = true;
            }
        }
        if (finished) {
            ray = (ray == 0 && (todo & 2u)) ? 1 : 2;
            dx = bx; dy = by; dz = bz;
            g = make_grid_ray(bvh.info, ox, oy, oz, dx, dy, dz);
            sp = 0;
            cur = 0;
        }
    }
    return occluded;
}

// Closest hit (ordered binary walk, shrinking tmax): returns the ORIGINAL index of the nearest triangle (-1 = miss) and its
// distance / barycentrics of v1, v2.  Used by nvdr_trace_closest and the G-buffer producer (gbuffer.hip).
__device__ __forceinline__ int bvh_closest_hit(const BvhView &bvh, float ox, float oy, float oz, float dx, float dy, float dz,
                                               const TravStack &stack, float &best_t, float &best_u, float &best_v)
{
    best_t = NVDR_RAY_TMAX;
    best_u = 0.0f;
    best_v = 0.0f;
    int best = -1;
    auto test_leaf = [&](int slot) {
        const float4 a = bvh.tris[3 * slot + 0], b = bvh.tris[3 * slot + 1], c = bvh.tris[3 * slot + 2];
        float t, u, v, det;
        if (nvdr_ray_tri(ox, oy, oz, dx, dy, dz, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, &t, &u, &v, &det)) {
            const float tt = t / det;
            if (tt < best_t) {
                best_t = tt;
                best_u = u / det;
                best_v = v / det;
                best = __float_as_int(c.y);
            }
        }
    };
    if (bvh.n_tris == 1) {
        test_leaf(0);
        return best;
    }
    const GridRay g = make_grid_ray(bvh.info, ox, oy, oz, dx, dy, dz);
    int sp = 0, cur = 0;
    while (true) {
        NodeHit h = visit_node(bvh.nodes, cur, g, best_t);
        if (h.hl && h.cl < 0) { test_leaf(~h.cl); h.hl = false; }
        if (h.hr && h.cr < 0) { test_leaf(~h.cr); h.hr = false; }
        if (h.hl && h.hr) {
            const bool left_first = h.tl <= h.tr;
            sp = stack.push(sp, left_first ? h.cr : h.cl);
            cur = left_first ? h.cl : h.cr;
        } else if (h.hl) {
            cur = h.cl;
        } else if (h.hr) {
            cur = h.cr;
        } else {
            if (sp == 0) break;
            sp--;
            cur = stack.pop(sp);
        }
    }
    return best;
}

// one ray per lane
template <bool COUNT>
__device__ __forceinline__ bool bvh_any_hit(const BvhView &bvh, float ox, float oy, float oz, float dx, float dy,
                                            float dz, const TravStack &stack, unsigned &n_box, unsigned &n_tri)
{
    return bvh_any_hit2<COUNT>(bvh, ox, oy, oz, dx, dy, dz, dx, dy, dz, 1u, stack, n_box, n_tri) != 0;
}
