// bvh.hip -- LBVH construction on device + ray-query test hooks.
//
// Replaces optix_build_bvh (render/optixutils/c_src/torch_bindings.cpp:37-116): the reference hands
// verts/tris to the closed-source optixAccelBuild every training iteration (geometry/dlmesh.py:50,
// geometry/dmtet.py:202), with a cudaMalloc/cudaFree pair per call.  Here:
//   1. vertex AABB            (one pass, wave-reduced atomics on order-preserving ints)
//   2. 30-bit Morton codes of triangle centroids
//   3. device radix sort (rocPRIM, 30 key bits) of (code, triangle) pairs
//   4. Karras-2012 hierarchy, one thread per internal node
//   5. bottom-up bounds: one thread per leaf climbs, the second arriver at a node continues;
//      the child boxes are written INTO the parent's 32-B record as 16-bit grid coordinates (layout:
//      bvh.h); unions are taken on the integers, so quantisation slack does not grow up the tree
// Buffers live in the context and are reused across iterations; rebuild == 0 re-runs only 1 + 5
// (OPTIX_BUILD_OPERATION_UPDATE).  Nothing synchronises the host.
#include "bvh.h"

#include <rocprim/rocprim.hpp>

// rocPRIM's radix_sort_pairs runs a merge sort below 1 M items (one block sort + 2 launches per doubling: 21 launches and 0.27 ms for the
// 684 k keys of the large benchmark mesh).  Its onesweep radix sort (histogram + one launch per 8 key bits, ~0.1 ms alone) was
// measured in round 4 and is NOT used: its workgroups spin on their predecessors (decoupled look-back), and beside the persistent
// sample-generation kernel that fills the GPU while the build runs on its side stream one pass in four took 0.5 ms.  The merge
// sort has no dependency between the workgroups of a launch.
using nvdr_sort_config = rocprim::default_config;

// ---------------------------------------------------------------------------------------------
// kernels

__device__ __forceinline__ int float_to_ordered(float f)
{
    int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __host__ __forceinline__ float ordered_to_float(int i)
{
    i = i ^ ((i >> 31) & 0x7fffffff);
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(i);
#else
    float f;
    memcpy(&f, &i, 4);
    return f;
#endif
}

// Vertex AABB + quantisation grid + leaf padding in ONE launch: every workgroup reduces its share to six floats in `part`, takes a
// ticket, and the workgroup that draws the last one reduces the partials and writes the grid (bounds as order-preserving ints, as
// the Morton kernel reads them).  Rounds 1-3: three launches and six atomicMin / atomicMax per WAVEFRONT on the same six words --
// 24 k serialised L2 atomics = 0.28 ms for 342 k vertices, during which anything else on the GPU crawled (a 15 us elementwise kernel on
// the caller's stream took 0.29 ms beside it).
#define BVH_BOUNDS_BLOCKS 128
__global__ void __launch_bounds__(256) bvh_bounds_kernel(const float *__restrict__ verts, int64_t n_verts, BvhDeviceInfo *info, float *part, unsigned *ticket)
{
    __shared__ float red[6][4];
    __shared__ bool last;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_verts; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = verts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto reduce = [&]() {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            for (int o = 32; o >= 1; o >>= 1) {
                mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
                mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
            }
            if (lane == 0) { red[a][wave] = mn[a]; red[3 + a][wave] = mx[a]; }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(fminf(red[a][0], red[a][1]), fminf(red[a][2], red[a][3]));
            mx[a] = fmaxf(fmaxf(red[3 + a][0], red[3 + a][1]), fmaxf(red[3 + a][2], red[3 + a][3]));
        }
    };
    reduce();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {   // write-through stores (the hand-off recipe of bvh_fit_kernel: no cache to write back or drop)
            __hip_atomic_store(&part[6 * blockIdx.x + a], mn[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&part[6 * blockIdx.x + 3 + a], mx[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the partials have arrived before the ticket moves
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], __hip_atomic_load(&part[6 * b + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            mx[a] = fmaxf(mx[a], __hip_atomic_load(&part[6 * b + 3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    __syncthreads();
    reduce();
    if (threadIdx.x == 0) {
        *ticket = 0u;                                                       // ready for the next build
        float scale = 0.0f;
        for (int a = 0; a < 3; ++a) {
            info->bounds[a] = float_to_ordered(mn[a]);
            info->bounds[3 + a] = float_to_ordered(mx[a]);
            scale = fmaxf(scale, fmaxf(mx[a] - mn[a], fmaxf(fabsf(mn[a]), fabsf(mx[a]))));
        }
        info->height = 0;
        info->root = 0;
        // conservative padding: the slab test must never cull a triangle the fp32 Moeller-Trumbore
        // predicate would accept (its acceptance band is a few ulp of the scene scale wide)
        const float pad = 1e-5f * scale;
        info->pad = pad;
        for (int a = 0; a < 3; ++a) {
            const float g0 = mn[a] - 2.0f * pad, g1 = mx[a] + 2.0f * pad;
            const float ext = fmaxf(g1 - g0, 1e-6f * scale + 1e-30f);
            info->g_lo[a] = g0;
            info->g_scale[a] = NVDR_GRID_MAX / ext;
        }
    }
}

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void bvh_morton_kernel(const float *__restrict__ verts, const int32_t *__restrict__ tris, int n_tris,
                                  const BvhDeviceInfo *__restrict__ info, uint32_t *__restrict__ keys,
                                  uint32_t *__restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    const int i0 = tris[3 * i + 0], i1 = tris[3 * i + 1], i2 = tris[3 * i + 2];
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = ordered_to_float(info->bounds[a]), hi = ordered_to_float(info->bounds[3 + a]);
        const float c = (verts[3 * i0 + a] + verts[3 * i1 + a] + verts[3 * i2 + a]) * (1.0f / 3.0f);
        const float ext = hi - lo;
        float f = ext > 0.0f ? (c - lo) / ext : 0.5f;
        f = fminf(fmaxf(f * 1024.0f, 0.0f), 1023.0f);
        code |= expand_bits10((uint32_t)f) << (2 - a);
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// common-prefix length of sorted keys i and j, ties broken by position (Karras 2012, section 4)
__device__ __forceinline__ int lbvh_delta(const uint32_t *__restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    return a == b ? 32 + __clz(i ^ j) : __clz(a ^ b);
}

// Besides the child links, ONE word per child says everything a climb needs: up[c] = (parent << 2) | (side << 1) | local, c an
// internal node [0, n - 1) or leaf k at [n + k]; side = 0 left / 1 right; local = the parent's leaf range lies inside one block of
// NVDR_FIT_BLOCK consecutive leaves, i.e. both of its subtrees are climbed by threads of the same workgroup of bvh_fit_kernel, which
// then meet in LDS.  (The eight-wide builder's budget walk reads the same words.)
#define NVDR_FIT_BLOCK 256
// control words of the eight-wide collapse (below), one 128-byte line each
#define OCT_CTL_ROOTS 0         // wide roots found (step 1)
#define OCT_CTL_ALLOC 32        // oct nodes (written by the emit step: 1 + sum of the internal slots)
#define OCT_CTL_DONE 64         // oct nodes written
#define OCT_CTL_TRIS 96         // triangles placed
#define OCT_CTL_TREELETS 128    // treelet roots listed (bvh_treelet_mark_kernel)
#define OCT_CTL_WORDS 160
// (Also clears what the two stages behind it start from -- the arrival counters of the fit, the control words of the eight-wide collapse:
// two 5-us launches less per rebuild; a refit, which does not come through here, keeps them.)
__global__ void bvh_hierarchy_kernel(const uint32_t *__restrict__ keys, int n, uint4 *__restrict__ nodes, uint2 *__restrict__ up, int *__restrict__ flags,
                                     unsigned *__restrict__ oct_ctl, uint2 *__restrict__ range)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    flags[i] = 0;
    if (i == n - 2) flags[n - 1] = 0;
    if (i == 0) { oct_ctl[OCT_CTL_ROOTS] = 0u; oct_ctl[OCT_CTL_ALLOC] = 1u; oct_ctl[OCT_CTL_DONE] = 0u; oct_ctl[OCT_CTL_TRIS] = 0u; oct_ctl[OCT_CTL_TREELETS] = 0u; }
    const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    range[i] = make_uint2((unsigned)lo, (unsigned)hi);      // the node's leaves: positions lo .. hi of the Morton order (bvh_treelet_*_kernel)
    const int left = (lo == gamma) ? ~gamma : gamma;
    const int right = (hi == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
    unsigned *rec = (unsigned *)(nodes + 2 * i);
    rec[6] = (unsigned)left;
    rec[7] = (unsigned)right;
    const unsigned w = ((unsigned)i << 2) | ((lo / NVDR_FIT_BLOCK) == (hi / NVDR_FIT_BLOCK) ? 1u : 0u);
    up[left < 0 ? n + gamma : left].x = w;          // (.y of an internal node: its collapse-DP record, written by bvh_fit_kernel --
    up[right < 0 ? n + gamma + 1 : right].x = w | 2u;   //  one 8-byte load per level serves the budget walk of the eight-wide builder)
    if (i == 0) up[0].x = 0u;
}

// ---------------------------------------------------------------------------------------------
// TREELETS: the bottom of the Morton tree rebuilt by agglomerative clustering (round 5).
//
// A Karras tree splits a key range at its highest differing bit: fine at the top, poor at the bottom, where a handful of triangles
// that share a Morton cell are paired by the accidents of the Z curve.  On the CPU model (tools/treelet_model.py) rebuilding every
// maximal subtree of <= 64 leaves by greedy agglomerative clustering -- merge the two clusters whose union has the smallest surface
// area -- takes 6.0 % of the node steps of a shadow ray off the eight-wide walk on bob (a full binned-SAH build: 8.5 %, a SAH tree
// over the TOP of the Morton tree: 1-2 %: the loss is at the bottom).  It is local work: ONE WAVEFRONT per treelet, a lane per cluster,
//   * every round each live cluster looks for the partner with the smallest union (the candidates' boxes come as scalars out of
//     v_readlane: 6 reads + ~16 VALU per candidate), MUTUAL nearest neighbours merge (Walter et al. 2008 / PLOC, Meister & Bittner 2018:
//     the closest pair is always mutual, so every round merges at least one pair; ~8 rounds for 64 leaves); ties go to the lower lane,
//     so the tree is a function of the input alone;
//   * the new internal nodes take the ids the Karras subtree had -- a Karras subtree over the leaves lo .. hi owns the ids lo + 1 .. hi - 1
//     and its root id (lo or hi) -- so nothing outside the treelet changes, and "a node lies inside its own leaf range", which the
//     LDS hand-off of bvh_fit_kernel indexes by, still holds;
//   * treelets never straddle a block of NVDR_FIT_BLOCK leaves (a straddling subtree is split further), so every node in them is "local";
//   * a rebuilt treelet deeper than NVDR_TREELET_CAP levels is not committed (never seen: the deepest of bob's 380 has 8 levels), which
//     keeps the proven stack bound: h_max + NVDR_TREELET_CAP (bvh.h).
// Refits (rebuild = 0) keep whatever topology the last build left.
#ifndef NVDR_TREELET_W
#define NVDR_TREELET_W 64           // leaves per treelet of meshes up to NVDR_TREELET_LARGE triangles (0: plain Karras tree, A/B)
#endif
#ifndef NVDR_TREELET_W_LARGE
#define NVDR_TREELET_W_LARGE 32     // ... and of larger ones: two treelets per wavefront (the search is quadratic in the treelet: 227 -> ~50 us on
#endif                              // 684 k triangles alone on the GPU, where the build is on the critical path of a one-view iteration with trained
#define NVDR_TREELET_LARGE 65536    // geometry; CPU model: -5.0 % instead of -6.0 % node steps on bob)

__device__ __forceinline__ bool treelet_ok(uint2 r, unsigned w) { return r.y - r.x < w && (r.x / NVDR_FIT_BLOCK) == (r.y / NVDR_FIT_BLOCK); }

// list the maximal subtrees that fit a treelet (>= 3 leaves: two leaves have one tree)
__global__ void __launch_bounds__(1024) bvh_treelet_mark_kernel(const uint2 *__restrict__ range, const uint2 *__restrict__ up, int n_int, unsigned w,
                                                                int *__restrict__ list, unsigned *__restrict__ ctl)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    bool is_root = false;
    if (v < n_int) {
        const uint2 r = range[v];
        is_root = treelet_ok(r, w) && r.y - r.x >= 2u && (v == 0 || !treelet_ok(range[up[v].x >> 2], w));
    }
    __shared__ unsigned wave_cnt[16], wave_base[16];
    const unsigned long long m = __ballot(is_root);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) { wave_base[w2] = tot; tot += wave_cnt[w2]; }
        const unsigned base = tot ? atomicAdd(&ctl[OCT_CTL_TREELETS], tot) : 0u;
        for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) wave_base[w2] += base;
    }
    __syncthreads();
    if (is_root) list[wave_base[wave] + __popcll(m & ((1ull << lane) - 1ull))] = v;
}

// W leaves per treelet: 64 / W treelets per wavefront ("parts" of W lanes; everything below that says "own part" is per-lane state that is
// uniform inside a part).  The candidates' boxes come through ds_bpermute (the source lane differs between the parts).
template <int W>
__global__ void __launch_bounds__(256) bvh_treelet_kernel(const float *__restrict__ verts, const int32_t *__restrict__ tris, const uint32_t *__restrict__ order, int n,
                                                          const uint2 *__restrict__ range, uint4 *__restrict__ nodes, uint2 *__restrict__ up,
                                                          const int *__restrict__ list, const unsigned *__restrict__ ctl)
{
    constexpr int G = 64 / W;
    constexpr unsigned long long WMASK = W == 64 ? ~0ull : ((1ull << (W & 63)) - 1ull);
    __shared__ int m_id[4][64], m_l[4][64], m_r[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (W - 1), pbase = lane & ~(W - 1);
    const unsigned count = ctl[OCT_CTL_TREELETS];
    const unsigned waves_total = gridDim.x * (blockDim.x >> 6);
    for (unsigned e0 = (blockIdx.x * (blockDim.x >> 6) + wave) * G; e0 < count; e0 += waves_total * G) {
        const unsigned e = e0 + (unsigned)(lane / W);
        const bool has = e < count;
        const int root = has ? list[e] : 0;
        const uint2 rg = has ? range[root] : make_uint2(0u, 0u);
        const int lo = (int)rg.x, k = has ? (int)(rg.y - rg.x) + 1 : 0;
        bool active = sub < k;
        int ref = ~(lo + sub), depth = 0;
        float bx0 = 0.0f, by0 = 0.0f, bz0 = 0.0f, bx1 = 0.0f, by1 = 0.0f, bz1 = 0.0f;
        if (active) {
            const uint32_t t = order[lo + sub];
            const int i0 = tris[3 * t], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
            const float ax = verts[3 * i0], ay = verts[3 * i0 + 1], az = verts[3 * i0 + 2];
            const float bx = verts[3 * i1], by = verts[3 * i1 + 1], bz = verts[3 * i1 + 2];
            const float cx = verts[3 * i2], cy = verts[3 * i2 + 1], cz = verts[3 * i2 + 2];
            bx0 = fminf(ax, fminf(bx, cx)); by0 = fminf(ay, fminf(by, cy)); bz0 = fminf(az, fminf(bz, cz));
            bx1 = fmaxf(ax, fmaxf(bx, cx)); by1 = fmaxf(ay, fmaxf(by, cy)); bz1 = fmaxf(az, fmaxf(bz, cz));
        }
        int created = 0, n_act = k;
        while (__ballot(n_act > 1) != 0ull) {
            const unsigned long long m = __ballot(active);
            const unsigned long long mine = (m >> pbase) & WMASK;                   // the live clusters of the own part, by position
            unsigned long long visit = 0ull;                                        // positions some part has a live cluster at (wave-uniform)
#pragma unroll
            for (int g = 0; g < G; ++g) visit |= (m >> (g * W)) & WMASK;
            // nearest neighbour by the surface area of the union; candidates in rising position, strict <: ties go to the lower one
            float best = 3.0e38f;
            int best_j = sub;
            for (unsigned long long vv = visit; vv; vv &= vv - 1ull) {
                const int s2 = __builtin_ctzll(vv), src = pbase + s2;
                const float x0 = __shfl(bx0, src), y0 = __shfl(by0, src), z0 = __shfl(bz0, src);
                const float x1 = __shfl(bx1, src), y1 = __shfl(by1, src), z1 = __shfl(bz1, src);
                const float ex = fmaxf(bx1, x1) - fminf(bx0, x0), ey = fmaxf(by1, y1) - fminf(by0, y0), ez = fmaxf(bz1, z1) - fminf(bz0, z0);
                const float a = (ex * ey + ey * ez) + ez * ex;
                if (((mine >> s2) & 1ull) && s2 != sub && a < best) { best = a; best_j = s2; }
            }
            const bool seeking = active && n_act > 1;
            // the partner's choice and state (every lane takes part in the permutes; a lane without a partner reads itself)
            const int src = pbase + (seeking ? best_j : sub);
            const int their = __shfl(best_j, src);
            const bool mutual = seeking && best_j != sub && their == sub;
            const bool merger = mutual && sub < best_j;
            const int p_ref = __shfl(ref, src), p_depth = __shfl(depth, src);
            const float px0 = __shfl(bx0, src), py0 = __shfl(by0, src), pz0 = __shfl(bz0, src);
            const float px1 = __shfl(bx1, src), py1 = __shfl(by1, src), pz1 = __shfl(bz1, src);
            const unsigned long long mg = (__ballot(merger) >> pbase) & WMASK;      // the merging clusters of the own part
            const int merges = __popcll(mg);
            if (merger) {
                const int idx = created + __popcll(mg & ((1ull << sub) - 1ull));
                const int nid = n_act == 2 ? root : lo + 1 + idx;           // the last merge is the treelet's root; the others take the ids lo + 1 .. hi - 1
                m_id[wave][pbase + idx] = nid; m_l[wave][pbase + idx] = ref; m_r[wave][pbase + idx] = p_ref;
                ref = nid;
                depth = 1 + max(depth, p_depth);
                bx0 = fminf(bx0, px0); by0 = fminf(by0, py0); bz0 = fminf(bz0, pz0);
                bx1 = fmaxf(bx1, px1); by1 = fmaxf(by1, py1); bz1 = fmaxf(bz1, pz1);
            }
            if (mutual && sub > best_j) active = false;
            created += merges;
            n_act = (n_act > 1 && merges == 0) ? -1 : n_act - merges;      // (-1: cannot happen -- the closest pair is mutual; a NaN box lands here and keeps its Karras subtree)
        }
        const unsigned long long last = (__ballot(active) >> pbase) & WMASK;
        const int top_depth = __shfl(depth, pbase + (last ? __builtin_ctzll(last) : 0));
        __builtin_amdgcn_wave_barrier();
        if (n_act == 1 && k >= 3 && top_depth <= NVDR_TREELET_CAP && sub < k - 1) {
            const int nid = m_id[wave][pbase + sub], l = m_l[wave][pbase + sub], r = m_r[wave][pbase + sub];
            unsigned *rec = (unsigned *)(nodes + 2 * (int64_t)nid);
            rec[6] = (unsigned)l;
            rec[7] = (unsigned)r;
            up[l < 0 ? n + ~l : l].x = ((unsigned)nid << 2) | 1u;                // (every node of a treelet is local: treelet_ok)
            up[r < 0 ? n + ~r : r].x = ((unsigned)nid << 2) | 3u;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Write-through (sc1) accesses of the hand-off below: they reach / come from the level all XCDs agree on, so no cache has to be
// written back or invalidated around them.
typedef unsigned nvdr_u3 __attribute__((ext_vector_type(3)));
typedef float nvdr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1(unsigned *p, nvdr_u3 v) { asm volatile("global_store_dwordx3 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_sc1(float *p, nvdr_f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ nvdr_u3 load_sc1_u3(const unsigned *p)
{
    nvdr_u3 v;
    asm volatile("global_load_dwordx3 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ nvdr_f4 load_sc1_f4(const float *p)
{
    nvdr_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// budget a child on side `side` (0 left, 1 right) of a node with DP record `rec` receives when that node holds budget i (1..8)
__device__ __forceinline__ unsigned oct_child_budget(unsigned rec, unsigned i, int side)
{
    const unsigned kr = (rec >> 24) & 7u;
    const unsigned sp = (i >= 2u && i <= 7u) ? ((rec >> (3u * i)) & 7u) : 0u;     // 0: the node stays one slot = it roots a wide node
    const unsigned l = sp ? sp : kr, tot = sp ? i : 8u;
    return side == 0 ? l : tot - l;
}

// One thread per leaf: write the triangle record, then climb; the thread that arrives second at a node takes it on.  What the two
// subtrees of a node hand each other -- box, height, collapse-DP table: 44 bytes -- travels
//   * through LDS when both subtrees belong to this workgroup's 256 consecutive (Morton-sorted) leaves: the bottom ~8-20 levels of
//     every path, > 99 % of the nodes (round 4; a level costs an LDS round trip instead of three dependent trips to memory);
//   * through memory otherwise, as WRITE-THROUGH stores (cdna_hip_programming.md G16, the sc1 variant): sc1 stores -> drained vmcnt ->
//     relaxed agent-scope atomic on the node's counter; the second arriver reads them back with sc1 loads.  (Rounds 1-3 used plain stores
//     between an agent-scope release and acquire fence PER NODE: each release writes back the whole L2 of the XCD and each acquire
//     drops the CU's L1 -- 1.4 M of each on the 684 k-triangle mesh, 7-8 ms for this kernel, and the sample-generation kernel that
//     runs beside it on the other stream, whose stores those write-backs kept flushing, took 10.3 instead of 4 ms.)
// `xchg`: [2 * n][8] floats, the memory-side record of the (left, right) child of a node: cost[0..6] of the collapse DP, then the height.
// DP = false (round 6, the REFIT): boxes only.  The collapse into eight-wide nodes keeps the slot splits the last full build's dynamic programme chose (up[].y,
// the wide roots, the prefix sums: all still in the context) -- any collapse of the binary tree is a valid tree, the moved boxes are exact, and the periodic
// rebuild refreshes the choice -- so the DP tables, their 32-byte hand-off per level and the budget / scan kernels behind this one drop out of a refit.
template <bool DP>
__global__ void __launch_bounds__(NVDR_FIT_BLOCK) bvh_fit_kernel(const float *__restrict__ verts, const int32_t *__restrict__ tris,
                               const uint32_t *__restrict__ order, int n, float4 *__restrict__ tri_rec, uint4 *nodes,
                               uint2 *up, int *flags, BvhDeviceInfo *info,
                               float *xchg, float c_leaf, unsigned long long *jump)
{
    __shared__ unsigned l_box[NVDR_FIT_BLOCK][2][3];
    __shared__ float l_x[DP ? NVDR_FIT_BLOCK : 1][2][8];
    __shared__ int l_flag[NVDR_FIT_BLOCK];
    l_flag[threadIdx.x] = 0;
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t orig = order[k];
    const int i0 = tris[3 * orig + 0], i1 = tris[3 * orig + 1], i2 = tris[3 * orig + 2];
    const float ax = verts[3 * i0], ay = verts[3 * i0 + 1], az = verts[3 * i0 + 2];
    const float bx = verts[3 * i1], by = verts[3 * i1 + 1], bz = verts[3 * i1 + 2];
    const float cx = verts[3 * i2], cy = verts[3 * i2 + 1], cz = verts[3 * i2 + 2];
    const float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    const float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    tri_rec[3 * k + 0] = make_float4(ax, ay, az, e1x);
    tri_rec[3 * k + 1] = make_float4(e1y, e1z, e2x, e2y);
    tri_rec[3 * k + 2] = make_float4(e2z, __int_as_float((int)orig), 0.0f, 0.0f);
    if (n == 1) return;

    // leaf box: exact min/max of the vertices, padded, then snapped OUTWARD to the 16-bit grid with one
    // extra cell of slack (covers the rounding of the ray's own transform into grid space)
    const float pad = info->pad;
    const float mn[3] = {fminf(ax, fminf(bx, cx)) - pad, fminf(ay, fminf(by, cy)) - pad, fminf(az, fminf(bz, cz)) - pad};
    const float mx[3] = {fmaxf(ax, fmaxf(bx, cx)) + pad, fmaxf(ay, fmaxf(by, cy)) + pad, fmaxf(az, fmaxf(bz, cz)) + pad};
    int qmn[3], qmx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = (mn[a] - info->g_lo[a]) * info->g_scale[a] + 2.0f;
        const float hi = (mx[a] - info->g_lo[a]) * info->g_scale[a] + 2.0f;
        qmn[a] = min(max((int)floorf(lo) - 1, 0), 65535);
        qmx[a] = min(max((int)ceilf(hi) + 1, 0), 65535);
    }
    int height = 0;
    unsigned w = up[n + k].x;
    // The collapse into eight-wide nodes is chosen by dynamic programming over this same bottom-up pass (Ylitie, Karras, Laine 2017,
    // section 4.1): cost[i - 1] = cheapest representation of the subtree carried by this thread in AT MOST i slots of an ancestor's
    // wide node (i = 1..7), in units of (surface area x cost of one node step); a triangle costs c_leaf of a node step.
    //   c(v, 1) = A_v + min_k c(l, k) + c(r, 8 - k)            v becomes the root of a wide node; the argmin is kept as its root split
    //   c(v, i) = min(c(v, 1), min_k c(l, k) + c(r, i - k))    or its slots go to the children; the argmin (0 = stay one slot) is kept
    const float wxs = 1.0f / info->g_scale[0], wys = 1.0f / info->g_scale[1], wzs = 1.0f / info->g_scale[2];
    float cost[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (DP) {
        const float ex = (float)(qmx[0] - qmn[0]) * wxs, ey = (float)(qmx[1] - qmn[1]) * wys, ez = (float)(qmx[2] - qmn[2]) * wzs;
        const float leaf = (ex * ey + ey * ez + ez * ex) * c_leaf;
#pragma unroll
        for (int i = 0; i < 7; ++i) cost[i] = leaf;
    }
    while (true) {
        const int node = (int)(w >> 2), slot = (int)((w >> 1) & 1u);
        const bool local = (w & 1u) != 0u;
        const unsigned w_next = up[node].x;                                 // in flight while this level is worked on (up[0].x = 0: unused)
        unsigned *rec = (unsigned *)(nodes + 2 * (int64_t)node);
        nvdr_u3 box;
        box.x = (unsigned)qmn[0] | ((unsigned)qmn[1] << 16);
        box.y = (unsigned)qmn[2] | ((unsigned)qmx[0] << 16);
        box.z = (unsigned)qmx[1] | ((unsigned)qmx[2] << 16);
        unsigned s0, s1, s2;
        float sib[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int sib_height = 0;
        if (local) {
            // both subtrees are this workgroup's: the node record gets a plain store (nobody reads it in this kernel), the hand-off is LDS
            rec[3 * slot] = box.x; rec[3 * slot + 1] = box.y; rec[3 * slot + 2] = box.z;
            const int li = node & (NVDR_FIT_BLOCK - 1);                     // a Karras node lies inside its own leaf range
            l_box[li][slot][0] = box.x; l_box[li][slot][1] = box.y; l_box[li][slot][2] = box.z;
            if (DP) {
#pragma unroll
                for (int i = 0; i < 7; ++i) l_x[li][slot][i] = cost[i];
                l_x[li][slot][7] = __int_as_float(height);
            }
            // LDS-only ordering (s_waitcnt lgkmcnt): a workgroup-scope acq_rel atomic would also drain the global stores above, a trip to
            // memory per level -- exactly what this path is there to avoid
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            const int old = __hip_atomic_fetch_add(&l_flag[li], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            if (old == 0) return; // first arriver: the sibling subtree finishes this node
            s0 = l_box[li][1 - slot][0]; s1 = l_box[li][1 - slot][1]; s2 = l_box[li][1 - slot][2];
            if (DP) {
#pragma unroll
                for (int i = 0; i < 7; ++i) sib[i] = l_x[li][1 - slot][i];
                sib_height = __float_as_int(l_x[li][1 - slot][7]);
            }
        } else {
            store_sc1(rec + 3 * slot, box);
            float *xc = xchg + 8 * (2 * (int64_t)node + slot);
            if (DP) {
                nvdr_f4 x0, x1;
                x0.x = cost[0]; x0.y = cost[1]; x0.z = cost[2]; x0.w = cost[3];
                x1.x = cost[4]; x1.y = cost[5]; x1.z = cost[6]; x1.w = __int_as_float(height);
                store_sc1(xc, x0);
                store_sc1(xc + 4, x1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the write-through stores have arrived ...
            const int old = __hip_atomic_fetch_add(&flags[node], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ... before the counter moves
            if (old == 0) return; // first arriver: the sibling subtree finishes this node
            const nvdr_u3 sb = load_sc1_u3(rec + 3 * (1 - slot));
            s0 = sb.x; s1 = sb.y; s2 = sb.z;
            if (DP) {
                const float *sx = xchg + 8 * (2 * (int64_t)node + (1 - slot));
                const nvdr_f4 y0 = load_sc1_f4(sx), y1 = load_sc1_f4(sx + 4);
                sib[0] = y0.x; sib[1] = y0.y; sib[2] = y0.z; sib[3] = y0.w; sib[4] = y1.x; sib[5] = y1.y; sib[6] = y1.z;
                sib_height = __float_as_int(y1.w);
            }
        }
        qmn[0] = min(qmn[0], (int)(s0 & 0xffffu)); qmn[1] = min(qmn[1], (int)(s0 >> 16)); qmn[2] = min(qmn[2], (int)(s1 & 0xffffu));
        qmx[0] = max(qmx[0], (int)(s1 >> 16)); qmx[1] = max(qmx[1], (int)(s2 & 0xffffu)); qmx[2] = max(qmx[2], (int)(s2 >> 16));
        height = 1 + max(height, sib_height);
        if (DP) {
            // (left, right) tables in tree order, whichever of the two this thread carried
            float Lc[7], Rc[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) { Lc[i] = slot == 0 ? cost[i] : sib[i]; Rc[i] = slot == 0 ? sib[i] : cost[i]; }
            const float ex = (float)(qmx[0] - qmn[0]) * wxs, ey = (float)(qmx[1] - qmn[1]) * wys, ez = (float)(qmx[2] - qmn[2]) * wzs;
            const float area = ex * ey + ey * ez + ez * ex;
            float best = 3.0e38f;
            unsigned packed = 0u, bk = 1u;
#pragma unroll
            for (int kk = 1; kk <= 7; ++kk) {               // root of a wide node: kk slots to the left child, 8 - kk to the right
                const float cc = Lc[kk - 1] + Rc[7 - kk];
                if (cc < best) { best = cc; bk = (unsigned)kk; }
            }
            packed |= bk << 24;
            const float c1 = area + best;
            cost[0] = c1;
#pragma unroll
            for (int i = 2; i <= 7; ++i) {
                float bi = c1;
                unsigned ki = 0u;
#pragma unroll
                for (int kk = 1; kk < i; ++kk) {
                    const float cc = Lc[kk - 1] + Rc[i - kk - 1];
                    if (cc < bi) { bi = cc; ki = (unsigned)kk; }
                }
                cost[i - 1] = bi;
                packed |= ki << (3 * i);
            }
            up[node].y = packed;                            // bits 3i..3i+2: slots for the left child at budget i (0: stay one slot); 24..26: the root split
            // starting point of the eight-wide builder's budget resolution (bvh_oct_budget_kernel): for every internal child c,
            // jump[c] = (this node, F) with F[i - 1] = budget of c when this node holds budget i (a nibble each)
            const int cl = (int)rec[6], cr = (int)rec[7];
            unsigned Fl = 0u, Fr = 0u;
#pragma unroll
            for (unsigned i = 1; i <= 8; ++i) {
                Fl |= oct_child_budget(packed, i, 0) << (4u * (i - 1u));
                Fr |= oct_child_budget(packed, i, 1) << (4u * (i - 1u));
            }
            if (cl >= 0) jump[cl] = ((unsigned long long)Fl << 32) | (unsigned)node;
            if (cr >= 0) jump[cr] = ((unsigned long long)Fr << 32) | (unsigned)node;
            if (node == 0) jump[0] = 0x8888888800000000ull;         // the tree root holds budget 8 whatever is asked
        }
        if (node == 0) {
            if (DP) info->height = height;
            return;
        }
        w = w_next;
    }
}

// ---------------------------------------------------------------------------------------------
// ray-query hooks

template <bool COUNT>
__global__ void __launch_bounds__(NVDR_QUERY_BLOCK) trace_visibility_kernel(BvhView bvh, const float *__restrict__ ro,
                                                                             const float *__restrict__ rd, int64_t n_rays,
                                                                             uint8_t *__restrict__ out,
                                                                             unsigned long long *counters, int *spill)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const TravStack stack = make_stack(smem, spill, bvh.stack_max, bvh.overflow);
    unsigned nb = 0, nt = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        const bool occ = bvh_any_hit<COUNT>(bvh, ro[3 * r], ro[3 * r + 1], ro[3 * r + 2], rd[3 * r], rd[3 * r + 1],
                                            rd[3 * r + 2], stack, nb, nt);
        out[r] = occ ? 0 : 1;
    }
    if (COUNT) {
        for (int o = 32; o >= 1; o >>= 1) {
            nb += __shfl_xor(nb, o);
            nt += __shfl_xor(nt, o);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[0], (unsigned long long)nb);
            atomicAdd(&counters[1], (unsigned long long)nt);
        }
    }
}

// closest hit (ordered traversal, shrinking tmax); used by the G-buffer producer
__global__ void __launch_bounds__(NVDR_QUERY_BLOCK) trace_closest_kernel(BvhView bvh, const float *__restrict__ ro,
                                                                          const float *__restrict__ rd, int64_t n_rays,
                                                                          float *__restrict__ out_t,
                                                                          int32_t *__restrict__ out_tri,
                                                                          float *__restrict__ out_uv, int *spill)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const TravStack stack = make_stack(smem, spill, bvh.stack_max, bvh.overflow);
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        float best_t, best_u, best_v;
        const int best = bvh_closest_hit(bvh, ro[3 * r], ro[3 * r + 1], ro[3 * r + 2], rd[3 * r], rd[3 * r + 1], rd[3 * r + 2], stack,
                                         best_t, best_u, best_v);
        out_t[r] = best >= 0 ? best_t : -1.0f;
        out_tri[r] = best;
        out_uv[2 * r] = best_u;
        out_uv[2 * r + 1] = best_v;
    }
}

// ---------------------------------------------------------------------------------------------
// Eight-wide nodes (layout: bvh.h "oct"): the collapse of the fitted binary tree the dynamic programme of bvh_fit_kernel chose,
// built WITHOUT any dependency between threads (round 4; rounds 2-3 built it top-down, one thread per oct node behind a ticket
// queue: every oct level waited for the one above -- 0.73 ms for 684 k triangles, 0.2 ms for bob on 11 workgroups, and ~130 k
// resident threads polling their task word):
//   1. budgets.   Whether binary node v becomes the root of a wide node, or a slot of an ancestor's that is split further, depends on
//      the BUDGET b(v) in 1..7 its parent hands it: b(child) = F_child(b(parent)) with F read off the parent's DP record (root split
//      if the parent is a wide root -- budget 1, or its record says "stay one slot" --, else the split for that budget).  The maps
//      (8 entries of 4 bits) compose; every node resolves its budget by lock-free pointer jumping over (ancestor, map) pairs until its
//      map is constant -- budgets fall by at least one per level inside a wide node, so most maps are constant after a few levels,
//      and the jumps double for the rest.  Wide roots are appended to a list.
//   2. count.     Every wide root expands its slots (as the old builder did) and counts its internal / leaf slots.
//   3. scan.      Exclusive prefix sum of the counts over the binary node index = where every wide node's children block and
//      triangle block start.  The layout is therefore a function of the tree alone (the ticket builder's depended on thread timing),
//      and follows the Karras numbering, i.e. the Morton order of the triangles.
//   4. emit.      Every wide root expands again, finds its own index (parent's block + its position, recorded in step 2) and writes
//      its 64-byte node and its triangles.
// Slots are ordered internal-first (children contiguous in oct[], no per-slot index), the internal ones by rising surface area (the walk
// takes a group's children lowest index first, and an any-hit ray is done at its first hit), and their 16-bit boxes re-quantised to
// 8 bits in the node's own frame.


#ifndef NVDR_OCT_ORDER
#define NVDR_OCT_ORDER 1        // internal children of a wide node: smaller surface area first (0: slot order, 2: larger first; A/B)
#endif
struct OctBuildArgs {
    const uint4 *nodes;     // fitted binary nodes
    const float4 *tris;     // triangle records in Morton order
    uint4 *oct;
    float4 *tris8;
    unsigned long long *jump;   // [n] per binary node: (ancestor, budget map) of the budget resolution, see bvh_oct_budget_kernel
    const uint2 *up;        // per binary node: .x = (parent << 2) | (side << 1) | local (bvh_hierarchy_kernel), .y = the slot splits the collapse DP chose (bvh_fit_kernel)
    int *roots;             // [n] the wide roots (binary node ids), in no particular order
    unsigned *wslot;        // [n] per wide root: (parent wide root << 3) | position among the parent's internal slots
    unsigned long long *cnt;    // [n] per binary node: (internal slots << 32) | leaf slots of the wide node rooted there, 0 elsewhere
    unsigned long long *scan;   // [n] exclusive prefix sum of cnt
    unsigned *ctl;
    int *ovf;               // the context's host-mapped fault word (bit 1: a loop of the build gave up at its iteration bound)
    const BvhDeviceInfo *info;
    int n_int_nodes;        // n_tris - 1
};

__global__ void bvh_oct_init_kernel(OctBuildArgs a, int n_tris)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.ctl[OCT_CTL_ROOTS] = 0u;
        a.ctl[OCT_CTL_ALLOC] = 1u;
        a.ctl[OCT_CTL_DONE] = n_tris > 1 ? 0u : 1u;
        a.ctl[OCT_CTL_TRIS] = n_tris > 1 ? 0u : 1u;
        if (n_tris == 1) {
            // one triangle, no binary node: a root with a single leaf slot that spans the whole grid (org 0, cell 2^9)
            a.oct[0] = make_uint4(0u, (9u << 16) | (9u << 20) | (9u << 24), 0u, 1u << 28);
            a.oct[1] = make_uint4(0u, 0u, 0u, 0u);
            a.oct[2] = make_uint4(0u, 0u, 0xffu, 0u);
            a.oct[3] = make_uint4(0xffu, 0u, 0xffu, 0u);
            a.tris8[0] = a.tris[0]; a.tris8[1] = a.tris[1]; a.tris8[2] = a.tris[2];
        }
    }
}

__global__ void __launch_bounds__(1024) bvh_oct_budget_kernel(OctBuildArgs a)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    bool is_root = false;
    if (v < a.n_int_nodes) {
        // jump[v] = (anc, G): "v holds budget G[i - 1] when its ancestor anc holds budget i" (a nibble per i = 1..8).  bvh_fit_kernel left
        // (parent, F_v) there; the tree root's entry is the constant 8.  Lock-free pointer jumping: compose with the ancestor's CURRENT
        // entry -- whatever it has reached by now is a true statement about it -- publish the longer jump for the nodes below, repeat
        // until G is constant.  (The first version of this kernel walked up one level at a time until the composition became constant:
        // a handful of levels for most nodes, but all the way to the root for some, and the longest walk x the latency of a
        // dependent load was the kernel's time whatever the loads cost: 0.13 ms on 684 k triangles.)
        unsigned long long e = a.jump[v];
        unsigned G = (unsigned)(e >> 32), anc = (unsigned)e;
        // (the bound -- far above the tree's depth, which bounds the jumps -- only turns inconsistent input into an error report
        // instead of a hung queue)
        for (unsigned guard = 0u; G != (G & 15u) * 0x11111111u; ++guard) {
            if (guard > 4096u) {
                atomicOr(a.ovf, 2);
                break;
            }
            const unsigned long long t = __hip_atomic_load(&a.jump[anc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned G2 = (unsigned)(t >> 32);
            unsigned H = 0u;
#pragma unroll
            for (unsigned i = 0; i < 8; ++i) {
                const unsigned mid = (G2 >> (4u * i)) & 15u;                // what anc holds when ITS ancestor holds i + 1 ...
                H |= ((G >> (4u * (mid - 1u))) & 15u) << (4u * i);          // ... and what v holds then
            }
            G = H;
            anc = (unsigned)t;
            __hip_atomic_store(&a.jump[v], ((unsigned long long)G << 32) | anc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned b = G & 15u;
        const unsigned own = a.up[v].y;
        const unsigned rec = own;
        is_root = v == 0 || b == 1u || (b <= 7u && ((rec >> (3u * b)) & 7u) == 0u);
        a.cnt[v] = 0ull;
    }
    // append the wide roots: ONE atomic per workgroup of 1 024 threads.  (One per wavefront -- 10.7 k atomics on the same word for 684 k
    // triangles -- was 0.13 ms whatever the rest of the kernel did: same-address atomics retire one every ~12 ns.)
    __shared__ unsigned wave_cnt[16], wave_base[16];
    const unsigned long long m = __ballot(is_root);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) { wave_base[w2] = tot; tot += wave_cnt[w2]; }
        const unsigned base = tot ? atomicAdd(&a.ctl[OCT_CTL_ROOTS], tot) : 0u;
        for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) wave_base[w2] += base;
    }
    __syncthreads();
    if (is_root) a.roots[wave_base[wave] + __popcll(m & ((1ull << lane) - 1ull))] = v;
}

// A thread's eight slots live in LDS, word-major (word w of slot k of thread t at ((k * 4 + w) * OCT_THREADS + t): consecutive threads
// hit consecutive banks).  Private arrays with run-time indices would live in scratch memory: the first version did, and spent
// 0.3 ms on bob's 3 300 nodes (~30 us per node, ten levels deep).  A slot is the four words of a child in the binary node record --
// reference + the 16-bit box packed in three -- and a workgroup has 128 threads: 16 KB of LDS (rounds 2-4: seven unpacked words x
// 256 threads = 57 KB, which did not fit beside the sample-generation kernel's workgroups: at one view per GPU, where the build is what
// the traversal waits for, bvh_oct_count_kernel took 148 us against 27 us alone).
#define OCT_THREADS 128
#define OCT_SLOT_WORDS 4        // ref, lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16
#define OCT_W(k, w) sl[((k) * OCT_SLOT_WORDS + (w)) * OCT_THREADS]
// field f of slot k: 0 = the reference, 1..3 = lo.xyz, 4..6 = hi.xyz
#define OCT_SL(k, f) oct_field(sl, (k), (f))
__device__ __forceinline__ int oct_field(const int *sl, int k, int f)
{
    if (f == 0) return OCT_W(k, 0);
    const unsigned w = (unsigned)OCT_W(k, 1 + ((f - 1) >> 1));
    return (int)(((f - 1) & 1) ? (w >> 16) : (w & 0xffffu));
}

__device__ __forceinline__ void oct_load_children(const uint4 *__restrict__ nodes, int b, int *sl, int kl, int kr)
{
    const uint4 p = nodes[2 * (int64_t)b], q = nodes[2 * (int64_t)b + 1];
    OCT_W(kl, 0) = (int)q.z; OCT_W(kl, 1) = (int)p.x; OCT_W(kl, 2) = (int)p.y; OCT_W(kl, 3) = (int)p.z;
    OCT_W(kr, 0) = (int)q.w; OCT_W(kr, 1) = (int)p.w; OCT_W(kr, 2) = (int)q.x; OCT_W(kr, 3) = (int)q.y;
}

// the slots of the wide node rooted at binary node b: n of them in LDS, `perm` = their order (4 bits per position: internal slots
// first, smaller surface area first, then the leaves)
__device__ __forceinline__ void oct_expand(const OctBuildArgs &a, int b, int *sl, int &n, unsigned &perm, int &n_int, int &n_leaf)
{
    n = 2;
    oct_load_children(a.nodes, b, sl, 0, 1);
    const float wx = 1.0f / a.info->g_scale[0], wy = 1.0f / a.info->g_scale[1], wz = 1.0f / a.info->g_scale[2];
    // the split the DP chose: slot k carries a budget; an internal child with budget i > 1 whose table says "give s of them to my
    // left child" is replaced by its two children with budgets (s, i - s)
    int budget[8];
    {
        const unsigned root = a.up[b].y >> 24;
        budget[0] = (int)root;
        budget[1] = 8 - (int)root;
    }
    bool again = true;
    while (again) {
        again = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= n) continue;
            const int c = OCT_SL(k, 0);
            if (c < 0 || budget[k] < 2) continue;
            const unsigned sp = (a.up[c].y >> (3 * budget[k])) & 7u;
            if (sp == 0u) { budget[k] = 1; continue; }         // stays one slot (decided once)
            const int bl = (int)sp, br = budget[k] - (int)sp;
            oct_load_children(a.nodes, c, sl, k, n);
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (j == k) budget[j] = bl; if (j == n) budget[j] = br; }
            n++;
            again = true;
        }
    }
    perm = 0u;
    n_int = 0; n_leaf = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < n && OCT_SL(k, 0) >= 0) perm |= (unsigned)k << (4 * n_int++);
#if NVDR_OCT_ORDER
    {
        float ar[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float ex = (float)(OCT_SL(k, 4) - OCT_SL(k, 1)) * wx, ey = (float)(OCT_SL(k, 5) - OCT_SL(k, 2)) * wy,
                        ez = (float)(OCT_SL(k, 6) - OCT_SL(k, 3)) * wz;
            ar[k] = ex * ey + ey * ez + ez * ex;
        }
        auto area_of = [&](unsigned slot) { float a = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) if ((unsigned)k == slot) a = ar[k];
            return a; };
        for (int i = 1; i < n_int; ++i)
            for (int j = i; j > 0; --j) {
                const unsigned a = (perm >> (4 * j)) & 15u, b = (perm >> (4 * (j - 1))) & 15u;
#if NVDR_OCT_ORDER == 2
                if (!(area_of(a) > area_of(b))) break;       // larger first (A/B)
#else
                if (!(area_of(a) < area_of(b))) break;
#endif
                perm = (perm & ~(0xffu << (4 * (j - 1)))) | (a << (4 * (j - 1))) | (b << (4 * j));
            }
    }
#endif
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < n && OCT_SL(k, 0) < 0) perm |= (unsigned)k << (4 * (n_int + n_leaf++));
}

__global__ void __launch_bounds__(OCT_THREADS) bvh_oct_count_kernel(OctBuildArgs a)
{
    __shared__ int slots[8 * OCT_SLOT_WORDS * OCT_THREADS];
    int *sl = slots + threadIdx.x;
    const unsigned n_roots = a.ctl[OCT_CTL_ROOTS];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_roots; i += gridDim.x * blockDim.x) {
        const int b = a.roots[i];
        int n, n_int, n_leaf;
        unsigned perm;
        oct_expand(a, b, sl, n, perm, n_int, n_leaf);
        a.cnt[b] = ((unsigned long long)n_int << 32) | (unsigned long long)n_leaf;
        for (int p = 0; p < n_int; ++p) a.wslot[OCT_SL((perm >> (4 * p)) & 15u, 0)] = ((unsigned)b << 3) | (unsigned)p;
    }
}

// Exclusive prefix sum of cnt[] in three launches without any dependency between workgroups (reduce per 2 048 elements -> one
// workgroup scans the partials -> every workgroup scans its elements from its offset).  (rocPRIM's single-pass scan spins on its
// predecessors -- decoupled look-back -- and, sharing the GPU with the persistent sample-generation kernel, 2 builds in 51 waited
// 0.2-0.3 ms for it.)  Both halves of the packed (internal, leaf) counters are summed at once: neither can carry into the other
// (< 2^29 each).
#define OCT_SCAN_TILE 2048
__global__ void __launch_bounds__(256) bvh_oct_scan_reduce_kernel(const unsigned long long *__restrict__ cnt, int n, unsigned long long *__restrict__ part)
{
    __shared__ unsigned long long red[4];
    const int base = blockIdx.x * OCT_SCAN_TILE;
    unsigned long long sum = 0ull;
#pragma unroll
    for (int j = 0; j < OCT_SCAN_TILE / 256; ++j) {
        const int i = base + j * 256 + threadIdx.x;
        if (i < n) sum += cnt[i];
    }
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(1024) bvh_oct_scan_partials_kernel(unsigned long long *part, int n_part, unsigned *ctl)
{
    // one workgroup, exclusive scan in place, 1 024 partials per round with a running carry
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0ull;
    __syncthreads();
    for (int base = 0; base < n_part; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned long long v = i < n_part ? part[i] : 0ull;
        unsigned long long inc = v;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wave_sum[wave] = inc;
        __syncthreads();
        unsigned long long before = carry_s;
        for (int w2 = 0; w2 < wave; ++w2) before += wave_sum[w2];
        if (i < n_part) part[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
    // the grand totals: oct nodes = the root + every internal slot, triangles placed = every leaf slot
    if (threadIdx.x == 0) { ctl[OCT_CTL_ALLOC] = 1u + (unsigned)(carry_s >> 32); ctl[OCT_CTL_TRIS] = (unsigned)(carry_s & 0xffffffffull); }
}

__global__ void __launch_bounds__(256) bvh_oct_scan_apply_kernel(const unsigned long long *__restrict__ cnt, int n, const unsigned long long *__restrict__ part,
                                                                  unsigned long long *__restrict__ scan)
{
    __shared__ unsigned long long wave_sum[4];
    const int base = blockIdx.x * OCT_SCAN_TILE + threadIdx.x * (OCT_SCAN_TILE / 256);     // eight consecutive elements per thread
    unsigned long long v[OCT_SCAN_TILE / 256], sum = 0ull;
#pragma unroll
    for (int j = 0; j < OCT_SCAN_TILE / 256; ++j) {
        v[j] = base + j < n ? cnt[base + j] : 0ull;
        sum += v[j];
    }
    unsigned long long inc = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    unsigned long long run = part[blockIdx.x] + inc - sum;
    for (int w2 = 0; w2 < wave; ++w2) run += wave_sum[w2];
#pragma unroll
    for (int j = 0; j < OCT_SCAN_TILE / 256; ++j) {
        if (base + j < n) scan[base + j] = run;
        run += v[j];
    }
}

// the three launches above in one workgroup, for trees of up to OCT_SCAN_SMALL binary nodes (the headline mesh has 10 687)
#define OCT_SCAN_SMALL 16384
__global__ void __launch_bounds__(1024) bvh_oct_scan_small_kernel(const unsigned long long *__restrict__ cnt, int n, unsigned long long *__restrict__ scan,
                                                                   unsigned *ctl)
{
    __shared__ unsigned long long wave_sum[16];
    constexpr int PER = OCT_SCAN_SMALL / 1024;
    const int base = threadIdx.x * PER;
    unsigned long long v[PER], sum = 0ull;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = base + j < n ? cnt[base + j] : 0ull;
        sum += v[j];
    }
    unsigned long long inc = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    unsigned long long run = inc - sum;
    for (int w2 = 0; w2 < wave; ++w2) run += wave_sum[w2];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (base + j < n) scan[base + j] = run;
        run += v[j];
    }
    if (threadIdx.x == 1023) { ctl[OCT_CTL_ALLOC] = 1u + (unsigned)(run >> 32); ctl[OCT_CTL_TRIS] = (unsigned)(run & 0xffffffffull); }
}

__global__ void __launch_bounds__(OCT_THREADS) bvh_oct_emit_kernel(OctBuildArgs a)
{
    __shared__ int slots[8 * OCT_SLOT_WORDS * OCT_THREADS];
    int *sl = slots + threadIdx.x;
    const unsigned n_roots = a.ctl[OCT_CTL_ROOTS];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_roots; i += gridDim.x * blockDim.x) {
        const int b = a.roots[i];
        int n, n_int, n_leaf;
        unsigned perm;
        oct_expand(a, b, sl, n, perm, n_int, n_leaf);
        const unsigned long long sc = a.scan[b];
        const unsigned cb = 1u + (unsigned)(sc >> 32), tb = (unsigned)(sc & 0xffffffffull);
        unsigned m = 0u;
        if (b != 0) {
            const unsigned ws = a.wslot[b];
            m = 1u + (unsigned)(a.scan[ws >> 3] >> 32) + (ws & 7u);
        }
        // the node's frame: lower corner + one power-of-two cell per axis such that the extent fits 8 bits
        int org[3], e[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            int lo = 65535, hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < n) { lo = min(lo, OCT_SL(k, 1 + ax)); hi = max(hi, OCT_SL(k, 4 + ax)); }
            org[ax] = lo;
            int ee = 0;
            while ((((hi - lo) + (1 << ee) - 1) >> ee) > 255) ++ee;
            e[ax] = ee;
        }
        unsigned planes[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // lo.x[2], lo.y[2], lo.z[2], hi.x[2], hi.y[2], hi.z[2]
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p >= n) continue;
            const int k = (int)((perm >> (4 * p)) & 15u);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const unsigned qlo = (unsigned)((OCT_SL(k, 1 + ax) - org[ax]) >> e[ax]);                              // floor
                const unsigned qhi = (unsigned)(((OCT_SL(k, 4 + ax) - org[ax]) + (1 << e[ax]) - 1) >> e[ax]);          // ceil
                planes[2 * ax + (p >> 2)] |= qlo << (8 * (p & 3));
                planes[6 + 2 * ax + (p >> 2)] |= qhi << (8 * (p & 3));
            }
        }
        uint4 *o = a.oct + 4 * (int64_t)m;
        o[0] = make_uint4((unsigned)org[0] | ((unsigned)org[1] << 16),
                          (unsigned)org[2] | ((unsigned)e[0] << 16) | ((unsigned)e[1] << 20) | ((unsigned)e[2] << 24),
                          cb | ((unsigned)n_int << 28), tb | ((unsigned)n_leaf << 28));
        o[1] = make_uint4(planes[0], planes[1], planes[2], planes[3]);
        o[2] = make_uint4(planes[4], planes[5], planes[6], planes[7]);
        o[3] = make_uint4(planes[8], planes[9], planes[10], planes[11]);
        for (int j = 0; j < n_leaf; ++j) {
            const int k = (int)((perm >> (4 * (n_int + j))) & 15u);
            const int64_t src = 3 * (int64_t)(~OCT_SL(k, 0)), dst = 3 * (int64_t)(tb + j);
            a.tris8[dst] = a.tris[src]; a.tris8[dst + 1] = a.tris[src + 1]; a.tris8[dst + 2] = a.tris[src + 2];
        }
    }
    // (counters for the export / the structural checks: nodes written = wide roots; the totals of the internal and the leaf slots
    // come out of the prefix sum, bvh_oct_scan_partials_kernel)
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl[OCT_CTL_DONE] = n_roots;
}

// The treetop table (bvh.h): the first nodes of the eight-wide tree in breadth-first order, one workgroup, a thread per entry.
// Level by level: the threads of a level read their node's header, a prefix sum over their internal-child counts places the children behind
// the level; a node's group moves into the table only as a whole, and only while everything before it in the level did (the table is a
// prefix of the breadth-first order).  Entries copy their node; the child_base of a group that moved points into the table (NVDR_OCT_TOP_FLAG).
#if NVDR_TRACE_TOP_MAX
__global__ void __launch_bounds__(NVDR_TRACE_TOP_MAX) bvh_oct_top_kernel(const uint4 *__restrict__ oct, uint4 *__restrict__ top, unsigned K)
{
    __shared__ unsigned gidx[NVDR_TRACE_TOP_MAX], scan_s[2][NVDR_TRACE_TOP_MAX];
    __shared__ unsigned level_end;
    const unsigned i = threadIdx.x;
    if (i == 0) { gidx[0] = 0u; level_end = 1u; }
    unsigned lo = 0u, hi = 1u, moved_to = 0u;
    bool have = false, moved = false;
    uint4 h = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    while (lo < hi) {
        const bool mine = i >= lo && i < hi;
        unsigned n_int = 0u, cb = 0u;
        if (mine) {
            h = oct[4 * (int64_t)gidx[i]];
            have = true;
            n_int = h.z >> 28;
            cb = h.z & (NVDR_OCT_TOP_FLAG - 1u);
        }
        // inclusive prefix sum of n_int over the workgroup (Hillis-Steele, double-buffered)
        int cur = 0;
        scan_s[0][i] = n_int;
        __syncthreads();
        for (unsigned o = 1u; o < K; o <<= 1) {
            scan_s[cur ^ 1][i] = scan_s[cur][i] + (i >= o ? scan_s[cur][i - o] : 0u);
            cur ^= 1;
            __syncthreads();
        }
        const unsigned inc = scan_s[cur][i];
        if (mine && n_int != 0u && hi + inc <= K) {
            moved = true;
            moved_to = hi + inc - n_int;
            for (unsigned c2 = 0u; c2 < n_int; ++c2) gidx[moved_to + c2] = cb + c2;
            atomicMax(&level_end, hi + inc);
        }
        __syncthreads();
        lo = hi;
        hi = level_end;
        __syncthreads();
    }
    if (have) {
        const uint4 *src = oct + 4 * (int64_t)gidx[i];
        if (moved) h.z = (h.z & 0xF0000000u) | NVDR_OCT_TOP_FLAG | moved_to;
        top[4 * i + 0] = h; top[4 * i + 1] = src[1]; top[4 * i + 2] = src[2]; top[4 * i + 3] = src[3];
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// host side

// A traversal kernel of an EARLIER launch on this context pushed beyond the stack bound (bvh.h): its visibility results
// were not trustworthy.  The flag lives in host-mapped memory, so this costs no synchronisation.
int ctx_check_overflow(nvdr_ctx *c, const char *who)
{
    const int f = c->ovf_host ? *(volatile int *)c->ovf_host : 0;
    if (f & 1) {
        nvdr_set_error("%s: a traversal launch on this context overflowed its %d-entry stack (degenerate BVH); "
                       "the visibility it produced is invalid; the context must be destroyed", who, c->stack_max);
        return -2;
    }
    if (f & 6) {
        nvdr_set_error("%s: %s on this context did not terminate within its iteration bound (internal error); "
                       "results are invalid; the context must be destroyed", who, (f & 2) ? "the eight-wide BVH build" : "a shadow-ray launch");
        return -3;
    }
    return 0;
}

// Every consumer of the tree calls this before its first launch that reads it: the build may still be running on the side stream.
int ctx_wait_built(nvdr_ctx *c, hipStream_t stream)
{
    if (c->queued) { if (int r = ctx_launch_build(c, stream)) return r; }      // (a deferred build: launched now, on the side stream)
    if (!c->built_pending || (c->built_waited_valid && c->built_waited == stream)) return 0;
    NVDR_HIP_TRY(hipStreamWaitEvent(stream, c->ev_built, 0));
    c->built_waited = stream;
    c->built_waited_valid = true;
    return 0;
}

extern "C" int nvdr_bvh_wait(nvdr_ctx *c, void *stream_)
{
    NVDR_REQUIRE(c != nullptr, "nvdr_bvh_wait: ctx is NULL");
    return ctx_wait_built(c, (hipStream_t)stream_);
}

extern "C" int nvdr_bvh_mark_joined(nvdr_ctx *c)
{
    NVDR_REQUIRE(c != nullptr, "nvdr_bvh_mark_joined: ctx is NULL");
    NVDR_REQUIRE(!c->queued, "nvdr_bvh_mark_joined: a deferred build has not been launched yet");
    c->built_pending = false;
    c->built_waited_valid = false;
    return 0;
}

extern "C" int nvdr_ctx_check(nvdr_ctx *c, void *stream_)
{
    NVDR_REQUIRE(c != nullptr, "nvdr_ctx_check: ctx is NULL");
    NVDR_HIP_TRY(hipSetDevice(c->device));
    if (c->queued) { if (int rq = ctx_launch_build(c, (hipStream_t)stream_)) return rq; }
    if (c->build_stream) NVDR_HIP_TRY(hipStreamSynchronize(c->build_stream));
    NVDR_HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    return ctx_check_overflow(c, "nvdr_ctx_check");
}

static int ctx_free_bvh(nvdr_ctx *c)
{
    ctx_free(c, c->nodes); ctx_free(c, c->oct); ctx_free(c, c->tris8); ctx_free(c, c->oct_task); ctx_free(c, c->tris);
    ctx_free(c, c->keys[0]); ctx_free(c, c->keys[1]); ctx_free(c, c->vals[0]); ctx_free(c, c->vals[1]);
    ctx_free(c, c->up); ctx_free(c, c->flags); ctx_free(c, c->dp_cost); ctx_free(c, c->sort_tmp);
    ctx_free(c, c->oct_wslot); ctx_free(c, c->oct_jump); ctx_free(c, c->oct_cnt); ctx_free(c, c->oct_scan);
    c->sort_tmp_bytes = 0;
    c->cap_tris = 0;
    return 0;
}

extern "C" int nvdr_ctx_create(nvdr_ctx **out, int device)
{
    NVDR_REQUIRE(out != nullptr, "nvdr_ctx_create: out is NULL");
    NVDR_HIP_TRY(hipSetDevice(device));
    nvdr_ctx *c = new nvdr_ctx();
    c->device = device;
    hipError_t e = hipMalloc((void **)&c->dinfo, sizeof(BvhDeviceInfo));
    if (e != hipSuccess) {
        delete c;
        nvdr_set_error("nvdr_ctx_create: hipMalloc failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    hipMemset(c->dinfo, 0, sizeof(BvhDeviceInfo));
    // host-mapped flag a traversal kernel raises when a stack push would exceed the context's bound (bvh.h)
    e = hipHostMalloc((void **)&c->ovf_host, 2 * sizeof(int), hipHostMallocMapped);      // [0] the fault word, [1] live rays of the last traversal launch (launch_trace)
    if (e == hipSuccess) {
        c->ovf_host[0] = 0;
        c->ovf_host[1] = 0;
        e = hipHostGetDevicePointer((void **)&c->ovf_dev, c->ovf_host, 0);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&c->chunk_counts, sizeof(unsigned) * NVDR_MAX_CHUNKS * 544);
    if (e == hipSuccess) e = hipMalloc((void **)&c->queues, sizeof(unsigned) * 32 * 256);
    if (e == hipSuccess) e = hipMalloc((void **)&c->oct_ctl, sizeof(unsigned) * OCT_CTL_WORDS);
    if (e == hipSuccess) e = hipMalloc((void **)&c->bounds_part, sizeof(float) * 6 * BVH_BOUNDS_BLOCKS + 128);
    if (e == hipSuccess) { c->bounds_ticket = (unsigned *)(c->bounds_part + 6 * BVH_BOUNDS_BLOCKS); e = hipMemset(c->bounds_ticket, 0, 128); }
    if (e != hipSuccess) {
        hipFree(c->dinfo);
        if (c->ovf_host) hipHostFree(c->ovf_host);
        delete c;
        nvdr_set_error("nvdr_ctx_create: allocation failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    hipMemset(c->chunk_counts, 0, sizeof(unsigned) * NVDR_MAX_CHUNKS * 544);
    hipMemset(c->queues, 0, sizeof(unsigned) * 32 * 256);
    if (const char *e = getenv("NVDR_STREAM_BUDGET_MB")) {
        const long long mb = atoll(e);
        if (mb >= 1) c->stream_budget = (int64_t)mb << 20;
    }
    // NVDR_DEBUG (experiments only: 1 skip tracing, 2 skip the light gradient, 8 trace dead samples, 16 light-gradient
    // atomics instead of the band gather, 32 pretend the traversal stacks hold 13 / 2 entries, 64 an explicit reset kernel in
    // front of the traversal kernel) is read ONCE here, not per launch, and announced when set
    if (const char *dbg = nvdr_tuning_env("NVDR_DEBUG")) {
        c->debug = (unsigned)atoi(dbg);
        if (c->debug) fprintf(stderr, "[nvdr] NVDR_DEBUG=%u is active on this context (experiment switches; not for production)\n", c->debug);
    }
    if (const char *pb = nvdr_tuning_env("NVDR_PBLOCKS")) {
        sscanf(pb, "%d,%d,%d", &c->per_cu[0], &c->per_cu[1], &c->per_cu[2]);
        c->per_cu_user = true;
        for (int k = 0; k < 3; ++k) c->per_cu[k] = c->per_cu[k] < 1 ? 1 : (c->per_cu[k] > 16 ? 16 : c->per_cu[k]);
    }
    if (const char *ab = nvdr_tuning_env("NVDR_ASYNC_BUILD")) c->async_build = atoi(ab) != 0;
    // NVDR_OCT_CLEAF: cost of a triangle test in units of a node step in the collapse DP (experiments)
    if (const char *cl = nvdr_tuning_env("NVDR_OCT_CLEAF")) { const float v = (float)atof(cl); if (v > 0.0f) c->oct_c_leaf = v; }
    // NVDR_LG_MODE: work split of the light-gradient gather (env_shade.hip): 0 every workgroup walks all bands, 1 one set of
    // workgroups per band, unset = by launch size
    if (const char *rk = nvdr_tuning_env("NVDR_REFIT_KEEP_COLLAPSE")) c->refit_keep_collapse = atoi(rk) != 0;
    if (const char *lm = nvdr_tuning_env("NVDR_LG_MODE")) c->lg_mode = atoi(lm) ? 1 : 0;
    if (const char *lf = nvdr_tuning_env("NVDR_LG_F64")) c->lg_f64 = atoi(lf) > 0 ? 1 : (atoi(lf) < 0 ? -1 : 0);
    if (const char *sq = nvdr_tuning_env("NVDR_SHADE_QUEUE")) c->shade_queue = atoi(sq) & 7;
    if (const char *sm = nvdr_tuning_env("NVDR_TRACE_SPLIT_MODE")) c->trace_split_mode = atoi(sm);
    if (const char *tt = nvdr_tuning_env("NVDR_TRACE_TOP_NODES")) {
        const int k = atoi(tt);
        c->trace_top = k < 0 ? 0 : (k > NVDR_TRACE_TOP_MAX ? NVDR_TRACE_TOP_MAX : k);
    }
    (void)hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device);
    if (c->n_cus <= 0) c->n_cus = 256;
    *out = c;
    return 0;
}

extern "C" int nvdr_ctx_destroy(nvdr_ctx *c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    // the buffers go back to the caller's allocator (torch's caching pool): nothing enqueued on ANY stream may still be reading them
    (void)hipDeviceSynchronize();
    if (c->build_stream) {
        (void)hipStreamSynchronize(c->build_stream);
        (void)hipEventDestroy(c->ev_inputs);
        (void)hipEventDestroy(c->ev_built);
        (void)hipStreamDestroy(c->build_stream);
    }
    ctx_free(c, c->in_verts);
    ctx_free(c, c->in_tris);
    ctx_free_bvh(c);
    if (c->prof_ev[0][0])
        for (int i = 0; i < NVDR_PROF_RING; ++i)
            for (int k = 0; k < 4; ++k) hipEventDestroy(c->prof_ev[i][k]);
    hipFree(c->dinfo);
    hipFree(c->chunk_counts);
    hipFree(c->queues);
    hipFree(c->oct_ctl);
    hipFree(c->bounds_part);
    ctx_free(c, c->spill);
    ctx_free(c, c->pix_list);
    ctx_free(c, c->rays); ctx_free(c, c->texel); ctx_free(c, c->vis); ctx_free(c, c->live); ctx_free(c, c->pix_origin); ctx_free(c, c->pix_setup); ctx_free(c, c->pix_grad); ctx_free(c, c->lg_part); ctx_free(c, c->lg_tags); ctx_free(c, c->cdf_guide);
    if (c->ovf_host) hipHostFree(c->ovf_host);
    delete c;
    return 0;
}

static int ctx_reserve(nvdr_ctx *c, int64_t n_tris, hipStream_t stream)
{
    if (n_tris <= c->cap_tris) return 0;
    // the reference frees and re-allocates the GAS on every build (torch_bindings.cpp:52,84-95,114);
    // here the buffers only ever grow (by 1.5x) and are reused across iterations
    NVDR_HIP_TRY(hipDeviceSynchronize());
    ctx_free_bvh(c);
    const int64_t cap = n_tris + n_tris / 2 + 64;
    NVDR_HIP_TRY(ctx_malloc(c, &c->nodes, sizeof(uint4) * 2 * cap, stream));
    NVDR_HIP_TRY(ctx_malloc(c, &c->oct, sizeof(uint4) * 4 * (cap + NVDR_TRACE_TOP_MAX), stream));
    c->oct_top = c->oct + 4 * cap;
    NVDR_HIP_TRY(ctx_malloc(c, &c->tris8, sizeof(float4) * 3 * cap, stream));
    NVDR_HIP_TRY(ctx_malloc(c, &c->oct_task, sizeof(int) * cap, stream));
    NVDR_HIP_TRY(ctx_malloc(c, &c->tris, sizeof(float4) * 3 * cap, stream));
    for (int i = 0; i < 2; ++i) {
        NVDR_HIP_TRY(ctx_malloc(c, &c->keys[i], sizeof(uint32_t) * cap, stream));
        NVDR_HIP_TRY(ctx_malloc(c, &c->vals[i], sizeof(uint32_t) * cap, stream));
    }
    NVDR_HIP_TRY(ctx_malloc(c, &c->up, sizeof(uint2) * 2 * cap, stream));
    NVDR_HIP_TRY(ctx_malloc(c, &c->flags, sizeof(int) * cap, stream));
    NVDR_HIP_TRY(ctx_malloc(c, &c->dp_cost, sizeof(float) * 16 * cap, stream));
    NVDR_HIP_TRY(ctx_malloc(c, &c->oct_wslot, sizeof(unsigned) * cap, stream));
    NVDR_HIP_TRY(ctx_malloc(c, &c->oct_jump, sizeof(unsigned long long) * cap, stream));

    NVDR_HIP_TRY(ctx_malloc(c, &c->oct_cnt, sizeof(unsigned long long) * cap, stream));
    NVDR_HIP_TRY(ctx_malloc(c, &c->oct_scan, sizeof(unsigned long long) * cap, stream));
    size_t bytes = 0;
    NVDR_HIP_TRY(rocprim::radix_sort_pairs<nvdr_sort_config>(nullptr, bytes, c->keys[0], c->keys[1], c->vals[0], c->vals[1], (size_t)cap, 0, 30));
    const size_t bytes2 = sizeof(unsigned long long) * (size_t)(cap / OCT_SCAN_TILE + 2);       // partial sums of the slot-count scan
    if (bytes2 > bytes) bytes = bytes2;
    NVDR_HIP_TRY(ctx_malloc(c, &c->sort_tmp, bytes + 256, stream));
    c->sort_tmp_bytes = bytes + 256;
    c->cap_tris = cap;
    return 0;
}

// the build's own copy of the caller's vertices and indices, both in one launch
__global__ void __launch_bounds__(256) bvh_copy_inputs_kernel(const float *__restrict__ v, int64_t nv, float *__restrict__ vo,
                                                              const int32_t *__restrict__ t, int64_t nt, int32_t *__restrict__ to)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv || i < nt; i += stride) {
        if (i < nv) vo[i] = v[i];
        if (i < nt) to[i] = t[i];
    }
}

extern "C" int nvdr_ctx_set_build_mode(nvdr_ctx *c, int mode)
{
    NVDR_REQUIRE(c, "nvdr_ctx_set_build_mode: NULL ctx");
    NVDR_REQUIRE(mode >= 0 && mode <= 2, "nvdr_ctx_set_build_mode: mode %d (0 caller's stream, 1 side stream, 2 side stream with deferred launches)", mode);
    NVDR_REQUIRE(!c->queued, "nvdr_ctx_set_build_mode: a deferred build is waiting for its consumer on this context");
    c->async_build = mode != 0;
    c->build_deferred = mode == 2;
    return 0;
}

extern "C" int nvdr_bvh_build(nvdr_ctx *c, const float *verts, int64_t n_verts, const int32_t *tris, int64_t n_tris,
                              int rebuild, void *stream_)
{
    NvdrRange range(rebuild ? "nvdr_bvh_build" : "nvdr_bvh_build(refit)");
    NVDR_REQUIRE(c != nullptr, "nvdr_bvh_build: ctx is NULL");
    // same message as the Python asserts of the reference (render/optixutils/ops.py:131-132)
    NVDR_REQUIRE(n_tris > 0 && n_verts > 0, "Got empty training triangle mesh (unrecoverable discontinuity)");
    NVDR_REQUIRE(n_tris < (long long)NVDR_OCT_TOP_FLAG, "nvdr_bvh_build: too many triangles (%lld, the limit is 2^27)", (long long)n_tris);
    NVDR_REQUIRE(verts && tris, "nvdr_bvh_build: NULL geometry pointer");
    hipStream_t stream = (hipStream_t)stream_;
    NVDR_HIP_TRY(hipSetDevice(c->device));
    // a context whose traversal stack overflowed (or whose build / walk gave up) is unusable: say so before touching any buffer
    if (int r0 = ctx_check_overflow(c, "nvdr_bvh_build")) return r0;
    if (c->queued) { if (int rq = ctx_launch_build(c, stream)) return rq; }      // a deferred build nobody consumed: a refit may follow it
    if (rebuild == 0) {
        NVDR_REQUIRE(c->n_tris == n_tris && c->n_verts == n_verts,
                     "nvdr_bvh_build: refit (rebuild=0) needs the topology of the last full build "
                     "(%lld tris / %lld verts, got %lld / %lld)",
                     (long long)c->n_tris, (long long)c->n_verts, (long long)n_tris, (long long)n_verts);
    } else {
        int r = ctx_reserve(c, n_tris, (hipStream_t)stream_);
        if (r) return r;
    }
    // spill columns for the proven stack bound of a tree over n_tris triangles (bvh.h); grow-only.  The binary walks keep 4-byte
    // entries beyond their NVDR_STACK_LDS, the oct walk 8-byte entries beyond its NVDR_OSTACK_LDS: sized for the larger of the two
    const int smax = nvdr_stack_bound(n_tris);
    if (smax > c->spill_cap) {
        NVDR_HIP_TRY(hipStreamSynchronize(stream));
        if (c->build_stream) NVDR_HIP_TRY(hipStreamSynchronize(c->build_stream));
        ctx_free(c, c->spill);
        c->spill_cap = 0;
        const size_t d_bin = (size_t)(smax > NVDR_STACK_LDS ? smax - NVDR_STACK_LDS : 0) * sizeof(int);
        const size_t d_oct = (size_t)(smax > NVDR_OSTACK_LDS ? smax - NVDR_OSTACK_LDS : 0) * sizeof(uint2);
        const size_t per_lane = d_bin > d_oct ? d_bin : d_oct;
        NVDR_HIP_TRY(ctx_malloc(c, &c->spill, (per_lane > 0 ? per_lane : sizeof(uint2)) * NVDR_QUERY_MAX_BLOCKS * NVDR_QUERY_BLOCK, stream));
        c->spill_cap = smax;
    }
    // NVDR_DEBUG bit 32 (tests only): pretend the stacks are one entry deeper than their LDS part, to exercise the overflow report
    c->stack_max = (c->debug & 32u) ? NVDR_STACK_LDS + 1 : c->spill_cap;
    c->oct_stack_max = (c->debug & 32u) ? 2 : c->spill_cap;
    const int n = (int)n_tris;
    // the build runs on the context's side stream: it starts when everything enqueued so far on the caller's stream is done (the
    // vertices are ready, earlier traversals have finished with the old tree) and consumers wait for it (ctx_wait_built)
    hipStream_t caller = stream;
    if (c->async_build) {
        if (!c->build_stream) {
            NVDR_HIP_TRY(hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking));
            NVDR_HIP_TRY(hipEventCreateWithFlags(&c->ev_inputs, hipEventDisableTiming));
            NVDR_HIP_TRY(hipEventCreateWithFlags(&c->ev_built, hipEventDisableTiming));
        }
        // the geometry is copied on the caller's stream first (36 B per triangle-ish: microseconds), so that the side stream never
        // reads caller memory: stream-ordered use of verts / tris by the caller stays correct without knowing about the side stream
        if (c->in_verts_cap < n_verts || c->in_tris_cap < n_tris) {
            NVDR_HIP_TRY(hipStreamSynchronize(caller));
            NVDR_HIP_TRY(hipStreamSynchronize(c->build_stream));
            if (c->in_verts_cap < n_verts) {
                ctx_free(c, c->in_verts);
                c->in_verts_cap = 0;
                NVDR_HIP_TRY(ctx_malloc(c, &c->in_verts, sizeof(float) * 3 * (n_verts + n_verts / 2 + 64), caller));
                c->in_verts_cap = n_verts + n_verts / 2 + 64;
            }
            if (c->in_tris_cap < n_tris) {
                ctx_free(c, c->in_tris);
                c->in_tris_cap = 0;
                NVDR_HIP_TRY(ctx_malloc(c, &c->in_tris, sizeof(int32_t) * 3 * (n_tris + n_tris / 2 + 64), caller));
                c->in_tris_cap = n_tris + n_tris / 2 + 64;
            }
        }
        // a build that nobody has waited for yet may still be reading the staging copies (two builds in a row, a build then a refit)
        if (c->built_pending) { if (int rw = ctx_wait_built(c, caller)) return rw; }
        // ONE copy launch for both buffers (two hipMemcpyAsync nodes cost a captured HIP graph ~15 us each on this runtime)
        {
            const int64_t nv3 = 3 * n_verts, nt3 = 3 * n_tris, most = nv3 > nt3 ? nv3 : nt3;
            bvh_copy_inputs_kernel<<<min(div_up(most, 256 * 4), 2048u), 256, 0, caller>>>(verts, nv3, c->in_verts, tris, nt3, c->in_tris);
        }
        verts = c->in_verts;
        tris = c->in_tris;
        NVDR_HIP_TRY(hipEventRecord(c->ev_inputs, caller));
        stream = c->build_stream;
    }
    c->queued_verts = verts; c->queued_tris = tris; c->queued_rebuild = rebuild; c->queued_n_verts = n_verts; c->queued_n_tris = n_tris;
    c->queued = true;
    c->n_tris = n_tris;
    c->n_verts = n_verts;
    c->stream_id = 0; // a ray stream traced against the old geometry must not be reused
    if (c->async_build) {
        c->built_pending = true;        // (consumers call ctx_wait_built, which launches a deferred build first)
        c->built_waited_valid = false;
    }
    // build_mode 2: the launches wait for the first consumer of the tree (ctx_wait_built) -- in program order BEHIND whatever the caller
    // enqueues in between (env-shade: pixel compaction, sample generation), so that in a captured HIP graph those nodes come first
    if (c->async_build && c->build_deferred) return 0;
    return ctx_launch_build(c, stream);
}

__global__ void __launch_bounds__(1024) bvh_clear_flags_kernel(int *__restrict__ flags, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = 0;
}

// the kernels of a prepared build (nvdr_bvh_build), on the context's side stream -- or the caller's, without one
int ctx_launch_build(nvdr_ctx *c, hipStream_t stream)
{
    if (!c->queued) return 0;
    c->queued = false;
    const float *verts = c->queued_verts;
    const int32_t *tris = c->queued_tris;
    const int rebuild = c->queued_rebuild;
    const int64_t n_verts = c->queued_n_verts;
    const int n = (int)c->queued_n_tris;
    if (c->async_build) {
        NVDR_HIP_TRY(hipStreamWaitEvent(c->build_stream, c->ev_inputs, 0));
        stream = c->build_stream;
    }
    bvh_bounds_kernel<<<min(div_up(n_verts, 1024), (unsigned)BVH_BOUNDS_BLOCKS), 256, 0, stream>>>(verts, n_verts, c->dinfo, c->bounds_part, c->bounds_ticket);
    if (rebuild != 0) {
        bvh_morton_kernel<<<div_up(n, 256), 256, 0, stream>>>(verts, tris, n, c->dinfo, c->keys[0], c->vals[0]);
        size_t bytes = c->sort_tmp_bytes;
        NVDR_HIP_TRY(rocprim::radix_sort_pairs<nvdr_sort_config>(c->sort_tmp, bytes, c->keys[0], c->keys[1], c->vals[0], c->vals[1],
                                                                 (size_t)n, 0, 30, stream));
        if (n > 1)
        {
            bvh_hierarchy_kernel<<<div_up(n - 1, 256), 256, 0, stream>>>(c->keys[1], n, c->nodes, c->up, c->flags, c->oct_ctl, (uint2 *)c->oct_scan);
#if NVDR_TREELET_W >= 3
            if (n >= 4) {       // (the ranges live in oct_scan and the list in oct_task until the collapse, which runs behind the fit, takes them over)
                const bool large = n > NVDR_TREELET_LARGE && NVDR_TREELET_W_LARGE < NVDR_TREELET_W;
                const unsigned w = large ? NVDR_TREELET_W_LARGE : NVDR_TREELET_W;
                bvh_treelet_mark_kernel<<<div_up(n - 1, 1024), 1024, 0, stream>>>((const uint2 *)c->oct_scan, c->up, n - 1, w, c->oct_task, c->oct_ctl);
                const unsigned tb = min(div_up(n, 4 * 24), (unsigned)c->n_cus * 8u);
                if (large)
                    bvh_treelet_kernel<NVDR_TREELET_W_LARGE><<<tb < 1u ? 1u : tb, 256, 0, stream>>>(verts, tris, c->vals[1], n, (const uint2 *)c->oct_scan, c->nodes, c->up, c->oct_task, c->oct_ctl);
                else
                    bvh_treelet_kernel<NVDR_TREELET_W><<<tb < 1u ? 1u : tb, 256, 0, stream>>>(verts, tris, c->vals[1], n, (const uint2 *)c->oct_scan, c->nodes, c->up, c->oct_task, c->oct_ctl);
            }
#endif
        }
    }
    const bool cleared = rebuild != 0 && n > 1;        // (the hierarchy kernel has cleared the counters and the control words)
    // (a kernel, not hipMemsetAsync: the refit is replayed from HIP graphs, where memset / memcpy nodes are slow on this runtime)
    if (!cleared) bvh_clear_flags_kernel<<<div_up(n, 1024), 1024, 0, stream>>>(c->flags, n);
    // A REFIT keeps the collapse of the last full build (bvh_fit_kernel<false>): boxes, the slots' order (count) and the nodes (emit) are redone,
    // the dynamic programme, the budget resolution and the prefix sums are not.  NVDR_REFIT_KEEP_COLLAPSE=0 (tuning): every refit collapses anew.
    const bool keep = rebuild == 0 && n > 1 && c->refit_keep_collapse && c->collapse_valid;
    if (keep)
        bvh_fit_kernel<false><<<div_up(n, NVDR_FIT_BLOCK), NVDR_FIT_BLOCK, 0, stream>>>(verts, tris, c->vals[1], n, c->tris, c->nodes, c->up,
                                                                                         c->flags, c->dinfo, c->dp_cost, c->oct_c_leaf, c->oct_jump);
    else
        bvh_fit_kernel<true><<<div_up(n, NVDR_FIT_BLOCK), NVDR_FIT_BLOCK, 0, stream>>>(verts, tris, c->vals[1], n, c->tris, c->nodes, c->up,
                                                                                        c->flags, c->dinfo, c->dp_cost, c->oct_c_leaf, c->oct_jump);
    c->collapse_valid = n > 1;
    {
        // eight-wide nodes for the shadow-ray walk: budgets -> counts -> prefix sum -> emit, no inter-thread dependency (see above)
        OctBuildArgs oa;
        oa.nodes = c->nodes; oa.tris = c->tris; oa.oct = c->oct; oa.tris8 = c->tris8; oa.up = c->up; oa.jump = c->oct_jump;
        oa.roots = c->oct_task; oa.wslot = c->oct_wslot; oa.cnt = c->oct_cnt; oa.scan = c->oct_scan; oa.ctl = c->oct_ctl;
        oa.info = c->dinfo; oa.n_int_nodes = n - 1; oa.ovf = c->ovf_dev;
        if (!cleared && !keep) bvh_oct_init_kernel<<<1, 64, 0, stream>>>(oa, n);
        if (n > 1 && keep) {
            // (the slots of a wide node are ordered by their boxes' areas: the positions of the children are recorded again, the counts come out the same)
            const unsigned blocks = min(div_up((n + 3) / 4, OCT_THREADS), (unsigned)c->n_cus * 16u);
            bvh_oct_count_kernel<<<blocks < 1u ? 1u : blocks, OCT_THREADS, 0, stream>>>(oa);
            bvh_oct_emit_kernel<<<blocks < 1u ? 1u : blocks, OCT_THREADS, 0, stream>>>(oa);
        } else if (n > 1) {
            bvh_oct_budget_kernel<<<div_up(n - 1, 1024), 1024, 0, stream>>>(oa);
            // wide roots are ~n / 4.9; the grid-stride loops of the two expansion kernels cover whatever the device-side count says
            const unsigned blocks = min(div_up((n + 3) / 4, OCT_THREADS), (unsigned)c->n_cus * 16u);
            bvh_oct_count_kernel<<<blocks < 1u ? 1u : blocks, OCT_THREADS, 0, stream>>>(oa);
            const unsigned tiles = div_up(n - 1, OCT_SCAN_TILE);
            unsigned long long *part = (unsigned long long *)c->sort_tmp;      // the sort is done with its scratch by now
            if (n - 1 <= OCT_SCAN_SMALL) {
                bvh_oct_scan_small_kernel<<<1, 1024, 0, stream>>>(c->oct_cnt, n - 1, c->oct_scan, c->oct_ctl);
            } else {
                bvh_oct_scan_reduce_kernel<<<tiles, 256, 0, stream>>>(c->oct_cnt, n - 1, part);
                bvh_oct_scan_partials_kernel<<<1, 1024, 0, stream>>>(part, (int)tiles, c->oct_ctl);
                bvh_oct_scan_apply_kernel<<<tiles, 256, 0, stream>>>(c->oct_cnt, n - 1, part, c->oct_scan);
            }
            bvh_oct_emit_kernel<<<blocks < 1u ? 1u : blocks, OCT_THREADS, 0, stream>>>(oa);
        }
#if NVDR_TRACE_TOP_MAX
        if (c->trace_top > 0) bvh_oct_top_kernel<<<1, (unsigned)((c->trace_top + 63) & ~63), 0, stream>>>(c->oct, c->oct_top, (unsigned)c->trace_top);
#endif
    }
    NVDR_LAUNCH_CHECK();
    if (c->async_build) NVDR_HIP_TRY(hipEventRecord(c->ev_built, c->build_stream));
    return 0;
}

// grid of a grid-stride traversal launch over `items` work items
unsigned query_grid(const nvdr_ctx *c, int64_t items)
{
    int64_t b = (items + NVDR_QUERY_BLOCK - 1) / NVDR_QUERY_BLOCK;
    const int64_t cap = (int64_t)c->n_cus * 8 < NVDR_QUERY_MAX_BLOCKS ? (int64_t)c->n_cus * 8 : NVDR_QUERY_MAX_BLOCKS;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

extern "C" int nvdr_bvh_info_get(nvdr_ctx *c, nvdr_bvh_info *out, void *stream_)
{
    NVDR_REQUIRE(c && out, "nvdr_bvh_info_get: NULL argument");
    NVDR_REQUIRE(c->n_tris > 0, "nvdr_bvh_info_get: no BVH built");
    hipStream_t stream = (hipStream_t)stream_;
    if (int rw = ctx_wait_built(c, stream)) return rw;
    BvhDeviceInfo h;
    NVDR_HIP_TRY(hipMemcpyAsync(&h, c->dinfo, sizeof(h), hipMemcpyDeviceToHost, stream));
    NVDR_HIP_TRY(hipStreamSynchronize(stream));
    out->n_tris = c->n_tris;
    out->n_nodes = c->n_tris - 1;
    out->height = h.height;
    out->root = h.root;
    for (int a = 0; a < 3; ++a) {
        out->aabb_min[a] = ordered_to_float(h.bounds[a]);
        out->aabb_max[a] = ordered_to_float(h.bounds[3 + a]);
        out->grid_lo[a] = h.g_lo[a];
        out->grid_scale[a] = h.g_scale[a];
    }
    out->stack_max = c->stack_max;
    return ctx_check_overflow(c, "nvdr_bvh_info_get");
}

extern "C" int nvdr_bvh_export(nvdr_ctx *c, float *nodes_host, float *tri_host, void *stream_)
{
    NVDR_REQUIRE(c && c->n_tris > 0, "nvdr_bvh_export: no BVH built");
    hipStream_t stream = (hipStream_t)stream_;
    if (int rw = ctx_wait_built(c, stream)) return rw;
    if (nodes_host && c->n_tris > 1)
        NVDR_HIP_TRY(hipMemcpyAsync(nodes_host, c->nodes, sizeof(uint32_t) * 8 * (c->n_tris - 1), hipMemcpyDeviceToHost, stream));
    if (tri_host)
        NVDR_HIP_TRY(hipMemcpyAsync(tri_host, c->tris, sizeof(float) * 12 * c->n_tris, hipMemcpyDeviceToHost, stream));
    NVDR_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

extern "C" int nvdr_ctx_set_allocator(nvdr_ctx *c, nvdr_alloc_fn alloc_fn, nvdr_free_fn free_fn, void *user)
{
    NVDR_REQUIRE(c, "nvdr_ctx_set_allocator: NULL ctx");
    NVDR_REQUIRE((alloc_fn == nullptr) == (free_fn == nullptr), "nvdr_ctx_set_allocator: give both functions or neither");
    NVDR_REQUIRE(c->n_allocs == 0, "nvdr_ctx_set_allocator: the context already owns %lld buffers from the previous allocator; "
                 "set the allocator right after nvdr_ctx_create", (long long)c->n_allocs);
    c->alloc_fn = alloc_fn;
    c->free_fn = free_fn;
    c->alloc_user = user;
    return 0;
}

extern "C" int nvdr_bvh_export_oct(nvdr_ctx *c, uint32_t *oct_host, float *tris8_host, int64_t *counts_host, void *stream_)
{
    NVDR_REQUIRE(c && c->n_tris > 0, "nvdr_bvh_export_oct: no BVH built");
    NVDR_REQUIRE(counts_host != nullptr, "nvdr_bvh_export_oct: counts_host is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    if (int rw = ctx_wait_built(c, stream)) return rw;
    unsigned ctl[OCT_CTL_WORDS];
    NVDR_HIP_TRY(hipMemcpyAsync(ctl, c->oct_ctl, sizeof(ctl), hipMemcpyDeviceToHost, stream));
    NVDR_HIP_TRY(hipStreamSynchronize(stream));
    counts_host[0] = ctl[OCT_CTL_ALLOC];    // oct nodes
    counts_host[1] = ctl[OCT_CTL_TRIS];     // triangles placed (== n_tris)
    counts_host[2] = ctl[OCT_CTL_DONE];     // nodes finished (== nodes)
    if (oct_host) NVDR_HIP_TRY(hipMemcpyAsync(oct_host, c->oct, sizeof(uint4) * 4 * (size_t)ctl[OCT_CTL_ALLOC], hipMemcpyDeviceToHost, stream));
    if (tris8_host) NVDR_HIP_TRY(hipMemcpyAsync(tris8_host, c->tris8, sizeof(float) * 12 * c->n_tris, hipMemcpyDeviceToHost, stream));
    NVDR_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

extern "C" int nvdr_trace_visibility(nvdr_ctx *c, const float *ro, const float *rd, int64_t n_rays, uint8_t *out_vis,
                                     unsigned long long *counters, void *stream_)
{
    NVDR_REQUIRE(c && c->n_tris > 0, "nvdr_trace_visibility: no BVH built");
    if (int r0 = ctx_check_overflow(c, "nvdr_trace_visibility")) return r0;
    if (n_rays <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rw = ctx_wait_built(c, stream)) return rw;
    const size_t lds = NVDR_STACK_LDS_BYTES(NVDR_QUERY_BLOCK);
    const unsigned grid = query_grid(c, n_rays);
    if (counters)
        trace_visibility_kernel<true><<<grid, NVDR_QUERY_BLOCK, lds, stream>>>(bvh_view(c), ro, rd, n_rays, out_vis, counters, c->spill);
    else
        trace_visibility_kernel<false><<<grid, NVDR_QUERY_BLOCK, lds, stream>>>(bvh_view(c), ro, rd, n_rays, out_vis, nullptr, c->spill);
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_trace_closest(nvdr_ctx *c, const float *ro, const float *rd, int64_t n_rays, float *out_t,
                                  int32_t *out_tri, float *out_uv, void *stream_)
{
    NVDR_REQUIRE(c && c->n_tris > 0, "nvdr_trace_closest: no BVH built");
    if (int r0 = ctx_check_overflow(c, "nvdr_trace_closest")) return r0;
    if (n_rays <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    if (int rw = ctx_wait_built(c, stream)) return rw;
    trace_closest_kernel<<<query_grid(c, n_rays), NVDR_QUERY_BLOCK, NVDR_STACK_LDS_BYTES(NVDR_QUERY_BLOCK), stream>>>(
        bvh_view(c), ro, rd, n_rays, out_t, out_tri, out_uv, c->spill);
    NVDR_LAUNCH_CHECK();
    return 0;
}
