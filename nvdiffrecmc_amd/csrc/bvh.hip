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
//      the child boxes are written INTO the parent's 64-B record (layout: bvh.h)
// Buffers live in the context and are reused across iterations; rebuild == 0 re-runs only 1 + 5
// (OPTIX_BUILD_OPERATION_UPDATE).  Nothing synchronises the host.
#include "bvh.h"

#include <rocprim/rocprim.hpp>

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

__global__ void bvh_init_info_kernel(BvhDeviceInfo *info)
{
    if (threadIdx.x == 0) {
        info->bounds[0] = info->bounds[1] = info->bounds[2] = 0x7fffffff;
        info->bounds[3] = info->bounds[4] = info->bounds[5] = (int)0x80000000;
        info->height = 0;
        info->root = 0;
    }
}

__global__ void bvh_bounds_kernel(const float *__restrict__ verts, int64_t n_verts, BvhDeviceInfo *info)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_verts; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = verts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o >= 1; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&info->bounds[a], float_to_ordered(mn[a]));
            atomicMax(&info->bounds[3 + a], float_to_ordered(mx[a]));
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

__global__ void bvh_hierarchy_kernel(const uint32_t *__restrict__ keys, int n, float4 *__restrict__ nodes,
                                     int *__restrict__ parent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
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
    const int left = (lo == gamma) ? ~gamma : gamma;
    const int right = (hi == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
    nodes[4 * i + 3] = make_float4(__int_as_float(left), __int_as_float(right), __int_as_float(0), __int_as_float(0));
    if (left < 0) parent[n + gamma] = i; else parent[left] = i;
    if (right < 0) parent[n + gamma + 1] = i; else parent[right] = i;
    if (i == 0) parent[0] = -1;
}

// One thread per leaf: write the triangle record, then climb.  Inter-workgroup hand-off of the
// sibling box follows the agent-scope release/acquire recipe (cdna_hip_programming.md G16):
// plain stores -> release fence -> drained vmcnt -> device-scope atomic; the second arriver does
// one acquire fence and then plain loads.
__global__ void bvh_fit_kernel(const float *__restrict__ verts, const int32_t *__restrict__ tris,
                               const uint32_t *__restrict__ order, int n, float4 *__restrict__ tri_rec,
                               float4 *nodes, const int *__restrict__ parent, int *flags, BvhDeviceInfo *info)
{
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

    // conservative padding: the slab test must never cull a triangle the fp32 Moeller-Trumbore
    // predicate would accept (its acceptance band is a few ulp of the scene scale wide)
    float scale = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = ordered_to_float(info->bounds[a]), hi = ordered_to_float(info->bounds[3 + a]);
        scale = fmaxf(scale, fmaxf(hi - lo, fmaxf(fabsf(lo), fabsf(hi))));
    }
    const float pad = 1e-5f * scale;
    float mnx = fminf(ax, fminf(bx, cx)) - pad, mny = fminf(ay, fminf(by, cy)) - pad, mnz = fminf(az, fminf(bz, cz)) - pad;
    float mxx = fmaxf(ax, fmaxf(bx, cx)) + pad, mxy = fmaxf(ay, fmaxf(by, cy)) + pad, mxz = fmaxf(az, fmaxf(bz, cz)) + pad;
    int height = 0;
    int me = ~k;
    int node = parent[n + k];
    float *nf = (float *)nodes;
    while (true) {
        float *rec = nf + 16 * (int64_t)node;
        const int cl = __float_as_int(rec[12]);
        const int slot = (cl == me) ? 0 : 1;
        float *dst = rec + 6 * slot;
        dst[0] = mnx; dst[1] = mny; dst[2] = mnz; dst[3] = mxx; dst[4] = mxy; dst[5] = mxz;
        rec[14 + slot] = __int_as_float(height);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = atomicAdd(&flags[node], 1);
        if (old == 0) return; // first arriver: the sibling subtree finishes this node
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float *src = rec + 6 * (1 - slot);
        mnx = fminf(mnx, src[0]); mny = fminf(mny, src[1]); mnz = fminf(mnz, src[2]);
        mxx = fmaxf(mxx, src[3]); mxy = fmaxf(mxy, src[4]); mxz = fmaxf(mxz, src[5]);
        height = 1 + max(height, __float_as_int(rec[14 + (1 - slot)]));
        if (node == 0) {
            info->height = height;
            return;
        }
        me = node;
        node = parent[node];
    }
}

// ---------------------------------------------------------------------------------------------
// ray-query hooks

template <bool COUNT>
__global__ void __launch_bounds__(256) trace_visibility_kernel(BvhView bvh, const float *__restrict__ ro,
                                                                const float *__restrict__ rd, int64_t n_rays,
                                                                uint8_t *__restrict__ out, unsigned long long *counters,
                                                                int stack_depth)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *stack = smem + wave * stack_depth * 64 + lane;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nb = 0, nt = 0;
    if (r < n_rays) {
        const bool occ = bvh_any_hit<COUNT>(bvh, ro[3 * r], ro[3 * r + 1], ro[3 * r + 2], rd[3 * r], rd[3 * r + 1],
                                            rd[3 * r + 2], stack, nb, nt);
        out[r] = occ ? 0 : 1;
    }
    if (COUNT) {
        for (int o = 32; o >= 1; o >>= 1) {
            nb += __shfl_xor(nb, o);
            nt += __shfl_xor(nt, o);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], (unsigned long long)nb);
            atomicAdd(&counters[1], (unsigned long long)nt);
        }
    }
}

// closest hit (ordered traversal, shrinking tmax); used by the G-buffer producer
__global__ void __launch_bounds__(256) trace_closest_kernel(BvhView bvh, const float *__restrict__ ro,
                                                             const float *__restrict__ rd, int64_t n_rays,
                                                             float *__restrict__ out_t, int32_t *__restrict__ out_tri,
                                                             float *__restrict__ out_uv, int stack_depth)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *stack = smem + wave * stack_depth * 64 + lane;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float ox = ro[3 * r], oy = ro[3 * r + 1], oz = ro[3 * r + 2];
    const float dx = rd[3 * r], dy = rd[3 * r + 1], dz = rd[3 * r + 2];
    float best_t = NVDR_RAY_TMAX, best_u = 0.0f, best_v = 0.0f;
    int best = -1;
    auto test_leaf = [&](int slot) {
        const float4 a = bvh.tris[3 * slot + 0], b = bvh.tris[3 * slot + 1], c = bvh.tris[3 * slot + 2];
        float t, u, v, det;
        if (nvdr_ray_tri(ox, oy, oz, dx, dy, dz, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, &t, &u, &v, &det)) {
            const float tt = (t / det);
            if (tt < best_t) {
                best_t = tt;
                best_u = (u / det);
                best_v = (v / det);
                best = __float_as_int(c.y);
            }
        }
    };
    if (bvh.n_tris == 1) {
        test_leaf(0);
    } else {
        const float ix = (1.0f / dx), iy = (1.0f / dy), iz = (1.0f / dz);
        int sp = 0, cur = 0;
        while (true) {
            const float4 q0 = bvh.nodes[4 * cur + 0], q1 = bvh.nodes[4 * cur + 1];
            const float4 q2 = bvh.nodes[4 * cur + 2], q3 = bvh.nodes[4 * cur + 3];
            float tl, tr;
            bool hl = box_hit(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, best_t, tl);
            bool hr = box_hit(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, best_t, tr);
            const int cl = __float_as_int(q3.x), cr = __float_as_int(q3.y);
            if (hl && cl < 0) { test_leaf(~cl); hl = false; }
            if (hr && cr < 0) { test_leaf(~cr); hr = false; }
            if (hl && hr) {
                const bool left_first = tl <= tr;
                stack[sp * 64] = left_first ? cr : cl;
                sp++;
                cur = left_first ? cl : cr;
            } else if (hl) {
                cur = cl;
            } else if (hr) {
                cur = cr;
            } else {
                if (sp == 0) break;
                sp--;
                cur = stack[sp * 64];
            }
        }
    }
    out_t[r] = best >= 0 ? best_t : -1.0f;
    out_tri[r] = best;
    out_uv[2 * r] = best_u;
    out_uv[2 * r + 1] = best_v;
}

// ---------------------------------------------------------------------------------------------
// host side

static int ctx_free_bvh(nvdr_ctx *c)
{
    hipFree(c->nodes); hipFree(c->tris);
    hipFree(c->keys[0]); hipFree(c->keys[1]); hipFree(c->vals[0]); hipFree(c->vals[1]);
    hipFree(c->parent); hipFree(c->flags); hipFree(c->sort_tmp);
    c->nodes = c->tris = nullptr;
    c->keys[0] = c->keys[1] = c->vals[0] = c->vals[1] = nullptr;
    c->parent = c->flags = nullptr;
    c->sort_tmp = nullptr;
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
    *out = c;
    return 0;
}

extern "C" int nvdr_ctx_destroy(nvdr_ctx *c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    ctx_free_bvh(c);
    hipFree(c->dinfo);
    hipFree(c->pix_list);
    delete c;
    return 0;
}

static int ctx_reserve(nvdr_ctx *c, int64_t n_tris)
{
    if (n_tris <= c->cap_tris) return 0;
    // the reference frees and re-allocates the GAS on every build (torch_bindings.cpp:52,84-95,114);
    // here the buffers only ever grow (by 1.5x) and are reused across iterations
    NVDR_HIP_TRY(hipDeviceSynchronize());
    ctx_free_bvh(c);
    const int64_t cap = n_tris + n_tris / 2 + 64;
    NVDR_HIP_TRY(hipMalloc((void **)&c->nodes, sizeof(float4) * 4 * cap));
    NVDR_HIP_TRY(hipMalloc((void **)&c->tris, sizeof(float4) * 3 * cap));
    for (int i = 0; i < 2; ++i) {
        NVDR_HIP_TRY(hipMalloc((void **)&c->keys[i], sizeof(uint32_t) * cap));
        NVDR_HIP_TRY(hipMalloc((void **)&c->vals[i], sizeof(uint32_t) * cap));
    }
    NVDR_HIP_TRY(hipMalloc((void **)&c->parent, sizeof(int) * 2 * cap));
    NVDR_HIP_TRY(hipMalloc((void **)&c->flags, sizeof(int) * cap));
    size_t bytes = 0;
    NVDR_HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, c->keys[0], c->keys[1], c->vals[0], c->vals[1], (size_t)cap, 0, 30));
    NVDR_HIP_TRY(hipMalloc(&c->sort_tmp, bytes + 256));
    c->sort_tmp_bytes = bytes + 256;
    c->cap_tris = cap;
    return 0;
}

extern "C" int nvdr_bvh_build(nvdr_ctx *c, const float *verts, int64_t n_verts, const int32_t *tris, int64_t n_tris,
                              int rebuild, void *stream_)
{
    NVDR_REQUIRE(c != nullptr, "nvdr_bvh_build: ctx is NULL");
    // same message as the Python asserts of the reference (render/optixutils/ops.py:131-132)
    NVDR_REQUIRE(n_tris > 0 && n_verts > 0, "Got empty training triangle mesh (unrecoverable discontinuity)");
    NVDR_REQUIRE(n_tris < (1ll << 30), "nvdr_bvh_build: too many triangles (%lld)", (long long)n_tris);
    NVDR_REQUIRE(verts && tris, "nvdr_bvh_build: NULL geometry pointer");
    hipStream_t stream = (hipStream_t)stream_;
    NVDR_HIP_TRY(hipSetDevice(c->device));
    if (rebuild == 0) {
        NVDR_REQUIRE(c->n_tris == n_tris && c->n_verts == n_verts,
                     "nvdr_bvh_build: refit (rebuild=0) needs the topology of the last full build "
                     "(%lld tris / %lld verts, got %lld / %lld)",
                     (long long)c->n_tris, (long long)c->n_verts, (long long)n_tris, (long long)n_verts);
    } else {
        int r = ctx_reserve(c, n_tris);
        if (r) return r;
    }
    const int n = (int)n_tris;
    bvh_init_info_kernel<<<1, 64, 0, stream>>>(c->dinfo);
    bvh_bounds_kernel<<<min(div_up(n_verts, 256), 1024u), 256, 0, stream>>>(verts, n_verts, c->dinfo);
    if (rebuild != 0) {
        bvh_morton_kernel<<<div_up(n, 256), 256, 0, stream>>>(verts, tris, n, c->dinfo, c->keys[0], c->vals[0]);
        size_t bytes = c->sort_tmp_bytes;
        NVDR_HIP_TRY(rocprim::radix_sort_pairs(c->sort_tmp, bytes, c->keys[0], c->keys[1], c->vals[0], c->vals[1],
                                               (size_t)n, 0, 30, stream));
        if (n > 1)
            bvh_hierarchy_kernel<<<div_up(n - 1, 256), 256, 0, stream>>>(c->keys[1], n, c->nodes, c->parent);
    }
    NVDR_HIP_TRY(hipMemsetAsync(c->flags, 0, sizeof(int) * n, stream));
    bvh_fit_kernel<<<div_up(n, 256), 256, 0, stream>>>(verts, tris, c->vals[1], n, c->tris, c->nodes, c->parent,
                                                        c->flags, c->dinfo);
    NVDR_LAUNCH_CHECK();
    c->n_tris = n_tris;
    c->n_verts = n_verts;
    c->height_host = -1;
    return 0;
}

// The traversal kernels size their LDS stack from the tree height, which is produced on the
// device; the first query after a build reads it back (one 4-byte copy on the query's stream).
int bvh_stack_depth(nvdr_ctx *c, hipStream_t stream, int *depth)
{
    if (c->height_host < 0) {
        int h = 0;
        NVDR_HIP_TRY(hipMemcpyAsync(&h, &c->dinfo->height, sizeof(int), hipMemcpyDeviceToHost, stream));
        NVDR_HIP_TRY(hipStreamSynchronize(stream));
        c->height_host = h;
    }
    int d = ((c->height_host + 1 + 7) / 8) * 8;
    if (d < 8) d = 8;
    NVDR_REQUIRE(d <= 96, "BVH too deep for the LDS traversal stack (height %d)", c->height_host);
    *depth = d;
    return 0;
}

extern "C" int nvdr_bvh_info_get(nvdr_ctx *c, nvdr_bvh_info *out, void *stream_)
{
    NVDR_REQUIRE(c && out, "nvdr_bvh_info_get: NULL argument");
    NVDR_REQUIRE(c->n_tris > 0, "nvdr_bvh_info_get: no BVH built");
    hipStream_t stream = (hipStream_t)stream_;
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
    }
    c->height_host = h.height;
    return 0;
}

extern "C" int nvdr_bvh_export(nvdr_ctx *c, float *nodes_host, float *tri_host, void *stream_)
{
    NVDR_REQUIRE(c && c->n_tris > 0, "nvdr_bvh_export: no BVH built");
    hipStream_t stream = (hipStream_t)stream_;
    if (nodes_host && c->n_tris > 1)
        NVDR_HIP_TRY(hipMemcpyAsync(nodes_host, c->nodes, sizeof(float) * 16 * (c->n_tris - 1), hipMemcpyDeviceToHost, stream));
    if (tri_host)
        NVDR_HIP_TRY(hipMemcpyAsync(tri_host, c->tris, sizeof(float) * 12 * c->n_tris, hipMemcpyDeviceToHost, stream));
    NVDR_HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

extern "C" int nvdr_trace_visibility(nvdr_ctx *c, const float *ro, const float *rd, int64_t n_rays, uint8_t *out_vis,
                                     unsigned long long *counters, void *stream_)
{
    NVDR_REQUIRE(c && c->n_tris > 0, "nvdr_trace_visibility: no BVH built");
    if (n_rays <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    int depth;
    int r = bvh_stack_depth(c, stream, &depth);
    if (r) return r;
    const size_t lds = (size_t)4 * depth * 64 * sizeof(int);
    if (counters)
        trace_visibility_kernel<true><<<div_up(n_rays, 256), 256, lds, stream>>>(bvh_view(c), ro, rd, n_rays, out_vis, counters, depth);
    else
        trace_visibility_kernel<false><<<div_up(n_rays, 256), 256, lds, stream>>>(bvh_view(c), ro, rd, n_rays, out_vis, nullptr, depth);
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_trace_closest(nvdr_ctx *c, const float *ro, const float *rd, int64_t n_rays, float *out_t,
                                  int32_t *out_tri, float *out_uv, void *stream_)
{
    NVDR_REQUIRE(c && c->n_tris > 0, "nvdr_trace_closest: no BVH built");
    if (n_rays <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    int depth;
    int r = bvh_stack_depth(c, stream, &depth);
    if (r) return r;
    const size_t lds = (size_t)4 * depth * 64 * sizeof(int);
    trace_closest_kernel<<<div_up(n_rays, 256), 256, lds, stream>>>(bvh_view(c), ro, rd, n_rays, out_t, out_tri, out_uv, depth);
    NVDR_LAUNCH_CHECK();
    return 0;
}
