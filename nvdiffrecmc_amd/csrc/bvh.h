// bvh.h -- device BVH layout, context object, and the wave-level traversal routines.
//
// Replaces the OptiX GAS + optixTrace of the reference (optixAccelBuild at
// render/optixutils/c_src/torch_bindings.cpp:97-110, shadow_test at envsampling/kernel.cu:101-118).
//
// Layout in HBM (all float4-aligned, see DESIGN.md "Data layout"):
//   nodes[n]  : 4 x float4 = 64 B per internal node, one cache line:
//       q0 = (lmin.x, lmin.y, lmin.z, lmax.x)   q1 = (lmax.y, lmax.z, rmin.x, rmin.y)
//       q2 = (rmin.z, rmax.x, rmax.y, rmax.z)   q3 = (left, right, hleft, hright) as int bits
//     the boxes of BOTH children live in the parent, so one 64-B fetch feeds two slab tests;
//     child >= 0: internal node index; child < 0: ~child = leaf slot (one triangle per leaf).
//   tris[k]   : 3 x float4 = 48 B per triangle in Morton order:
//       (v0.xyz, e1.x) (e1.yz, e2.xy) (e2.z, orig_index_bits, 0, 0)
#pragma once

#include "common.h"
#include "nvdr_raytri.h"

#define NVDR_STACK_DEPTH 32   // LDS traversal-stack entries per lane (one 4-B entry per level)
#define NVDR_STACK_DEPTH_DEEP 64

struct BvhDeviceInfo {
    int bounds[6];      // vertex AABB as order-preserving ints (min xyz, max xyz)
    int height;         // tree height in internal nodes
    int root;           // root node index (Karras numbering: 0)
    unsigned int pix_count; // env-shade: number of covered pixels appended to the work list
    unsigned int pad[7];
};

struct nvdr_ctx {
    int device = 0;
    int64_t cap_tris = 0;
    int64_t n_tris = 0;
    int64_t n_verts = 0;
    int height_host = -1;          // cached after nvdr_bvh_info_get
    float4 *nodes = nullptr;
    float4 *tris = nullptr;
    uint32_t *keys[2] = {nullptr, nullptr};
    uint32_t *vals[2] = {nullptr, nullptr};
    int *parent = nullptr;         // [2T]: parents of internal nodes [0,T-1) then of leaves [T, 2T)
    int *flags = nullptr;          // [T] arrival counters of the bottom-up pass
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    BvhDeviceInfo *dinfo = nullptr;
    // env-shade scratch
    int *pix_list = nullptr;
    int64_t pix_cap = 0;
};

struct BvhView {
    const float4 *nodes;
    const float4 *tris;
    int n_tris;
};

static inline BvhView bvh_view(const nvdr_ctx *c)
{
    BvhView v;
    v.nodes = c->nodes;
    v.tris = c->tris;
    v.n_tris = (int)c->n_tris;
    return v;
}

// ---------------------------------------------------------------------------------------------
// device side

__device__ __forceinline__ bool tri_any_hit(const float4 *__restrict__ tris, int slot, float ox, float oy, float oz,
                                            float dx, float dy, float dz)
{
    const float4 a = tris[3 * slot + 0], b = tris[3 * slot + 1], c = tris[3 * slot + 2];
    float t, u, v, det;
    return nvdr_ray_tri(ox, oy, oz, dx, dy, dz, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, &t, &u, &v, &det) != 0;
}

// slab test of one child box against the ray interval [0, tmax]; IEEE inf/NaN semantics make
// axis-parallel rays conservative (fminf/fmaxf drop NaNs).
__device__ __forceinline__ bool box_hit(float minx, float miny, float minz, float maxx, float maxy, float maxz,
                                        float ox, float oy, float oz, float ix, float iy, float iz, float tmax,
                                        float &tnear)
{
    const float x0 = (minx - ox) * ix, x1 = (maxx - ox) * ix;
    const float y0 = (miny - oy) * iy, y1 = (maxy - oy) * iy;
    const float z0 = (minz - oz) * iz, z1 = (maxz - oz) * iz;
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
    tnear = tn;
    return tn <= tf;
}

// Any-hit traversal for ONE lane's ray.  `stack` points at this lane's column of the wave's LDS
// stack: entry k lives at stack[k * 64] (bank = lane, conflict-free).  Returns true when the ray
// is OCCLUDED.  COUNT adds per-lane tallies of box and triangle tests.
template <bool COUNT>
__device__ __forceinline__ bool bvh_any_hit(const BvhView &bvh, float ox, float oy, float oz, float dx, float dy,
                                            float dz, int *stack, unsigned &n_box, unsigned &n_tri)
{
    if (bvh.n_tris == 1) {
        if (COUNT) n_tri++;
        return tri_any_hit(bvh.tris, 0, ox, oy, oz, dx, dy, dz);
    }
    const float ix = (1.0f / dx), iy = (1.0f / dy), iz = (1.0f / dz);
    int sp = 0;
    int cur = 0;
    while (true) {
        const float4 q0 = bvh.nodes[4 * cur + 0], q1 = bvh.nodes[4 * cur + 1];
        const float4 q2 = bvh.nodes[4 * cur + 2], q3 = bvh.nodes[4 * cur + 3];
        float tl, tr;
        bool hl = box_hit(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, ox, oy, oz, ix, iy, iz, NVDR_RAY_TMAX, tl);
        bool hr = box_hit(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, ox, oy, oz, ix, iy, iz, NVDR_RAY_TMAX, tr);
        const int cl = __float_as_int(q3.x), cr = __float_as_int(q3.y);
        if (COUNT) n_box += 2;
        if (hl && cl < 0) {
            if (COUNT) n_tri++;
            if (tri_any_hit(bvh.tris, ~cl, ox, oy, oz, dx, dy, dz)) return true;
            hl = false;
        }
        if (hr && cr < 0) {
            if (COUNT) n_tri++;
            if (tri_any_hit(bvh.tris, ~cr, ox, oy, oz, dx, dy, dz)) return true;
            hr = false;
        }
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
            if (sp == 0) return false;
            sp--;
            cur = stack[sp * 64];
        }
    }
}
