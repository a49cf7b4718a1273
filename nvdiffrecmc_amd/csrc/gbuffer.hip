// gbuffer.hip -- G-buffer producer: primary visibility + attribute interpolation in ONE kernel.
//
// Replaces, for the benchmark path, what render_layer gets from nvdiffrast (render/render.py:170-234):
//   rast, rast_db = dr.rasterize(...)                          render.py:279 (CUDA/OpenGL rasteriser: does not exist on ROCm)
//   gb_pos, gb_geometric_normal, gb_normal, gb_tangent          render.py:208-222  (interpolate of v_pos / face normals / v_nrm / v_tng)
//   gb_texc, gb_texc_deriv                                      render.py:225-226
//   gb_depth = (z/w, |dz|)                                      render.py:228-234
// Here a pixel's primary ray is traced through the eight-wide tree the shadow rays use (closest hit, oct_closest_hit below), and the hit is turned into
// the same per-pixel records nvdiffrast produces:
//   rast    = (u, v, z/w, triangle_id + 1)     u, v = perspective-correct barycentrics of vertex 0 and 1 (nvdiffrast's
//             convention: attr = u a0 + v a1 + (1-u-v) a2); 0 for background pixels
//   rast_db = (du/dX, du/dY, dv/dX, dv/dY)     per PIXEL, from the triangle's clip-space plane equations (analytic, like
//             the rasteriser), not finite differences
// Attribute interpolation, the face normal, and the depth pair follow render.py line by line -- including the reference's
// quirk that `clip_pos_deriv[..., 2:3]` / `[..., 3:4]` (render.py:232) pick d(clip.y)/dX and d(clip.y)/dY out of nvdiffrast's
// interleaved (dA/dX, dA/dY) layout rather than the z and w derivatives: the denoiser's depth weight sees the reference's numbers.
// Not differentiable (the reference gets geometry gradients from dr.antialias + interpolate; out of scope).
#include "trace_kernel.h"
#include "bsdf_device.h"

// Closest hit through the EIGHT-WIDE tree the shadow rays walk (round 4; rounds 2-3 walked the binary tree here: ~35 dependent
// 32-byte fetches per primary ray on 684 k triangles, 0.55 ms per 800^2 view).  One ray per lane, the node step of
// trace_kernel.h (64-byte node, eight children at 17 VALU each) with the current best distance as the far limit; a leaf's
// triangle is tested in place -- a closest hit needs the distance at once, nothing is deferred.  Children are taken lowest index
// first (they are ordered by surface area, not by distance): the walk is not front-to-back, the shrinking limit prunes what lies
// behind a hit.  Same predicate (include/nvdr_raytri.h) and the same conservative boxes as the binary walk of nvdr_trace_closest:
// the same triangle unless two triangles are hit at exactly the same distance.
__device__ __forceinline__ int oct_closest_hit(const BvhView &bvh, float ox, float oy, float oz, float dx, float dy, float dz,
                                               const OctStack &stack, float &best_t, float &best_u, float &best_v)
{
    best_t = NVDR_RAY_TMAX;
    best_u = 0.0f;
    best_v = 0.0f;
    int best = -1;
    const uint4 *__restrict__ oct = bvh.oct;
    const float4 *__restrict__ tris8 = bvh.tris8;
    const BvhDeviceInfo *__restrict__ info = bvh.info;
    OctRay g;
    g.ix = fminf(fmaxf(__builtin_amdgcn_rcpf(dx * info->g_scale[0]), -1.0e30f), 1.0e30f);
    g.iy = fminf(fmaxf(__builtin_amdgcn_rcpf(dy * info->g_scale[1]), -1.0e30f), 1.0e30f);
    g.iz = fminf(fmaxf(__builtin_amdgcn_rcpf(dz * info->g_scale[2]), -1.0e30f), 1.0e30f);
    g.nx = -((ox - info->g_lo[0]) * info->g_scale[0] + 2.0f) * g.ix;
    g.ny = -((oy - info->g_lo[1]) * info->g_scale[1] + 2.0f) * g.iy;
    g.nz = -((oz - info->g_lo[2]) * info->g_scale[2] + 2.0f) * g.iz;
    unsigned gbase = 0u, gbits = 1u;        // the root is "child 0 of group 0"
    int sp = 0;
    for (unsigned iters = 0; iters < (1u << 24); ++iters) {
        if (gbits == 0u) {
            if (sp == 0) break;
            sp--;
            const unsigned long long top = stack.pop(sp);
            gbase = (unsigned)top;
            gbits = (unsigned)(top >> 32);
        }
        const int k = __builtin_ctz(gbits);
        gbits &= gbits - 1u;
        const uint4 *nd = oct + 4 * (int64_t)(gbase + (unsigned)k);
        const uint4 h = nd[0], p1 = nd[1], p2 = nd[2], p3 = nd[3];
        const float ax = __builtin_ldexpf(g.ix, (int)((h.y >> 16) & 15u)), bx = fmaf((float)(h.x & 0xffffu), g.ix, g.nx);
        const float ay = __builtin_ldexpf(g.iy, (int)((h.y >> 20) & 15u)), by = fmaf((float)(h.x >> 16), g.iy, g.ny);
        const float az = __builtin_ldexpf(g.iz, (int)((h.y >> 24) & 15u)), bz = fmaf((float)(h.y & 0xffffu), g.iz, g.nz);
        const bool sx = g.ix < 0.0f, sy = g.iy < 0.0f, sz = g.iz < 0.0f;
        const unsigned nx0 = sx ? p2.z : p1.x, nx1 = sx ? p2.w : p1.y, fx0 = sx ? p1.x : p2.z, fx1 = sx ? p1.y : p2.w;
        const unsigned ny0 = sy ? p3.x : p1.z, ny1 = sy ? p3.y : p1.w, fy0 = sy ? p1.z : p3.x, fy1 = sy ? p1.w : p3.y;
        const unsigned nz0 = sz ? p3.z : p2.x, nz1 = sz ? p3.w : p2.y, fz0 = sz ? p2.x : p3.z, fz1 = sz ? p2.y : p3.w;
        unsigned miss = 0u;
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            const unsigned wnx = j < 4 ? nx0 : nx1, wny = j < 4 ? ny0 : ny1, wnz = j < 4 ? nz0 : nz1;
            const unsigned wfx = j < 4 ? fx0 : fx1, wfy = j < 4 ? fy0 : fy1, wfz = j < 4 ? fz0 : fz1;
            const float tnx = fmaf(ubyte_f32(wnx, j & 3), ax, bx), tfx = fmaf(ubyte_f32(wfx, j & 3), ax, bx);
            const float tny = fmaf(ubyte_f32(wny, j & 3), ay, by), tfy = fmaf(ubyte_f32(wfy, j & 3), ay, by);
            const float tnz = fmaf(ubyte_f32(wnz, j & 3), az, bz), tfz = fmaf(ubyte_f32(wfz, j & 3), az, bz);
            const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
            const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best_t));        // nothing behind the best hit so far
            miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn), 31u);
        }
        const unsigned n_int = h.z >> 28, n_leaf = h.w >> 28;
        const unsigned hits = ~miss & ((1u << (n_int + n_leaf)) - 1u);
        const unsigned hi = hits & ((1u << n_int) - 1u);
        unsigned leaf_bits = hits >> n_int;
        const unsigned leaf_base = h.w & (NVDR_OCT_MAX_INDEX - 1u);
        while (leaf_bits) {
            const int j = __builtin_ctz(leaf_bits);
            leaf_bits &= leaf_bits - 1u;
            const int slot = (int)(leaf_base + (unsigned)j);
            const float4 a = tris8[3 * slot + 0], b = tris8[3 * slot + 1], c = tris8[3 * slot + 2];
            float t, u, v, det;
            if (nvdr_ray_tri(ox, oy, oz, dx, dy, dz, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, &t, &u, &v, &det)) {
                const float tt = t / det;
                if (tt < best_t) { best_t = tt; best_u = u / det; best_v = v / det; best = __float_as_int(c.y); }
            }
        }
        if (hi != 0u) {
            if (gbits != 0u) sp = stack.push(sp, pack2(gbase, gbits));
            gbase = h.z & (NVDR_OCT_MAX_INDEX - 1u);
            gbits = hi;
        }
    }
    return best;
}

struct GbufferParams {
    const float *v_pos; const int *t_pos;
    const float *v_nrm; const int *t_nrm;
    const float *v_tng; const int *t_tng;
    const float *v_tex; const int *t_tex;
    const float *mvp;      // [N,4,4] row-major
    const float *cam;      // [N,4,3]: eye, U, V, W  (ray through NDC (X, Y): normalize(X U + Y V + W))
    int N, H, W;
    float *rast, *rast_db, *gb_pos, *gb_gnrm, *gb_nrm, *gb_tng, *gb_texc, *gb_texc_db, *gb_depth;
};

__device__ __forceinline__ F3 load3(const float *p, int i) { return f3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ void store3(float *p, int64_t i, F3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
// nvdiffrast's interpolation: u a0 + v a1 + (1 - u - v) a2
__device__ __forceinline__ F3 interp3(F3 a0, F3 a1, F3 a2, float u, float v) { return (a0 * u + a1 * v) + a2 * (1.0f - u - v); }

// Work dealing (round 6).  Rounds 3-5 ran a grid-stride loop over the linear pixel index: a wavefront took a 64 x 1 strip, and a launch of
// more pixels than 2048 workgroups x 256 (one 800^2 view: 2 500) left a fifth of the workgroups a second strip after the others had gone --
// one view cost 2.2 x its share of the 8-view launch.  Now the unit is an 8 x 8 TILE per wavefront (neighbouring primary rays walk
// the same nodes; a 64 x 1 strip when the extent is no multiple of eight) and the wavefronts CLAIM units: class q = unit % 64 behind its own
// counter (one per 128-byte line: same-address atomics retire every ~12 ns), wavefront w serves class w % 64, a wavefront that drew
// background tiles simply claims more.  The counters live in lines 128..255 of the context's counter block, zero between launches: the last
// wavefront of a class to leave resets its pair.
#define NVDR_GB_CLASSES 64u
__global__ void __launch_bounds__(NVDR_QUERY_BLOCK) gbuffer_kernel(BvhView bvh, GbufferParams p, int *spill, unsigned *queues, int mode)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    OctStack stack;             // per lane: NVDR_OSTACK_LDS (group, bits) entries in LDS, deeper ones in the context's spill columns
    stack.lds = (lds_pair_t *)((char *)smem + (threadIdx.x >> 6) * NVDR_OSTACK_LDS * 64 * 8) + (threadIdx.x & 63);
    stack.gstride = blockDim.x;
    stack.smax = bvh.oct_stack_max;
    stack.ovf = bvh.overflow;
    stack.glb = (glb_pair_t *)spill + (int64_t)blockIdx.x * blockDim.x * max(bvh.oct_stack_max - NVDR_OSTACK_LDS, 0) + threadIdx.x;
    const int64_t total = (int64_t)p.N * p.H * p.W;
    // mode (NVDR_GB_MODE, A/B): bit 0 tiles, bit 1 claimed units (else dealt round-robin).  Tiles only for launches of several units per
    // wavefront: with ~1 unit each (one 800^2 view) a covered tile is all deep walks and a background tile none, and the launch lasts as long as
    // its dearest tiles -- strips mix the two (session 10: one view 0.250 ms in strips, 0.314 in tiles; eight views 0.089 / 0.085 per view; on the
    // marching-tets mesh tiles win both: 0.247 against 0.326, 0.145 against 0.233)
    const unsigned n_waves_all = gridDim.x * (blockDim.x >> 6);
    const bool tiled = (mode & 1) && (p.W & 7) == 0 && (p.H & 7) == 0 && (mode & 4 ? true : ((total + 63) >> 6) >= 4 * (int64_t)n_waves_all);
    const unsigned tiles_x = (unsigned)p.W >> 3, tiles_per_view = tiles_x * ((unsigned)p.H >> 3);
    const unsigned n_units = (unsigned)((total + 63) >> 6);
    const unsigned lane = threadIdx.x & 63u, wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
    const unsigned cls = wid % NVDR_GB_CLASSES;
    unsigned *claim = queues + (128u + cls) * 32u, *gone = queues + (192u + cls) * 32u;
    for (unsigned round = 0u;; ++round) {
        unsigned unit;
        if (mode & 2) {
            unsigned j = 0u;
            if (lane == 0u) j = atomicAdd(claim, 1u);
            j = (unsigned)__builtin_amdgcn_readfirstlane((int)j);
            unit = cls + NVDR_GB_CLASSES * j;
        } else {
            unit = wid + round * n_waves;
        }
        if (unit >= n_units) break;
        int64_t i;
        int x, y, z;
        if (tiled) {
            const unsigned t = unit % tiles_per_view;
            z = (int)(unit / tiles_per_view);
            x = (int)(((t % tiles_x) << 3) + (lane & 7u));
            y = (int)(((t / tiles_x) << 3) + (lane >> 3));
            i = ((int64_t)z * p.H + y) * p.W + x;
        } else {
            i = (int64_t)unit * 64 + lane;
            if (i >= total) continue;
            x = (int)(i % p.W); y = (int)((i / p.W) % p.H); z = (int)(i / ((int64_t)p.W * p.H));
        }
        const float X = ((float)x + 0.5f) / (float)p.W * 2.0f - 1.0f;      // NDC of the pixel centre; row 0 = Y -1 (rasteriser layout)
        const float Y = ((float)y + 0.5f) / (float)p.H * 2.0f - 1.0f;
        const float *cam = p.cam + 12 * z;
        const F3 eye = f3(cam[0], cam[1], cam[2]);
        F3 d = (f3(cam[3], cam[4], cam[5]) * X + f3(cam[6], cam[7], cam[8]) * Y) + f3(cam[9], cam[10], cam[11]);
        d = d * (1.0f / sqrtf(dot3(d, d)));
        float t, bu, bv;
        const int tri = bvh.n_tris > 0 ? oct_closest_hit(bvh, eye.x, eye.y, eye.z, d.x, d.y, d.z, stack, t, bu, bv) : -1;

        float4 rast = make_float4(0.f, 0.f, 0.f, 0.f), rdb = make_float4(0.f, 0.f, 0.f, 0.f);
        F3 pos = f3(0.f), gn = f3(0.f), nrm = f3(0.f), tng = f3(0.f);
        // background: every interpolated attribute is 0, so render.py:230 gives z0 = clamp(0, eps) / clamp(0, eps) = 1 and |dz| = 0
        float tc0 = 0.f, tc1 = 0.f, tdb[4] = {0.f, 0.f, 0.f, 0.f}, z0 = 1.f, zg = 0.f;
        if (tri >= 0) {
            const int i0 = p.t_pos[3 * tri], i1 = p.t_pos[3 * tri + 1], i2 = p.t_pos[3 * tri + 2];
            const F3 p0 = load3(p.v_pos, i0), p1 = load3(p.v_pos, i1), p2 = load3(p.v_pos, i2);
            const float u = 1.0f - bu - bv, v = bu;        // Moeller-Trumbore weights of (v1, v2) -> nvdiffrast's (v0, v1)
            // clip-space vertices (x, y, z, w) = mvp * (p, 1)
            const float *M = p.mvp + 16 * z;
            float cx[3], cy[3], cz[3], cw[3];
            const F3 pv[3] = {p0, p1, p2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                cx[k] = M[0] * pv[k].x + M[1] * pv[k].y + M[2] * pv[k].z + M[3];
                cy[k] = M[4] * pv[k].x + M[5] * pv[k].y + M[6] * pv[k].z + M[7];
                cz[k] = M[8] * pv[k].x + M[9] * pv[k].y + M[10] * pv[k].z + M[11];
                cw[k] = M[12] * pv[k].x + M[13] * pv[k].y + M[14] * pv[k].z + M[15];
            }
            // Plane equations of the perspective-correct barycentrics: beta_k(X, Y) = n_k / (n_0 + n_1 + n_2) with
            // n_k = A_k X + B_k Y + C_k, (A_k, B_k, C_k) = row k of adj([cx; cy; cw]).  d(beta_k)/dX = (A_k - beta_k sum(A)) / s.
            const float A0 = cy[1] * cw[2] - cy[2] * cw[1], B0 = cx[2] * cw[1] - cx[1] * cw[2], C0 = cx[1] * cy[2] - cx[2] * cy[1];
            const float A1 = cy[2] * cw[0] - cy[0] * cw[2], B1 = cx[0] * cw[2] - cx[2] * cw[0], C1 = cx[2] * cy[0] - cx[0] * cy[2];
            const float A2 = cy[0] * cw[1] - cy[1] * cw[0], B2 = cx[1] * cw[0] - cx[0] * cw[1], C2 = cx[0] * cy[1] - cx[1] * cy[0];
            const float s = (A0 + A1 + A2) * X + (B0 + B1 + B2) * Y + (C0 + C1 + C2);
            const float is = fabsf(s) > 1e-30f ? 1.0f / s : 0.0f;
            const float px = 2.0f / (float)p.W, py = 2.0f / (float)p.H;             // NDC per pixel
            const float dudx = (A0 - u * (A0 + A1 + A2)) * is * px, dudy = (B0 - u * (B0 + B1 + B2)) * is * py;
            const float dvdx = (A1 - v * (A0 + A1 + A2)) * is * px, dvdy = (B1 - v * (B0 + B1 + B2)) * is * py;
            const float w2 = 1.0f - u - v;
            const float zc = u * cz[0] + v * cz[1] + w2 * cz[2], wc = u * cw[0] + v * cw[1] + w2 * cw[2];
            rast = make_float4(u, v, zc / wc, (float)(tri + 1));
            rdb = make_float4(dudx, dudy, dvdx, dvdy);
            // render.py:208-222
            pos = interp3(p0, p1, p2, u, v);
            const F3 fn = safe_normalize(cross3(p1 - p0, p2 - p0));
            gn = interp3(fn, fn, fn, u, v);
            nrm = interp3(load3(p.v_nrm, p.t_nrm[3 * tri]), load3(p.v_nrm, p.t_nrm[3 * tri + 1]), load3(p.v_nrm, p.t_nrm[3 * tri + 2]), u, v);
            tng = interp3(load3(p.v_tng, p.t_tng[3 * tri]), load3(p.v_tng, p.t_tng[3 * tri + 1]), load3(p.v_tng, p.t_tng[3 * tri + 2]), u, v);
            // render.py:225-226: texture coordinate + its image-space derivatives (ds/dX, ds/dY, dt/dX, dt/dY)
            const int j0 = p.t_tex[3 * tri], j1 = p.t_tex[3 * tri + 1], j2 = p.t_tex[3 * tri + 2];
            const float s0 = p.v_tex[2 * j0], t0 = p.v_tex[2 * j0 + 1], s1 = p.v_tex[2 * j1], t1 = p.v_tex[2 * j1 + 1];
            const float s2 = p.v_tex[2 * j2], t2 = p.v_tex[2 * j2 + 1];
            tc0 = (s0 * u + s1 * v) + s2 * w2;
            tc1 = (t0 * u + t1 * v) + t2 * w2;
            tdb[0] = dudx * (s0 - s2) + dvdx * (s1 - s2); tdb[1] = dudy * (s0 - s2) + dvdy * (s1 - s2);
            tdb[2] = dudx * (t0 - t2) + dvdx * (t1 - t2); tdb[3] = dudy * (t0 - t2) + dvdy * (t1 - t2);
            // render.py:228-234.  clip_pos_deriv is [x/dX, x/dY, y/dX, y/dY, z/dX, ...]: channels 2 and 3 are the Y-clip derivatives
            const float eps = 0.00001f;
            const float dyx = dudx * (cy[0] - cy[2]) + dvdx * (cy[1] - cy[2]);
            const float dyy = dudy * (cy[0] - cy[2]) + dvdy * (cy[1] - cy[2]);
            z0 = fmaxf(zc, eps) / fmaxf(wc, eps);
            const float z1 = fmaxf(zc + fabsf(dyx), eps) / fmaxf(wc + fabsf(dyy), eps);
            zg = fabsf(z1 - z0);
        }
        ((float4 *)p.rast)[i] = rast;
        ((float4 *)p.rast_db)[i] = rdb;
        store3(p.gb_pos, i, pos);
        store3(p.gb_gnrm, i, gn);
        store3(p.gb_nrm, i, nrm);
        store3(p.gb_tng, i, tng);
        p.gb_texc[2 * i] = tc0; p.gb_texc[2 * i + 1] = tc1;
        ((float4 *)p.gb_texc_db)[i] = make_float4(tdb[0], tdb[1], tdb[2], tdb[3]);
        p.gb_depth[2 * i] = z0; p.gb_depth[2 * i + 1] = zg;
    }
    // the last wavefront of this class to leave puts the pair of counters back to zero for the next launch (the waves of a class: those
    // with wid % 64 == cls)
    if (lane == 0u && (mode & 2)) {
        const unsigned mine = n_waves / NVDR_GB_CLASSES + (cls < n_waves % NVDR_GB_CLASSES ? 1u : 0u);
        if (atomicAdd(gone, 1u) + 1u == mine) {
            __hip_atomic_store(claim, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

unsigned query_grid(const nvdr_ctx *c, int64_t items);   // bvh.hip

static int gb_mode()
{
    static int mode = -1;
    if (mode < 0) {
        mode = 3;
        if (const char *e = nvdr_tuning_env("NVDR_GB_MODE")) mode = atoi(e) & 7;       // (bit 2: tiles whatever the launch size)
    }
    return mode;
}

extern "C" int nvdr_render_gbuffer(nvdr_ctx *c, const nvdr_gbuffer_args *a, void *stream_)
{
    NvdrRange range("nvdr_render_gbuffer");
    NVDR_REQUIRE(c && a, "nvdr_render_gbuffer: NULL argument");
    NVDR_REQUIRE(c->n_tris > 0, "nvdr_render_gbuffer: no BVH built on this context (call optix_build_bvh first)");
    NVDR_REQUIRE(a->n_tris == c->n_tris, "nvdr_render_gbuffer: the mesh has %lld triangles, the BVH of this context %lld",
                 (long long)a->n_tris, (long long)c->n_tris);
    NVDR_REQUIRE(a->v_pos && a->t_pos_idx && a->v_nrm && a->t_nrm_idx && a->v_tng && a->t_tng_idx && a->v_tex && a->t_tex_idx,
                 "nvdr_render_gbuffer: NULL mesh attribute");
    NVDR_REQUIRE(a->mvp && a->cam && a->n > 0 && a->h > 0 && a->w > 0, "nvdr_render_gbuffer: bad view block");
    NVDR_REQUIRE(a->rast && a->rast_db && a->gb_pos && a->gb_geometric_normal && a->gb_normal && a->gb_tangent && a->gb_texc &&
                     a->gb_texc_deriv && a->gb_depth, "nvdr_render_gbuffer: NULL output");
    if (int r0 = ctx_check_overflow(c, "nvdr_render_gbuffer")) return r0;
    NVDR_HIP_TRY(hipSetDevice(c->device));
    GbufferParams p;
    p.v_pos = a->v_pos; p.t_pos = a->t_pos_idx; p.v_nrm = a->v_nrm; p.t_nrm = a->t_nrm_idx;
    p.v_tng = a->v_tng; p.t_tng = a->t_tng_idx; p.v_tex = a->v_tex; p.t_tex = a->t_tex_idx;
    p.mvp = a->mvp; p.cam = a->cam; p.N = a->n; p.H = a->h; p.W = a->w;
    p.rast = a->rast; p.rast_db = a->rast_db; p.gb_pos = a->gb_pos; p.gb_gnrm = a->gb_geometric_normal; p.gb_nrm = a->gb_normal;
    p.gb_tng = a->gb_tangent; p.gb_texc = a->gb_texc; p.gb_texc_db = a->gb_texc_deriv; p.gb_depth = a->gb_depth;
    const int64_t total = (int64_t)p.N * p.H * p.W;
    if (int rw = ctx_wait_built(c, (hipStream_t)stream_)) return rw;
    gbuffer_kernel<<<query_grid(c, total), NVDR_QUERY_BLOCK, (NVDR_QUERY_BLOCK / 64) * NVDR_OSTACK_LDS * 64 * 8, (hipStream_t)stream_>>>(bvh_view(c), p, c->spill, c->queues, gb_mode());
    NVDR_LAUNCH_CHECK();
    return 0;
}
