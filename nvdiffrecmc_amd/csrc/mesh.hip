// mesh.hip -- the gradient route from the G-buffer back to the trained vertices and textures (second half).
//
// In the reference this route is made of nvdiffrast and torch autograd:
//   gb_* = dr.interpolate(attr, rast, idx)                         render/render.py:25,208-222  (adjoint: scatter by barycentrics)
//   rast = dr.rasterize(...)                                       render/render.py:308-310     (adjoint: barycentrics w.r.t. vertices)
//   v_nrm = auto_normals(v_pos), v_tng = compute_tangents(...)     render/mesh.py:150-219, geometry/dlmesh.py:45-55 (torch scatter_add)
//   kd, ks, normal = material[...].sample(gb_texc, ...)            render/render.py:61-68, train.py:171-192
// Here: one launch for normals + tangents and two for their adjoint (gathers over a vertex -> corner adjacency: a fixed
// summation order, no atomics, bit-reproducible vertex frames -- the G-buffer the shading decisions hang on comes from them);
// one launch for the adjoint of the interpolation including the barycentric term; one launch each way for the texel lookups.
// Nothing here synchronises the host.  dr.antialias (the silhouette gradient, render.py:290) has no counterpart.
#include "bsdf_device.h"

// util.safe_normalize (render/util.py:27-31): x / sqrt(clamp(dot(x, x), min = 1e-20))
__device__ __forceinline__ F3 sn_util(F3 x)
{
    const float l = sqrtf(fmaxf(dot3(x, x), 1e-20f));
    return f3(x.x / l, x.y / l, x.z / l);
}
// its adjoint as torch autograd forms it: the clamp passes no gradient below the bound
__device__ __forceinline__ F3 sn_util_bwd(F3 x, F3 g)
{
    const float d = dot3(x, x);
    if (d > 1e-20f) {
        const float l = sqrtf(d);
        const F3 y = f3(x.x / l, x.y / l, x.z / l);
        const float s = dot3(y, g);
        return f3((g.x - y.x * s) / l, (g.y - y.y * s) / l, (g.z - y.z * s) / l);
    }
    const float l = sqrtf(1e-20f);
    return f3(g.x / l, g.y / l, g.z / l);
}

__device__ __forceinline__ F3 ld3(const float *p, int64_t i) { return f3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ void st3(float *p, int64_t i, F3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

struct MeshView {
    const float *v_pos; const int *t_pos; const float *v_tex; const int *t_tex; const int *adj_start; const int *adj_corner;
    int V;
};

// per-triangle tangent of compute_tangents (mesh.py:196-204): returns the scale factors of pe1 and pe2 (tang = pe1 * k1 + pe2 * k2)
__device__ __forceinline__ void tangent_factors(const MeshView &m, int tri, float &k1, float &k2)
{
    const int j0 = m.t_tex[3 * tri], j1 = m.t_tex[3 * tri + 1], j2 = m.t_tex[3 * tri + 2];
    const float u1x = m.v_tex[2 * j1] - m.v_tex[2 * j0], u1y = m.v_tex[2 * j1 + 1] - m.v_tex[2 * j0 + 1];
    const float u2x = m.v_tex[2 * j2] - m.v_tex[2 * j0], u2y = m.v_tex[2 * j2 + 1] - m.v_tex[2 * j0 + 1];
    const float denom = u1x * u2y - u1y * u2x;
    const float d = denom > 0.0f ? fmaxf(denom, 1e-6f) : fminf(denom, -1e-6f);
    k1 = u2y / d;
    k2 = -u1y / d;
}

// sums over the corners that reference vertex v, in adjacency order: raw normal (sum of face normals) and raw tangent sum
__device__ __forceinline__ void vertex_sums(const MeshView &m, int v, F3 &vn, F3 &ts, int &cnt)
{
    vn = f3(0.0f); ts = f3(0.0f);
    const int b = m.adj_start[v], e = m.adj_start[v + 1];
    cnt = e - b;
    for (int q = b; q < e; ++q) {
        const int tri = m.adj_corner[q] / 3;
        const F3 p0 = ld3(m.v_pos, m.t_pos[3 * tri]), p1 = ld3(m.v_pos, m.t_pos[3 * tri + 1]), p2 = ld3(m.v_pos, m.t_pos[3 * tri + 2]);
        const F3 a = p1 - p0, c = p2 - p0;
        vn += cross3(a, c);
        if (m.v_tex) {
            float k1, k2;
            tangent_factors(m, tri, k1, k2);
            ts += f3(a.x * k1 + c.x * k2, a.y * k1 + c.y * k2, a.z * k1 + c.z * k2);   // nom / denom, mesh.py:199-204
        }
    }
}

__global__ void __launch_bounds__(256) mesh_frame_fwd_kernel(MeshView m, float *__restrict__ v_nrm, float *__restrict__ v_tng)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m.V) return;
    F3 vn, ts; int cnt;
    vertex_sums(m, v, vn, ts, cnt);
    // mesh.py:170-171: degenerate normals become (0, 0, 1), then safe_normalize
    if (!(dot3(vn, vn) > 1e-20f)) vn = f3(0.0f, 0.0f, 1.0f);
    const F3 n = sn_util(vn);
    st3(v_nrm, v, n);
    if (v_tng) {
        // mesh.py:206-214 (a vertex no triangle references keeps a zero tangent instead of the reference's 0 / 0)
        const float ic = 1.0f / (float)max(cnt, 1);
        const F3 t1 = sn_util(f3(ts.x * ic, ts.y * ic, ts.z * ic));
        const float s = dot3(t1, n);
        st3(v_tng, v, sn_util(f3(t1.x - s * n.x, t1.y - s * n.y, t1.z - s * n.z)));
    }
}

// adjoint, step A (per vertex): (v_nrm_grad, v_tng_grad) -> gradient of the raw normal sum and of the raw tangent sum
__global__ void __launch_bounds__(256) mesh_frame_bwd_vertex_kernel(MeshView m, const float *__restrict__ g_nrm, const float *__restrict__ g_tng,
                                                                    float *__restrict__ scratch)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m.V) return;
    F3 vn, ts; int cnt;
    vertex_sums(m, v, vn, ts, cnt);
    const bool live = dot3(vn, vn) > 1e-20f;
    const F3 vsel = live ? vn : f3(0.0f, 0.0f, 1.0f);
    const F3 n = sn_util(vsel);
    F3 gn = g_nrm ? ld3(g_nrm, v) : f3(0.0f);
    F3 gts = f3(0.0f);
    if (g_tng && m.v_tex) {
        const float ic = 1.0f / (float)max(cnt, 1);
        const F3 t0 = f3(ts.x * ic, ts.y * ic, ts.z * ic);
        const F3 t1 = sn_util(t0);
        const float s = dot3(t1, n);
        const F3 t2p = f3(t1.x - s * n.x, t1.y - s * n.y, t1.z - s * n.z);
        const F3 g2 = sn_util_bwd(t2p, ld3(g_tng, v));             // through the outer safe_normalize
        const float gdn = dot3(g2, n);
        const F3 g1 = f3(g2.x - gdn * n.x, g2.y - gdn * n.y, g2.z - gdn * n.z);      // t1 - (t1 . n) n  w.r.t. t1
        gn += f3(-s * g2.x - gdn * t1.x, -s * g2.y - gdn * t1.y, -s * g2.z - gdn * t1.z);   // ... w.r.t. n
        const F3 g0 = sn_util_bwd(t0, g1);
        gts = f3(g0.x * ic, g0.y * ic, g0.z * ic);
    }
    const F3 gvn = live ? sn_util_bwd(vsel, gn) : f3(0.0f);         // torch.where: no gradient into a replaced normal
    st3(scratch, 2 * (int64_t)v, gvn);
    st3(scratch, 2 * (int64_t)v + 1, gts);
}

// adjoint, step B (per vertex): gather over the corners of v what every adjacent triangle's face normal / tangent passes to v
__global__ void __launch_bounds__(256) mesh_frame_bwd_gather_kernel(MeshView m, const float *__restrict__ scratch, float *__restrict__ g_pos, int accumulate)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m.V) return;
    F3 acc = f3(0.0f);
    const int b = m.adj_start[v], e = m.adj_start[v + 1];
    for (int q = b; q < e; ++q) {
        const int ent = m.adj_corner[q], tri = ent / 3, corner = ent - 3 * tri;
        const int i0 = m.t_pos[3 * tri], i1 = m.t_pos[3 * tri + 1], i2 = m.t_pos[3 * tri + 2];
        const F3 p0 = ld3(m.v_pos, i0), p1 = ld3(m.v_pos, i1), p2 = ld3(m.v_pos, i2);
        const F3 a = p1 - p0, c = p2 - p0;
        const F3 gfn = (ld3(scratch, 2 * (int64_t)i0) + ld3(scratch, 2 * (int64_t)i1)) + ld3(scratch, 2 * (int64_t)i2);
        F3 ga = cross3(c, gfn), gc = cross3(gfn, a);                // fn = a x c
        if (m.v_tex) {
            const F3 gt = (ld3(scratch, 2 * (int64_t)i0 + 1) + ld3(scratch, 2 * (int64_t)i1 + 1)) + ld3(scratch, 2 * (int64_t)i2 + 1);
            float k1, k2;
            tangent_factors(m, tri, k1, k2);
            ga += gt * k1;
            gc += gt * k2;
        }
        acc += corner == 0 ? -(ga + gc) : (corner == 1 ? ga : gc);
    }
    if (accumulate) acc += ld3(g_pos, v);
    st3(g_pos, v, acc);
}

static int mesh_view(const nvdr_mesh_args *a, MeshView &m, const char *who)
{
    NVDR_REQUIRE(a && a->v_pos && a->t_pos_idx && a->adj_start && a->adj_corner, "%s: NULL mesh argument", who);
    NVDR_REQUIRE(a->n_verts > 0 && a->n_tris > 0 && a->n_verts < (1ll << 31) && a->n_tris < (1ll << 29), "%s: bad mesh size", who);
    NVDR_REQUIRE((a->v_tex == nullptr) == (a->t_tex_idx == nullptr), "%s: v_tex and t_tex_idx go together", who);
    m.v_pos = a->v_pos; m.t_pos = a->t_pos_idx; m.v_tex = a->v_tex; m.t_tex = a->t_tex_idx;
    m.adj_start = a->adj_start; m.adj_corner = a->adj_corner; m.V = (int)a->n_verts;
    return 0;
}

extern "C" int nvdr_mesh_frame_fwd(const nvdr_mesh_args *a, float *v_nrm, float *v_tng, void *stream)
{
    NvdrRange range("nvdr_mesh_frame_fwd");
    MeshView m;
    if (int r = mesh_view(a, m, "nvdr_mesh_frame_fwd")) return r;
    NVDR_REQUIRE(v_nrm, "nvdr_mesh_frame_fwd: NULL output");
    NVDR_REQUIRE(!v_tng || m.v_tex, "nvdr_mesh_frame_fwd: tangents need texture coordinates");
    mesh_frame_fwd_kernel<<<div_up(m.V, 256), 256, 0, (hipStream_t)stream>>>(m, v_nrm, v_tng);
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_mesh_frame_bwd(const nvdr_mesh_args *a, const float *v_nrm_grad, const float *v_tng_grad, float *scratch, float *v_pos_grad,
                                   int accumulate, void *stream)
{
    NvdrRange range("nvdr_mesh_frame_bwd");
    MeshView m;
    if (int r = mesh_view(a, m, "nvdr_mesh_frame_bwd")) return r;
    NVDR_REQUIRE(scratch && v_pos_grad, "nvdr_mesh_frame_bwd: NULL buffer");
    mesh_frame_bwd_vertex_kernel<<<div_up(m.V, 256), 256, 0, (hipStream_t)stream>>>(m, v_nrm_grad, v_tng_grad, scratch);
    NVDR_LAUNCH_CHECK();
    mesh_frame_bwd_gather_kernel<<<div_up(m.V, 256), 256, 0, (hipStream_t)stream>>>(m, scratch, v_pos_grad, accumulate);
    NVDR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// adjoint of the attribute interpolation (+ barycentrics)

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 d3(F3 a) { return D3{(double)a.x, (double)a.y, (double)a.z}; }
__device__ __forceinline__ D3 dsub(F3 a, F3 b) { return D3{(double)a.x - (double)b.x, (double)a.y - (double)b.y, (double)a.z - (double)b.z}; }
__device__ __forceinline__ D3 dadd(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 dmul(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 dcross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ F3 f3d(D3 a) { return f3((float)a.x, (float)a.y, (float)a.z); }

struct InterpBwd {
    const float *rast; int N, H, W;
    int64_t n_verts, n_tris;
    const float *v_pos; const int *t_pos; const float *v_nrm; const float *v_tng; const float *cam;
    const float *g_pos, *g_gn, *g_nrm, *g_tng;
    float *o_pos, *o_nrm, *o_tng;
};

__device__ __forceinline__ void atomic_add3(float *p, int64_t i, F3 v)
{
    __hip_atomic_fetch_add(p + 3 * i, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(p + 3 * i + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(p + 3 * i + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) interpolate_bwd_kernel(InterpBwd p)
{
    const int64_t total = (int64_t)p.N * p.H * p.W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float4 r = ((const float4 *)p.rast)[i];
    if (!(r.w > 0.0f)) return;
    const int64_t tri = (int64_t)r.w - 1;
    // a G-buffer of another topology (a stale `rast`, an index buffer that does not belong to v_pos) must not read or add out of bounds:
    // such a pixel is skipped, like an uncovered one
    if (tri < 0 || tri >= p.n_tris) return;            // (0 < w < 1 truncates to triangle -1)
    const float u = r.x, v = r.y, w2 = 1.0f - u - v;
    const int i0 = p.t_pos[3 * tri], i1 = p.t_pos[3 * tri + 1], i2 = p.t_pos[3 * tri + 2];
    if ((unsigned)i0 >= (uint64_t)p.n_verts || (unsigned)i1 >= (uint64_t)p.n_verts || (unsigned)i2 >= (uint64_t)p.n_verts) return;
    const F3 gp = p.g_pos ? ld3(p.g_pos, i) : f3(0.0f);
    const F3 gn = p.g_nrm ? ld3(p.g_nrm, i) : f3(0.0f);
    const F3 gt = p.g_tng ? ld3(p.g_tng, i) : f3(0.0f);
    F3 d0 = gp * u, d1 = gp * v, d2 = gp * w2;                       // gb_pos = u p0 + v p1 + (1 - u - v) p2
    const bool need_pos = p.o_pos && (p.g_pos || p.g_gn || p.cam);
    F3 p0 = f3(0.0f), p1 = f3(0.0f), p2 = f3(0.0f);
    if (need_pos) { p0 = ld3(p.v_pos, i0); p1 = ld3(p.v_pos, i1); p2 = ld3(p.v_pos, i2); }
    if (p.g_gn && p.o_pos) {
        // gb_geometric_normal = (u + v + (1 - u - v)) * safe_normalize(cross(p1 - p0, p2 - p0))   render.py:211-216
        const F3 g = ld3(p.g_gn, i) * ((u + v) + w2);
        const F3 a = p1 - p0, c = p2 - p0;
        const F3 gfn = sn_util_bwd(cross3(a, c), g);
        const F3 ga = cross3(c, gfn), gc = cross3(gfn, a);
        d1 += ga; d2 += gc; d0 -= ga + gc;
    }
    if (p.cam && p.o_pos) {
        // the barycentrics of the pixel's fixed primary ray as a function of the triangle (Moeller-Trumbore; b1, b2 = weights of
        // p1, p2, so u = 1 - b1 - b2, v = b1): dL/du = sum over attributes of g . (a0 - a2), dL/dv = g . (a1 - a2)
        double gu = ddot(d3(gp), dsub(p0, p2)), gv = ddot(d3(gp), dsub(p1, p2));
        if (p.g_nrm) { const F3 n2 = ld3(p.v_nrm, i2); gu += ddot(d3(gn), dsub(ld3(p.v_nrm, i0), n2)); gv += ddot(d3(gn), dsub(ld3(p.v_nrm, i1), n2)); }
        if (p.g_tng) { const F3 t2 = ld3(p.v_tng, i2); gu += ddot(d3(gt), dsub(ld3(p.v_tng, i0), t2)); gv += ddot(d3(gt), dsub(ld3(p.v_tng, i1), t2)); }
        const double gb1 = gv - gu, gb2 = -gu;
        const int x = (int)(i % p.W), y = (int)((i / p.W) % p.H), z = (int)(i / ((int64_t)p.W * p.H));
        const float X = ((float)x + 0.5f) / (float)p.W * 2.0f - 1.0f, Y = ((float)y + 0.5f) / (float)p.H * 2.0f - 1.0f;
        const float *cam = p.cam + 12 * z;
        const F3 eye = f3(cam[0], cam[1], cam[2]);
        const F3 d = (f3(cam[3], cam[4], cam[5]) * X + f3(cam[6], cam[7], cam[8]) * Y) + f3(cam[9], cam[10], cam[11]);
        // in double: for a triangle seen nearly edge-on det is a difference of nearly equal products and the adjoint goes with 1 / det^2
        // (fp32 left 4e-4 of the largest gradient on bob at 96 x 96); ~80 fp64 operations per covered pixel
        const D3 e1 = dsub(p1, p0), e2 = dsub(p2, p0), s = dsub(eye, p0), dd = d3(d);
        const D3 pv = dcross(dd, e2), qv = dcross(s, e1);
        const double det = ddot(e1, pv);
        if (fabs(det) > 1e-30) {
            const double inv = 1.0 / det, sp = ddot(s, pv), dq = ddot(dd, qv);
            const double g_det = -(gb1 * sp + gb2 * dq) * inv * inv;
            const double c1 = gb1 * inv, c2 = gb2 * inv;
            D3 g_s = dmul(pv, c1), g_pv = dadd(dmul(s, c1), dmul(e1, g_det)), g_e1 = dmul(pv, g_det);
            const D3 g_qv = dmul(dd, c2);
            const D3 g_e2 = dcross(g_pv, dd);                       // pv = d x e2
            g_s = dadd(g_s, dcross(e1, g_qv));                      // qv = s x e1
            g_e1 = dadd(g_e1, dcross(g_qv, s));
            d1 += f3d(g_e1); d2 += f3d(g_e2); d0 -= f3d(dadd(dadd(g_s, g_e1), g_e2));
        }
    }
    if (p.o_pos && need_pos) { atomic_add3(p.o_pos, i0, d0); atomic_add3(p.o_pos, i1, d1); atomic_add3(p.o_pos, i2, d2); }
    if (p.o_nrm && p.g_nrm) { atomic_add3(p.o_nrm, i0, gn * u); atomic_add3(p.o_nrm, i1, gn * v); atomic_add3(p.o_nrm, i2, gn * w2); }
    if (p.o_tng && p.g_tng) { atomic_add3(p.o_tng, i0, gt * u); atomic_add3(p.o_tng, i1, gt * v); atomic_add3(p.o_tng, i2, gt * w2); }
}

extern "C" int nvdr_interpolate_bwd(const nvdr_interpolate_bwd_args *a, void *stream)
{
    NvdrRange range("nvdr_interpolate_bwd");
    NVDR_REQUIRE(a && a->rast && a->v_pos && a->t_pos_idx, "nvdr_interpolate_bwd: NULL argument");
    NVDR_REQUIRE(a->n > 0 && a->h > 0 && a->w > 0 && a->n_verts > 0 && a->n_tris > 0, "nvdr_interpolate_bwd: bad extent");
    NVDR_REQUIRE(!a->cam || ((!a->gb_normal_grad || a->v_nrm) && (!a->gb_tangent_grad || a->v_tng)),
                 "nvdr_interpolate_bwd: the barycentric term needs v_nrm / v_tng for the gradients that are passed");
    NVDR_REQUIRE(!a->gb_normal_grad || a->v_nrm_grad, "nvdr_interpolate_bwd: gb_normal_grad without v_nrm_grad");
    NVDR_REQUIRE(!a->gb_tangent_grad || a->v_tng_grad, "nvdr_interpolate_bwd: gb_tangent_grad without v_tng_grad");
    InterpBwd p;
    p.rast = a->rast; p.N = a->n; p.H = a->h; p.W = a->w;
    p.n_verts = a->n_verts; p.n_tris = a->n_tris;
    p.v_pos = a->v_pos; p.t_pos = a->t_pos_idx; p.v_nrm = a->v_nrm; p.v_tng = a->v_tng; p.cam = a->cam;
    p.g_pos = a->gb_pos_grad; p.g_gn = a->gb_geometric_normal_grad; p.g_nrm = a->gb_normal_grad; p.g_tng = a->gb_tangent_grad;
    p.o_pos = a->v_pos_grad; p.o_nrm = a->v_nrm_grad; p.o_tng = a->v_tng_grad;
    const int64_t total = (int64_t)p.N * p.H * p.W;
    interpolate_bwd_kernel<<<div_up(total, 256), 256, 0, (hipStream_t)stream>>>(p);
    NVDR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// nearest-texel lookups of the trained textures

struct TexLookup {
    int n_tex; int res[NVDR_MAX_TEXTURES];
    const float *tex[NVDR_MAX_TEXTURES]; float *out[NVDR_MAX_TEXTURES];
    const float *dout[NVDR_MAX_TEXTURES]; float *dtex[NVDR_MAX_TEXTURES];
    const float *texc, *rast; int64_t P;
};

__device__ __forceinline__ int64_t texel_of(float s, float t, int R)
{
    // (tc * R).long().clamp(0, R - 1): truncation toward zero, then the clamp (a NaN coordinate lands on texel 0)
    const float fx = s * (float)R, fy = (1.0f - t) * (float)R;
    const int ix = fx >= 0.0f ? (fx < (float)R ? (int)fx : R - 1) : 0;
    const int iy = fy >= 0.0f ? (fy < (float)R ? (int)fy : R - 1) : 0;
    return (int64_t)iy * R + ix;
}

template <bool BWD>
__global__ void __launch_bounds__(256) texture_lookup_kernel(TexLookup p)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.P) return;
    const bool covered = ((const float4 *)p.rast)[i].w > 0.0f;
    const float s = p.texc[2 * i], t = p.texc[2 * i + 1];
#pragma unroll
    for (int k = 0; k < NVDR_MAX_TEXTURES; ++k) {
        if (k >= p.n_tex) break;
        const int64_t j = texel_of(s, t, p.res[k]);
        if (BWD) {
            if (covered) atomic_add3(p.dtex[k], j, ld3(p.dout[k], i));
        } else {
            st3(p.out[k], i, covered ? ld3(p.tex[k], j) : f3(0.0f));
        }
    }
}

static int tex_params(const nvdr_texture_args *a, TexLookup &p, bool bwd, const char *who)
{
    NVDR_REQUIRE(a && a->texc && a->rast && a->n_pix >= 0, "%s: NULL argument", who);
    NVDR_REQUIRE(a->n_tex >= 1 && a->n_tex <= NVDR_MAX_TEXTURES, "%s: %d textures (1..%d supported)", who, a->n_tex, NVDR_MAX_TEXTURES);
    memset(&p, 0, sizeof(p));
    p.n_tex = a->n_tex; p.texc = a->texc; p.rast = a->rast; p.P = a->n_pix;
    for (int k = 0; k < a->n_tex; ++k) {
        NVDR_REQUIRE(a->res[k] >= 1 && a->res[k] <= 32768, "%s: texture %d has resolution %d", who, k, a->res[k]);
        NVDR_REQUIRE(bwd ? (a->dout[k] && a->dtex[k]) : (a->tex[k] && a->out[k]), "%s: texture %d has a NULL buffer", who, k);
        p.res[k] = a->res[k]; p.tex[k] = a->tex[k]; p.out[k] = a->out[k]; p.dout[k] = a->dout[k]; p.dtex[k] = a->dtex[k];
    }
    return 0;
}

extern "C" int nvdr_texture_lookup_fwd(const nvdr_texture_args *a, void *stream)
{
    TexLookup p;
    if (int r = tex_params(a, p, false, "nvdr_texture_lookup_fwd")) return r;
    if (p.P == 0) return 0;
    texture_lookup_kernel<false><<<div_up(p.P, 256), 256, 0, (hipStream_t)stream>>>(p);
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_texture_lookup_bwd(const nvdr_texture_args *a, void *stream)
{
    TexLookup p;
    if (int r = tex_params(a, p, true, "nvdr_texture_lookup_bwd")) return r;
    if (!a->accumulate)
        for (int k = 0; k < p.n_tex; ++k)
            NVDR_HIP_TRY(hipMemsetAsync(p.dtex[k], 0, sizeof(float) * 3 * (size_t)p.res[k] * p.res[k], (hipStream_t)stream));
    if (p.P == 0) return 0;
    texture_lookup_kernel<true><<<div_up(p.P, 256), 256, 0, (hipStream_t)stream>>>(p);
    NVDR_LAUNCH_CHECK();
    return 0;
}
