/* nvdr_hip.h -- C ABI of libnvdr_hip.so: the MI355X-native replacement for the two native
 * plugins of NVlabs/nvdiffrecmc's differentiable Monte-Carlo direct-lighting path.
 *
 * The reference has no C ABI; its native surface is two pybind11 torch extensions:
 *   optixutils_plugin   render/optixutils/c_src/torch_bindings.cpp:321-328
 *   renderutils_plugin  render/renderutils/c_src/torch_bindings.cpp:866-888
 * Every entry point below replaces one of those bindings (cited per function).  The contract:
 *   - plain pointers + sizes + element strides, no torch types;
 *   - all pointers are DEVICE pointers (HBM) unless the name says host;
 *   - outputs are caller-allocated, contiguous, and fully overwritten (zero-filled where the
 *     reference relies on torch::zeros) by the callee;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and nothing
 *     synchronises the host (the reference blocks in cudaStreamSynchronize after every env-shade
 *     launch, torch_bindings.cpp:185,269 -- deliberately not reproduced);
 *   - every function returns 0 on success, otherwise a hipError_t value (or -1 for argument
 *     errors); nvdr_last_error() gives the message.  The reference swallows CUDA/OptiX errors
 *     (render/optixutils/c_src/common.h:37-61) -- deliberately not reproduced.
 *
 * INTEGRATION.md shows the Python (ctypes) binding a maintainer of the reference would add.
 */
#ifndef NVDR_HIP_H
#define NVDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A strided view of up to 4 dims.  Strides are in ELEMENTS.  A dim of size 1 broadcasts on read,
 * exactly like fetch3()/fetch2() in render/optixutils/c_src/common.h:13-27 and Tensor::nhwcIndex in
 * render/renderutils/c_src/tensor.h:31.  Unused leading/trailing dims have size 1, stride 0. */
typedef struct nvdr_tensor {
    void   *data;
    int64_t size[4];
    int64_t stride[4];
} nvdr_tensor;

typedef struct nvdr_ctx nvdr_ctx; /* replaces OptiXStateWrapper, optix_wrapper.h:17-37 */

const char *nvdr_last_error(void);
int nvdr_version(void);

/* ---- context: owns the BVH buffers, replaces OptiXStateWrapper ctor/dtor (optix_wrapper.cpp:306-348).
 * ONE STREAM PER CONTEXT AT A TIME: the scratch of a context (tree, ray stream, traversal stacks, chunk counters, light-gradient
 * partials) is shared by all its launches and is grown / freed after synchronising only the stream of the call that needs more.
 * Launches of one context must therefore be ordered on one stream (or by events); use one context per concurrent stream.  A
 * buffer that has to grow inside a HIP-graph capture fails the capture: run a few iterations eagerly first (trainer.py does). */
int nvdr_ctx_create(nvdr_ctx **out, int device);
int nvdr_ctx_destroy(nvdr_ctx *ctx);
/* Synchronises `stream` and reports (non-zero + nvdr_last_error) if any traversal launch on this context ever pushed
 * beyond its stack bound -- the only way a visibility answer could be wrong.  Every other entry point checks the same
 * (host-mapped) flag without synchronising, so such a failure surfaces at the latest on the next call.  The reference
 * has no counterpart: OptiX errors are swallowed (render/optixutils/c_src/common.h:37-77). */
int nvdr_ctx_check(nvdr_ctx *ctx, void *stream);
/* Bytes of HBM the env-shade ray stream of this context may take (default 8 GiB, or NVDR_STREAM_BUDGET_MB read when the
 * context is created).  The stream holds one chunk of covered pixels (2*S rays x 25 B each); a launch whose covered pixels
 * exceed the chunk is processed chunk by chunk with identical results.  The reference needs no scratch (one thread per
 * pixel keeps its rays in registers); a worst-case allocation would be N*H*W*2S*25 B (16 GB for 8 x 800^2 x 64 spp). */
int nvdr_ctx_set_stream_budget(nvdr_ctx *ctx, int64_t bytes);
/* Optional device allocator for everything the context owns beyond a few control words (BVH buffers, ray stream, stack spill:
 * hundreds of MB).  Without it the library calls hipMalloc / hipFree.  The Python shim passes torch's caching allocator, so the
 * scratch is visible to -- and recycled by -- the framework that owns the GPU's memory (the reference allocates its GAS with
 * cudaMalloc, torch_bindings.cpp:84-95, and leaks a launch-parameter block per call, :174-180).  alloc returns NULL on failure;
 * `stream` is the stream of the call that needs the memory.  Must be set before the context allocates anything. */
typedef void *(*nvdr_alloc_fn)(size_t bytes, int device, void *stream, void *user);
typedef void (*nvdr_free_fn)(void *ptr, void *user);
int nvdr_ctx_set_allocator(nvdr_ctx *ctx, nvdr_alloc_fn alloc_fn, nvdr_free_fn free_fn, void *user);
/* Where the BVH build runs: 1 (default) = on a side stream of the context, overlapped with whatever the caller enqueues next that
 * does not read the tree; 0 = on the caller's stream; 2 = side stream, but the build's launches are issued when the first consumer of
 * the tree is called (env-shade: behind its sample generation), so that in a captured HIP graph of a launch-bound iteration the
 * caller's own front nodes come first.  The caller's buffers are copied inside nvdr_bvh_build in every mode. */
int nvdr_ctx_set_build_mode(nvdr_ctx *ctx, int mode);
/* Make `stream` wait (device side, no host synchronisation) for the build nvdr_bvh_build last enqueued -- what every consumer of the tree does by
 * itself before its first launch.  For a caller that captures the build in a HIP graph of its own: a capture must not end with the side
 * stream's work unjoined. */
int nvdr_bvh_wait(nvdr_ctx *ctx, void *stream);
/* The caller has ordered its streams behind the context's last build itself (e.g. the build was captured into a HIP graph of its own,
 * joined inside that graph by nvdr_bvh_wait, and the replays are ordered by the caller's events): consumers issue no wait of their own
 * until the next nvdr_bvh_build. */
int nvdr_bvh_mark_joined(nvdr_ctx *ctx);
/* ---- optix_build_bvh (torch_bindings.cpp:37-116).  verts f32[V,3] contiguous, tris i32[T,3]
 * contiguous.  rebuild > 0: full LBVH build (Morton codes, radix sort, hierarchy, bounds);
 * rebuild == 0: refit the bounds of the existing hierarchy to moved vertices (OPTIX_BUILD_OPERATION_UPDATE). */
int nvdr_bvh_build(nvdr_ctx *ctx, const float *verts, int64_t n_verts, const int32_t *tris, int64_t n_tris,
                   int rebuild, void *stream);

/* Introspection / test hooks (additive; no reference counterpart). */
typedef struct nvdr_bvh_info {
    int64_t n_tris;
    int64_t n_nodes;      /* internal nodes = n_tris - 1 (0 for a single triangle) */
    int32_t height;       /* longest root-to-leaf path, in internal nodes */
    int32_t root;         /* index of the root node */
    float   aabb_min[3];
    float   aabb_max[3];
    float   grid_lo[3];   /* quantisation grid of the node boxes: grid = (world - grid_lo) * grid_scale + 2 */
    float   grid_scale[3];
    int32_t stack_max;    /* traversal-stack entries per lane the context provides (proven bound, see csrc/bvh.h) */
} nvdr_bvh_info;
int nvdr_bvh_info_get(nvdr_ctx *ctx, nvdr_bvh_info *out_host, void *stream); /* synchronises `stream` */
/* copy the device BVH to host buffers: nodes = n_nodes * 8 uint32 (32-B records: six words of 16-bit
 * quantised child boxes + two child indices, layout in nvdiffrecmc_amd/csrc/bvh.h and DESIGN.md),
 * tri_records = n_tris*12 floats (v0, e1, e2, {orig index bits, 0, 0}) */
int nvdr_bvh_export(nvdr_ctx *ctx, float *nodes_host, float *tri_records_host, void *stream);

/* Any-hit visibility of R rays against the BVH: out_vis[r] = 1 if NO triangle is hit for
 * t in (0, 1e16) (same convention as shadow_test(), kernel.cu:101-118: 1 = unoccluded).
 * ro, rd f32[R,3] contiguous.  counters (optional, may be NULL): uint64[2] device accumulators
 * {box tests, triangle tests} used for the algorithmic-byte roofline figure. */
int nvdr_trace_visibility(nvdr_ctx *ctx, const float *ro, const float *rd, int64_t n_rays, uint8_t *out_vis,
                          unsigned long long *counters, void *stream);

/* Same answer through the PRODUCTION shadow-ray kernel of env-shade (persistent wavefronts over the eight-wide compressed nodes,
 * deferred triangle tests): a test hook that lets arbitrary rays -- grazing, degenerate meshes -- reach the kernel the renderer
 * actually runs.  The _counted form runs the counting build of that kernel; counters: uint64[NVDR_COUNTERS_LEN] zeroed by the
 * caller, laid out as nvdr_env_shade_args.counters. */
int nvdr_trace_visibility_wide(nvdr_ctx *ctx, const float *ro, const float *rd, int64_t n_rays, uint8_t *out_vis, void *stream);
int nvdr_trace_visibility_wide_counted(nvdr_ctx *ctx, const float *ro, const float *rd, int64_t n_rays, uint8_t *out_vis,
                                       unsigned long long *counters, void *stream);
/* copy the eight-wide tree the shadow rays walk to host buffers: oct = counts[0] * 16 uint32 (64-B records, layout in
 * nvdiffrecmc_amd/csrc/bvh.h), tris8 = n_tris * 12 floats (the triangle records in the order the oct nodes refer to them);
 * counts_host int64[3] = {oct nodes, triangles placed, nodes finished}.  oct / tris8 may be NULL (counts only). */
int nvdr_bvh_export_oct(nvdr_ctx *ctx, uint32_t *oct_host, float *tris8_host, int64_t *counts_host, void *stream);

/* Closest hit of R rays: out_t f32[R] (<0 = miss), out_tri i32[R] (original triangle index, -1 = miss),
 * out_uv f32[R,2] barycentrics of v1, v2.  G-buffer producer building block. */
int nvdr_trace_closest(nvdr_ctx *ctx, const float *ro, const float *rd, int64_t n_rays, float *out_t,
                       int32_t *out_tri, float *out_uv, void *stream);

/* ---- G-buffer producer (additive): what render_layer obtains from nvdiffrast's rasterize + interpolate
 * (render/render.py:208-234, :279) for N camera views, from primary rays traced through the context's BVH.
 * All pointers are device pointers; index arrays are int32 [T,3]; outputs are contiguous NHWC float32. */
typedef struct nvdr_gbuffer_args {
    const float   *v_pos;     const int32_t *t_pos_idx;   /* [V,3], the mesh the context's BVH was built from */
    const float   *v_nrm;     const int32_t *t_nrm_idx;   /* [Vn,3] */
    const float   *v_tng;     const int32_t *t_tng_idx;   /* [Vn,3] tangents (render/mesh.py:181-219) */
    const float   *v_tex;     const int32_t *t_tex_idx;   /* [Vt,2] */
    int64_t        n_tris;
    const float   *mvp;       /* [N,4,4] row-major model-view-projection (render.py:271 v_pos_clip = mtx_in @ v_pos) */
    const float   *cam;       /* [N,4,3]: eye, U, V, W -- the primary ray through NDC (X, Y) is normalize(X U + Y V + W) */
    int32_t        n, h, w;
    float *rast;                /* [N,H,W,4] (u, v, z/w, triangle_id + 1): nvdiffrast's rasterize output; 0 = background */
    float *rast_db;             /* [N,H,W,4] (du/dX, du/dY, dv/dX, dv/dY) per pixel */
    float *gb_pos;              /* [N,H,W,3] render.py:208 */
    float *gb_geometric_normal; /* [N,H,W,3] render.py:211-216 */
    float *gb_normal;           /* [N,H,W,3] render.py:220 */
    float *gb_tangent;          /* [N,H,W,3] render.py:221 */
    float *gb_texc;             /* [N,H,W,2] render.py:225 */
    float *gb_texc_deriv;       /* [N,H,W,4] render.py:225 */
    float *gb_depth;            /* [N,H,W,2] (z/w, |dz|) render.py:228-234 */
} nvdr_gbuffer_args;
int nvdr_render_gbuffer(nvdr_ctx *ctx, const nvdr_gbuffer_args *args, void *stream);

/* ---- geometry / material gradient route (additive, SURVEY 8 f1 second half): the adjoints of what render_layer takes from
 * nvdiffrast and render/mesh.py, so that gb_pos_grad / gb_normal_grad of env_shade_bwd and the gradients of
 * prepare_shading_normal_bwd reach the trained vertices and textures (render/render.py:25,85-99,208-234, render/mesh.py:150-219,
 * geometry/dlmesh.py:45-55, train.py:171-192).  All pointers are device pointers. */

/* A mesh whose normals and tangents are indexed like its positions (mesh.py:178,219: t_nrm_idx = t_tng_idx = t_pos_idx) plus the
 * vertex -> (triangle, corner) adjacency that makes its per-vertex sums gathers in a FIXED order (deterministic, no atomics):
 * adj_corner[adj_start[v] .. adj_start[v+1]) = the values 3*triangle + corner of every corner that references vertex v, ascending.
 * (numpy: flat = t_pos_idx.reshape(-1); adj_corner = argsort(flat, kind='stable'); adj_start = searchsorted(flat[adj_corner], arange(V+1)).) */
typedef struct nvdr_mesh_args {
    const float   *v_pos;      int64_t n_verts;   /* [V,3] */
    const int32_t *t_pos_idx;  int64_t n_tris;    /* [T,3] */
    const float   *v_tex;      const int32_t *t_tex_idx;   /* [Vt,2], [T,3] */
    const int32_t *adj_start;  /* [V+1] */
    const int32_t *adj_corner; /* [3T] */
} nvdr_mesh_args;
/* auto_normals (mesh.py:150-178) + compute_tangents (mesh.py:181-219) in one launch: v_nrm, v_tng f32 [V,3] */
int nvdr_mesh_frame_fwd(const nvdr_mesh_args *mesh, float *v_nrm, float *v_tng, void *stream);
/* their adjoint: v_nrm_grad, v_tng_grad f32 [V,3] (either may be NULL = zero) -> v_pos_grad [V,3]; accumulate != 0 adds to what
 * v_pos_grad holds (the scatter of nvdr_interpolate_bwd), otherwise it is overwritten.  scratch: f32 [V,6]. */
int nvdr_mesh_frame_bwd(const nvdr_mesh_args *mesh, const float *v_nrm_grad, const float *v_tng_grad, float *scratch,
                        float *v_pos_grad, int accumulate, void *stream);

/* Adjoint of the attribute interpolation of render_layer (dr.interpolate, render.py:25,208-222) at the pixels of `rast`
 * (nvdr_render_gbuffer's output: (u, v, z/w, triangle_id + 1)): scatters the per-pixel gradients of gb_pos, gb_geometric_normal
 * (through safe_normalize(cross(v1 - v0, v2 - v0)), render.py:211-214), gb_normal and gb_tangent into v_pos_grad / v_nrm_grad /
 * v_tng_grad with fp32 atomics (the caller zero-fills them; any gradient pointer may be NULL).  With cam != NULL the barycentrics
 * are differentiated as well -- (u, v) of the pixel's FIXED primary ray as a function of the triangle's vertices, the part
 * nvdiffrast's rasterize backward supplies (same function of the vertices: a projective map keeps barycentrics) -- otherwise they
 * are held constant.  The silhouette term (dr.antialias, render.py:290) has no counterpart here. */
typedef struct nvdr_interpolate_bwd_args {
    const float   *rast;      /* [N,H,W,4] */
    int32_t        n, h, w;
    const float   *v_pos;     const int32_t *t_pos_idx;   /* [V,3], [T,3] */
    const float   *v_nrm;     const float   *v_tng;       /* [V,3] each, indexed by t_pos_idx (needed for the barycentric term) */
    int64_t        n_verts, n_tris;
    const float   *cam;       /* [N,4,3] as nvdr_gbuffer_args.cam, or NULL */
    const float   *gb_pos_grad, *gb_geometric_normal_grad, *gb_normal_grad, *gb_tangent_grad;   /* [N,H,W,3] contiguous or NULL */
    float         *v_pos_grad, *v_nrm_grad, *v_tng_grad;  /* [V,3], accumulated into */
} nvdr_interpolate_bwd_args;
int nvdr_interpolate_bwd(const nvdr_interpolate_bwd_args *args, void *stream);

/* Nearest-texel lookup of up to NVDR_MAX_TEXTURES three-channel textures at the interpolated texture coordinate (stand-in for
 * Texture2D.sample / dr.texture of render.py:61-68, whose trilinear mip filter lies outside the path): texel (ix, iy) =
 * (clamp(int(s * R), 0, R - 1), clamp(int((1 - t) * R), 0, R - 1)) of a row-major [R,R,3] texture; pixels with rast.w <= 0 get 0.
 * texc f32 [P,2], rast f32 [P,4], out[k] f32 [P,3].  The backward zero-fills dtex[k] [R_k,R_k,3] and accumulates with atomics. */
#define NVDR_MAX_TEXTURES 4
typedef struct nvdr_texture_args {
    int32_t        n_tex;
    int32_t        res[NVDR_MAX_TEXTURES];
    const float   *tex[NVDR_MAX_TEXTURES];
    const float   *texc, *rast;
    int64_t        n_pix;
    float         *out[NVDR_MAX_TEXTURES];          /* forward */
    const float   *dout[NVDR_MAX_TEXTURES];         /* backward: [P,3] contiguous */
    float         *dtex[NVDR_MAX_TEXTURES];
    int32_t        accumulate;                      /* backward: != 0 adds to what dtex[k] holds instead of zero-filling it first (a persistent
                                                       gradient buffer that its consumer re-zeroes: nvdr_adam_tensor.zero_grad) */
} nvdr_texture_args;
int nvdr_texture_lookup_fwd(const nvdr_texture_args *args, void *stream);
int nvdr_texture_lookup_bwd(const nvdr_texture_args *args, void *stream);

/* ---- env_shade_fwd / env_shade_bwd (torch_bindings.cpp:123-272; raygen program kernel.cu:463-542) */
#define NVDR_COUNTERS_BVH2 (8 + 2 * 8192)
#define NVDR_COUNTERS_PHASES (NVDR_COUNTERS_BVH2 + 8)
#define NVDR_COUNTERS_LEN (NVDR_COUNTERS_PHASES + 16)
typedef struct nvdr_env_shade_args {
    nvdr_tensor mask;        /* f32 [N,H,W]    (>0 = covered)                     params.h:17 */
    nvdr_tensor ro;          /* f32 [N,H,W,3]  shadow-ray origins                 params.h:14 */
    nvdr_tensor gb_pos;      /* f32 [N,H,W,3]                                      params.h:18 */
    nvdr_tensor gb_normal;   /* f32 [N,H,W,3]                                      params.h:20 */
    nvdr_tensor gb_view_pos; /* f32 [N,1,1,3] or [N,H,W,3]                         params.h:22 */
    nvdr_tensor gb_kd;       /* f32 [N,H,W,3]                                      params.h:23 */
    nvdr_tensor gb_ks;       /* f32 [N,H,W,3]  (occlusion, roughness, metalness)   params.h:25 */
    nvdr_tensor light;       /* f32 [Hl,Wl,3]  lat-long radiance                   params.h:29 */
    nvdr_tensor pdf;         /* f32 [Hl,Wl]                                        params.h:31 */
    nvdr_tensor rows;        /* f32 [Hl]       row CDF                             params.h:32 */
    nvdr_tensor cols;        /* f32 [Hl,Wl]    per-row column CDF                  params.h:33 */
    nvdr_tensor perms;       /* i32 [NP,S]     stratum permutation table           params.h:42 */
    uint32_t bsdf;           /* 0 'pbr', 1 'diffuse', 2 'white'   (ops.py:136) */
    uint32_t n_samples_x;    /* S = n_samples_x^2 strata, 2 shadow rays per stratum */
    uint32_t rnd_seed;
    float    shadow_scale;
    uint32_t pixel_index_offset; /* added to the linear pixel index that seeds the RNG (kernel.cu:504);
                                    0 reproduces the reference; rank*H*W makes a one-view-per-GPU
                                    shard draw the same streams as the single-GPU batch */
    /* forward outputs, f32 [N,H,W,3] contiguous */
    float *diff;
    float *spec;
    /* backward inputs */
    nvdr_tensor diff_grad;   /* f32 [N,H,W,3] */
    nvdr_tensor spec_grad;   /* f32 [N,H,W,3] */
    /* backward outputs, contiguous: four f32 [N,H,W,3] and light_grad f32 [Hl,Wl,3] */
    float *gb_pos_grad;
    float *gb_normal_grad;
    float *gb_kd_grad;
    float *gb_ks_grad;
    float *light_grad;
    /* optional visibility cache, uint32 [N*H*W, 2, ceil(S/32)] (bit i of plane 0 / 1 = the light- /
       BSDF-sampled shadow ray of stratum i is occluded): written by fwd when non-NULL; when non-NULL in
       bwd the shadow rays are NOT re-traced (valid only for identical seed and inputs). */
    uint32_t *vis_cache;
    /* optional device accumulators uint64[NVDR_COUNTERS_LEN], zeroed by the caller (a COUNTING build of the traversal
       kernel runs instead of the production one; feeds the roofline figures):
         [0] box tests of the production (oct) walk (non-empty slots only)   [1] its triangle tests   [2] rays traversed
         [3] sum and [4] max of the per-wavefront busy time in 100 MHz ticks   [5] wavefronts
         [6] sum of the per-wavefront shader-clock cycles   [7] bit mask of the XCDs that ran wavefronts
         [8 .. 8+2*8192) (begin, end) ticks of every wavefront
         [NVDR_COUNTERS_BVH2 + 0] node visits, [+1] triangle tests, [+2] rays of the CANONICAL binary any-hit walk over
         the same live rays (reference accounting layout: 32-B BVH2 node, 36-B triangle) -- invariant to how speculative
         the production walk is, checked against a CPU walk of the exported tree in tests/test_gpu_bvh.py.
         [NVDR_COUNTERS_BVH2 + 3] node steps of the production walk (one 64-byte node fetch + eight box tests each),
         [+4] triangle-test batches, [+5] lanes those batches filled (= triangle tests; / 64 / batches = their occupancy),
         [+6] node steps served by the treetop table in LDS (round 6; the rest fetch their node through the vector memory path).
         [NVDR_COUNTERS_PHASES + 0 .. 15] (round 6) shader-clock cycles of the wavefront loop by phase, summed over the wavefronts, from two
         PHASE-CLOCK builds of the kernel that a counting launch runs behind the counting kernel on the same rays (trace_kernel.h):
           build 1 (clock reads at wave-uniform points only, no vector register):  [0] refill (votes, chunk claims, ray fetch + set-up)
             [1] node step (pop, address, node fetch, box arithmetic, hit masks, push)  [2] leaf-queue append rounds  [3] triangle batches
             [4] loop iterations  [5] iterations with a node step in some lane  [6] total cycles of the wavefronts  [7] wavefronts
           build 2 (waits for the node behind its loads and reads the clock inside the node step):  [8] refill  [9] node fetch (pop, address,
             four 16-byte loads, wait)  [10] box arithmetic of the eight children  [11] hit masks + push + group bookkeeping  [12] queue rounds
             [13] triangle batches  [14] total cycles  [15] loop iterations.
       Rays traversed < 2*S*pixels: samples with dot(n, wi) <= 0 contribute exactly zero through the BSDF's own
       gates whatever their visibility and are not traced (NVDR_DEBUG bit 8 traces them anyway). */
    unsigned long long *counters;
    /* backward only: id (nvdr_env_shade_stream_id) of the forward launch whose inputs and seed this backward pass
       repeats.  When it is still the most recent ray stream generated on the context (and the launch's covered pixels
       fitted one chunk of the stream -- decided on the device), sample generation is skipped and the stored rays are
       traced again (or, with vis_cache, only re-shaded).  0 = always regenerate.  A backward pass consumes the stream
       (it leaves light-gradient records in it): a second backward pass with the same id regenerates it. */
    uint64_t reuse_stream_id;
    /* optional (may be NULL): device pointer to ONE uint32 added to rnd_seed by the kernels themselves.  A host-side seed
       is a launch parameter and would be frozen into a captured HIP graph; with the counter in device memory a replayed
       iteration draws fresh samples (render.py:112-116 increments its seed once per shade() call). */
    const uint32_t *rnd_seed_offset;
    /* optional, forward only (with rnd_seed_offset): the launch first copies *rnd_seed_offset to *rnd_seed_snapshot, uses THAT value,
       and then adds rnd_seed_advance to *rnd_seed_offset -- the "rnd_seed += 1" of render.py:116 and the snapshot a later backward
       pass needs (pass the snapshot as its rnd_seed_offset) without two extra launches per shade() call. */
    uint32_t *rnd_seed_snapshot;
    uint32_t  rnd_seed_advance;
    /* forward only: 0 = the whole launch (default).  A caller that wants to put something between the sample generation and the first
       kernel that reads the BVH -- trainer.py cuts its HIP graph there, so that a rebuild replayed on another stream is joined in front of
       the traversal instead of in front of the whole launch -- issues the SAME arguments twice: phase 1 = everything up to and including
       the sample generation, phase 2 = traversal and shading.  Only a launch that fits ONE chunk of the ray stream can be cut (the chunks
       of a larger one interleave their stages): phase 1 of a larger launch enqueues nothing and phase 2 all of it. */
    uint32_t  phase;
} nvdr_env_shade_args;
int nvdr_env_shade_fwd(nvdr_ctx *ctx, const nvdr_env_shade_args *args, void *stream);
int nvdr_env_shade_bwd(nvdr_ctx *ctx, const nvdr_env_shade_args *args, void *stream);
/* number of covered pixels seen by the last env-shade launch on this ctx (device counter read back;
 * synchronises `stream`).  rays per pass = 2 * S * this. */
int nvdr_env_shade_last_pixel_count(nvdr_ctx *ctx, int64_t *out_host, void *stream);
/* id of the ray stream written by the most recent env-shade launch on this ctx (host-side counter, no sync) */
int nvdr_env_shade_stream_id(nvdr_ctx *ctx, uint64_t *out_host);

/* Per-stage HIP-event timing of the env-shade launches (sample generation, traversal, shading incl. the light-gradient
 * gather), recorded on the launch stream itself into a ring of 512 records, one per (launch, chunk of the ray stream);
 * used by bench.py for the roofline figure of the traversal kernel.  nvdr_env_shade_stage_times sums the recorded
 * launches of one kind (backward = 0 | 1) into ms[3] and returns their number in *count (launches, not chunks); it
 * synchronises on the last recorded event.  Reading does not clear the ring (nvdr_ctx_set_profiling(ctx, 1) does); a launch
 * of more chunks than the ring has records is not recorded at all rather than partially. */
int nvdr_ctx_set_profiling(nvdr_ctx *ctx, int enable);
int nvdr_env_shade_stage_times(nvdr_ctx *ctx, int backward, double *ms, int64_t *count);

/* ---- bilateral_denoiser_fwd/bwd (torch_bindings.cpp:274-319; kernels denoising.cu:14-130).
 * col [N,H,W,3], nrm [N,H,W,3], zdz [N,H,W,2] strided views; out f32 [N,H,W,4] contiguous
 * (rgb*w sum, max(sum w, 1e-4)); out_grad [N,H,W,4] strided; col_grad f32 [N,H,W,3] contiguous. */
int nvdr_bilateral_denoiser_fwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz, float sigma,
                                float *out, void *stream);
int nvdr_bilateral_denoiser_bwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz, float sigma,
                                const nvdr_tensor *out_grad, float *col_grad, void *stream);
/* two images filtered with the same guides in one pass (additive): the weights are evaluated once per tap; bit-identical to two
 * single calls.  render.py:120-121 filters the diffuse and the specular light of shade() with the same normal / depth. */
int nvdr_bilateral_denoiser_pair_fwd(const nvdr_tensor *col_a, const nvdr_tensor *col_b, const nvdr_tensor *nrm, const nvdr_tensor *zdz,
                                     float sigma, float *out_a, float *out_b, void *stream);
int nvdr_bilateral_denoiser_pair_bwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz, float sigma,
                                     const nvdr_tensor *out_grad_a, const nvdr_tensor *out_grad_b, float *col_grad_a, float *col_grad_b,
                                     void *stream);

/* ---- renderutils_plugin (render/renderutils/c_src/torch_bindings.cpp).  All tensors are NHWC
 * 4-d views with size-1 broadcasting; the launch extent is the per-dim max over the inputs
 * (update_grid, torch_bindings.cpp:87-101); outputs and gradients are contiguous at that extent
 * (gradients of broadcast inputs are summed by the caller, tensor.h:60-62). */

/* image_loss_fwd/bwd (torch_bindings.cpp:727-798, loss.cu:105-228).
 * loss: 0 l1, 1 mse, 2 relmse, 3 smape, 4 n2n;  tonemapper: 0 none, 1 log_srgb.
 * fwd writes ONE partial sum per workgroup into `partials` (n_partials from nvdr_image_loss_num_partials);
 * the caller sums and divides by N*H*W (ops.py:494).  bwd: d_partials f32[n_partials] is the gradient
 * w.r.t. each partial (autograd hands back one value per partial; a pixel uses its own partial's).  target_grad may be
 * NULL: the gradient w.r.t. the target is then not written (a constant reference image). */
int64_t nvdr_image_loss_num_partials(int64_t n, int64_t h, int64_t w);
int nvdr_image_loss_fwd(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper, float *partials,
                        void *stream);
int nvdr_image_loss_bwd(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper,
                        const float *d_partials, float *img_grad, float *target_grad, void *stream);
/* The same loss as ONE scalar, mean over the N H W pixels (additive: `torch.sum(out) / (N H W)` of renderutils/ops.py:494 folded in):
 * partials = scratch of nvdr_image_loss_num_partials floats, out_mean f32 [1]; the backward takes the upstream gradient of the mean
 * as a device scalar d_mean f32 [1]. */
int nvdr_image_loss_mean_fwd(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper, float *partials,
                             float *out_mean, void *stream);
int nvdr_image_loss_mean_bwd(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper, const float *d_mean,
                             float *img_grad, float *target_grad, void *stream);

/* prepare_shading_normal_fwd/bwd (torch_bindings.cpp:148-219, normal.cu:95-179) */
int nvdr_prepare_shading_normal_fwd(const nvdr_tensor *pos, const nvdr_tensor *view_pos, const nvdr_tensor *perturbed_nrm,
                                    const nvdr_tensor *smooth_nrm, const nvdr_tensor *smooth_tng,
                                    const nvdr_tensor *geom_nrm, int two_sided_shading, int opengl, float *out,
                                    void *stream);
/* additive, forward only: shading normal + its unit copy (the denoiser's guide, render/util.py safe_normalize) + the shadow-ray origin
 * gb_pos + normal * ro_eps (render.py:107) in one launch */
int nvdr_shading_frame_fwd(const nvdr_tensor *pos, const nvdr_tensor *view_pos, const nvdr_tensor *perturbed_nrm,
                           const nvdr_tensor *smooth_nrm, const nvdr_tensor *smooth_tng, const nvdr_tensor *geom_nrm,
                           int two_sided_shading, int opengl, float ro_eps, float *out_nrm, float *out_unit, float *out_ro, void *stream);
int nvdr_prepare_shading_normal_bwd(const nvdr_tensor *pos, const nvdr_tensor *view_pos, const nvdr_tensor *perturbed_nrm,
                                    const nvdr_tensor *smooth_nrm, const nvdr_tensor *smooth_tng,
                                    const nvdr_tensor *geom_nrm, const nvdr_tensor *d_out, int two_sided_shading,
                                    int opengl, float *pos_grad, float *view_pos_grad, float *perturbed_nrm_grad,
                                    float *smooth_nrm_grad, float *smooth_tng_grad, float *geom_nrm_grad, void *stream);

/* xfm_fwd/bwd (torch_bindings.cpp:803-864, mesh.cu:19-91).  points [1|B,V,3], matrix [B,4,4] (contiguous),
 * out [B,V,4] (is_points) or [B,V,3]; points_grad [B,V,3]. */
int nvdr_xfm_fwd(const float *points, int64_t points_batch, int64_t n_points, const float *matrix, int64_t batch,
                 int is_points, float *out, void *stream);
int nvdr_xfm_bwd(const float *matrix, int64_t batch, int64_t n_points, const float *d_out, int is_points,
                 float *points_grad, void *stream);

/* stand-alone BSDF ops (torch_bindings.cpp:224-722, bsdf.cu:382-707) */
int nvdr_lambert_fwd(const nvdr_tensor *nrm, const nvdr_tensor *wi, float *out, void *stream);
int nvdr_lambert_bwd(const nvdr_tensor *nrm, const nvdr_tensor *wi, const nvdr_tensor *d_out, float *nrm_grad,
                     float *wi_grad, void *stream);
int nvdr_frostbite_fwd(const nvdr_tensor *nrm, const nvdr_tensor *wi, const nvdr_tensor *wo,
                       const nvdr_tensor *linear_roughness, float *out, void *stream);
int nvdr_frostbite_bwd(const nvdr_tensor *nrm, const nvdr_tensor *wi, const nvdr_tensor *wo,
                       const nvdr_tensor *linear_roughness, const nvdr_tensor *d_out, float *nrm_grad, float *wi_grad,
                       float *wo_grad, float *linear_roughness_grad, void *stream);
int nvdr_fresnel_shlick_fwd(const nvdr_tensor *f0, const nvdr_tensor *f90, const nvdr_tensor *cos_theta, float *out,
                            void *stream);
int nvdr_fresnel_shlick_bwd(const nvdr_tensor *f0, const nvdr_tensor *f90, const nvdr_tensor *cos_theta,
                            const nvdr_tensor *d_out, float *f0_grad, float *f90_grad, float *cos_theta_grad,
                            void *stream);
int nvdr_ndf_ggx_fwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta, float *out, void *stream);
int nvdr_ndf_ggx_bwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta, const nvdr_tensor *d_out,
                     float *alpha_sqr_grad, float *cos_theta_grad, void *stream);
int nvdr_lambda_ggx_fwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta, float *out, void *stream);
int nvdr_lambda_ggx_bwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta, const nvdr_tensor *d_out,
                        float *alpha_sqr_grad, float *cos_theta_grad, void *stream);
int nvdr_masking_smith_fwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta_i, const nvdr_tensor *cos_theta_o,
                           float *out, void *stream);
int nvdr_masking_smith_bwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta_i, const nvdr_tensor *cos_theta_o,
                           const nvdr_tensor *d_out, float *alpha_sqr_grad, float *cos_theta_i_grad,
                           float *cos_theta_o_grad, void *stream);
int nvdr_pbr_specular_fwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *wo, const nvdr_tensor *wi,
                          const nvdr_tensor *alpha, float min_roughness, float *out, void *stream);
int nvdr_pbr_specular_bwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *wo, const nvdr_tensor *wi,
                          const nvdr_tensor *alpha, float min_roughness, const nvdr_tensor *d_out, float *col_grad,
                          float *nrm_grad, float *wo_grad, float *wi_grad, float *alpha_grad, void *stream);
int nvdr_pbr_bsdf_fwd(const nvdr_tensor *kd, const nvdr_tensor *arm, const nvdr_tensor *pos, const nvdr_tensor *nrm,
                      const nvdr_tensor *view_pos, const nvdr_tensor *light_pos, float min_roughness, int bsdf,
                      float *out, void *stream);
int nvdr_pbr_bsdf_bwd(const nvdr_tensor *kd, const nvdr_tensor *arm, const nvdr_tensor *pos, const nvdr_tensor *nrm,
                      const nvdr_tensor *view_pos, const nvdr_tensor *light_pos, float min_roughness, int bsdf,
                      const nvdr_tensor *d_out, float *kd_grad, float *arm_grad, float *pos_grad, float *nrm_grad,
                      float *view_pos_grad, float *light_pos_grad, void *stream);

/* ---- fused shading composite (additive; replaces the torch expressions of render/render.py:119-127 and the division of
 * render/optixutils/ops.py:139-141):  out = (diff.rgb / diff.w) * kd * (1 - ks.z) + spec.rgb / spec.w  for bsdf 0 ('pbr'),
 * (diff.rgb / diff.w) * kd for bsdf 1 ('diffuse' / 'white').  diff, spec: f32 [N,H,W,4] (colour sum, weight: the raw
 * bilateral_denoiser_fwd output) or [N,H,W,3] (already normalised, weight 1); kd, ks [N,H,W,3] broadcastable.
 * Gradients are contiguous at the full extent with the channel count of their input. */
int nvdr_shade_composite_fwd(const nvdr_tensor *diff, const nvdr_tensor *spec, const nvdr_tensor *kd, const nvdr_tensor *ks,
                             int bsdf, float *out, void *stream);
int nvdr_shade_composite_bwd(const nvdr_tensor *diff, const nvdr_tensor *spec, const nvdr_tensor *kd, const nvdr_tensor *ks,
                             int bsdf, const nvdr_tensor *d_out, float *diff_grad, float *spec_grad, float *kd_grad,
                             float *ks_grad, void *stream);

/* Composite + mean image loss, forward and backward in ONE launch (additive, round 6: the tail of shade() and train.py:51-66 for a caller whose
 * loss is the mean image loss of the composite): out_mean f32 [1] = mean over N H W of the loss of (composite(diff, spec, kd, ks), target), and
 * the gradients of that mean -- times the upstream gradient *d_mean, a device scalar read by the launch -- with respect to diff, spec (channel
 * count of their input), kd and ks (full extent).  partials: scratch of nvdr_image_loss_num_partials floats.  Same arithmetic as
 * nvdr_shade_composite_fwd/bwd + nvdr_image_loss_mean_fwd/bwd, statement for statement. */
int nvdr_shade_loss_fused(const nvdr_tensor *diff, const nvdr_tensor *spec, const nvdr_tensor *kd, const nvdr_tensor *ks, int bsdf,
                          const nvdr_tensor *target, int loss, int tonemapper, const float *d_mean, float *partials, float *out_mean,
                          float *diff_grad, float *spec_grad, float *kd_grad, float *ks_grad, void *stream);

/* ---- rows of a table by index (additive; the nearest-texel lookup of a trained texture in the iteration harness):
 * out[i,:] = index[i] >= 0 ? table[index[i],:] : 0; the backward zeroes dtable [table_rows, channels] and accumulates. */
int nvdr_gather_rows_fwd(const float *table, const int *index, int64_t n, int channels, float *out, void *stream);
int nvdr_gather_rows_bwd(const float *dout, const int *index, int64_t n, int channels, int64_t table_rows, float *dtable, void *stream);

/* ---- EnvironmentLight.update_pdf (render/light.py:46-59) fused on device: base f32 [Hl,Wl,3] contiguous ->
 * pdf [Hl,Wl], cols [Hl,Wl], rows [Hl] (the reference materialises rows as [Hl,Wl] with identical
 * columns and passes rows[:,0], render.py:114). */
int nvdr_light_update_pdf(const float *base, int64_t hl, int64_t wl, float *pdf, float *cols, float *rows,
                          void *stream);

/* ---- the parameter update of an iteration in one launch (additive; train.py:439-476: light-gradient scale, torch.optim.Adam
 * without weight decay / amsgrad, parameter clamps).  Per element: g = grad * grad_scale; Adam with bias correction at step
 * state[0] + 1; p = clamp(p, max(lo, lo_vec[e % lo_vec_n]), min(hi, hi_vec[e % hi_vec_n])) (a NaN stays a NaN, as torch.clamp).  state: 32 bytes of device memory, 8-byte aligned,
 * zero-initialised by the caller once (int32 [0] = steps taken, [1] = scratch, then two doubles: beta1^step, beta2^step); the
 * launch advances it, so it can be replayed from a HIP graph. */
#define NVDR_ADAM_MAX_TENSORS 8
typedef struct nvdr_adam_tensor {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t n;              /* elements (all four buffers contiguous f32) */
    float grad_scale;       /* 1 for none */
    float lo, hi;           /* -INFINITY / INFINITY for none */
    const float *lo_vec;    /* optional per-channel lower bounds (device), NULL for none */
    int64_t lo_vec_n;
    const float *hi_vec;    /* optional per-channel upper bounds (device), NULL for none (Texture2D.clamp_, render/texture.py:86-90) */
    int64_t hi_vec_n;
    float lr_scale;         /* this tensor's learning rate = lr * lr_scale (train.py:336-338: position / material / light rates); 0 (a
                               zero-initialised block) means 1.  To freeze a tensor set `frozen`, not a zero rate */
    int32_t normalize3;     /* != 0: after the clamps every group of three elements is divided by max(its length, 1e-10)
                               (Texture2D.normalize_ of the normal map, train.py:473-474) */
    uint8_t *active;        /* optional (n % 3 == 0): one byte per tile of 64 three-channel texels, zero-initialised by the caller once.  Tiles
                               whose gradient is all zero and that never had a non-zero one are skipped after step 1 (their update is the
                               identity): the sparse gradients of a nearest-texel texture lookup.  NULL: every element is updated */
    int32_t zero_grad;      /* with `active`: the gradient of every updated tile is zeroed behind the update (`grad` is written) */
    int32_t frozen;         /* != 0: the tensor is left alone altogether (parameter, moments and gradient untouched) */
} nvdr_adam_tensor;
/* sizeof of the argument blocks as THIS build of the library sees them: a caller compiled (or a ctypes mirror written) against another
 * revision of this header compares before its first call.  which: 0 nvdr_adam_tensor, 1 nvdr_env_shade_args, 2 nvdr_texture_args,
 * 3 nvdr_interpolate_bwd_args, 4 nvdr_tensor, 5 nvdr_gbuffer_args, 6 nvdr_mesh_args, 7 nvdr_bvh_info; anything else returns 0. */
size_t nvdr_abi_sizeof(int which);
int nvdr_adam_step(const nvdr_adam_tensor *tensors, int n_tensors, double lr, double beta1, double beta2, double eps, int *state,
                   void *stream);
/* The same update for a SUBSET of an iteration's tensors (one chunk of the data-parallel gradient exchange: the update of chunk k
 * runs while chunk k + 1 is still being reduced).  Every launch of an iteration uses step state[0] + 1; only the one with
 * advance != 0 -- the iteration's last -- moves the counter. */
int nvdr_adam_step_partial(const nvdr_adam_tensor *tensors, int n_tensors, double lr, double beta1, double beta2, double eps, int *state,
                           int advance, void *stream);

/* ---- tile-sparse gradient exchange (additive; the reference has no distributed code).  The data-parallel step all-reduces the
 * gradients of the trained textures; with nearest-texel lookups they are zero outside the tiles (tile_floats contiguous floats, e.g.
 * 64 texels x 3 channels) a rank's pixels touched.  flags: one byte per tile, 1 = the tile holds a non-zero value (or a NaN).  After a
 * MAX all-reduce of the flags over the ranks, nvdr_tile_plan lists the flagged tiles in ascending order (list [n_tiles] int32, *count =
 * how many; device memory, identical on every rank), nvdr_tile_gather copies them into `compact` ([count, tile_floats], what the SUM
 * all-reduce sends), nvdr_tile_scatter copies the sums back to their places in the dense buffer.  Buffers 16-byte aligned, tile_floats a
 * multiple of 4; the count is read on the device by gather / scatter. */
int nvdr_tile_flags(const float *grad, int64_t n_tiles, int tile_floats, uint8_t *flags, void *stream);
int nvdr_tile_plan(const uint8_t *flags, int64_t n_tiles, int32_t *list, int32_t *count, void *stream);
int nvdr_tile_gather(const float *dense, const int32_t *list, const int32_t *count, int64_t n_tiles, int tile_floats, float *compact, void *stream);
int nvdr_tile_scatter(const float *compact, const int32_t *list, const int32_t *count, int64_t n_tiles, int tile_floats, float *dense, void *stream);

/* ---- test hook: evaluate include/nvdr_detmath.h on device.  op: 0 sin, 1 cos, 2 acos, 3 atan2(x,y). */
int nvdr_test_detmath(int op, const float *x, const float *y, int64_t n, float *out, void *stream);
/* ---- test hook: csrc/ieee_arith.h (the correctly rounded division / square root of the shading kernels without the compiler's range
 * scaling) against the compiler's own `/`, sqrtf, sqrt on the device: every float through the square root, 3 * 2^30 divisions inside
 * the documented domain, the special values, fp64 on doubles made of floats.  counters: 7 uint64 in device memory (zeroed here):
 * [0] sqrt mismatches in the domain, [1] among positive inputs below 2^-96 and negative denormals (outside it), [2] division, [3] special values, [4] fp64 division, [5] fp64 square
 * root, [6] comparisons made.  [0], [2]-[5] must read 0. */
int nvdr_test_arith(unsigned long long *counters, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NVDR_HIP_H */
