// renderutils.hip -- the element-wise operators of renderutils_plugin for gfx950.
//
// Replaces render/renderutils/c_src/{loss.cu, normal.cu, mesh.cu, bsdf.cu} and their launchers in
// render/renderutils/c_src/torch_bindings.cpp:148-864.  All operators are HBM-streaming: one
// thread per pixel over the broadcast extent (per-dim max over the inputs, update_grid
// torch_bindings.cpp:87-101), strided/broadcast reads, contiguous writes.  Gradients of broadcast
// inputs are written at the full extent and summed by the caller (tensor.h:60-62).
// The reference's 8x4 "warp" tiles (common.cpp:53-60, 32 lanes baked in) are replaced by wave64
// reductions; the partial-sum tensor shape of image_loss is an internal detail (ops.py:494 sums it).
#include "common.h"
#include "bsdf_device.h"

struct Extent {
    int N, H, W;
    int64_t total;
};

template <typename... Ts> static Extent make_extent(const Ts *...ts)
{
    Extent e = {1, 1, 1, 1};
    const nvdr_tensor *arr[] = {ts...};
    for (const nvdr_tensor *t : arr) {
        e.N = (int)max64(e.N, t->size[0]);
        e.H = (int)max64(e.H, t->size[1]);
        e.W = (int)max64(e.W, t->size[2]);
    }
    e.total = (int64_t)e.N * e.H * e.W;
    return e;
}

static int check_view(const nvdr_tensor *t, const Extent &e, int channels, const char *op, const char *name)
{
    NVDR_REQUIRE(t && t->data, "%s: %s is NULL", op, name);
    NVDR_REQUIRE((t->size[0] == e.N || t->size[0] == 1) && (t->size[1] == e.H || t->size[1] == 1) &&
                     (t->size[2] == e.W || t->size[2] == 1),
                 "%s: %s with shape [%lld,%lld,%lld,%lld] does not broadcast to [%d,%d,%d,*]", op, name,
                 (long long)t->size[0], (long long)t->size[1], (long long)t->size[2], (long long)t->size[3], e.N, e.H, e.W);
    NVDR_REQUIRE(t->size[3] == channels || t->size[3] == 1, "%s: %s must have %d channels (got %lld)", op, name, channels,
                 (long long)t->size[3]);
    return 0;
}
#define CHECK_VIEW(t, c) do { int _r = check_view(t, e, c, OP, #t); if (_r) return _r; } while (0)

template <class F> __global__ void __launch_bounds__(256) ew_kernel(Extent e, F f)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e.total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % e.W), h = (int)((i / e.W) % e.H), n = (int)(i / ((int64_t)e.W * e.H));
        f(n, h, w, i);
    }
}
template <class F> static int launch_ew(const Extent &e, hipStream_t stream, F f)
{
    if (e.total <= 0) return 0;
    unsigned blocks = div_up(e.total, 256);
    if (blocks > 8192) blocks = 8192;
    ew_kernel<<<blocks, 256, 0, stream>>>(e, f);
    NVDR_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ void store3(float *p, int64_t i, F3 v)
{
    p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z;
}

// =============================================================================================
// image loss (loss.cu:17-228)

__device__ __forceinline__ float bwd_abs(float x) { return x == 0.0f ? 0.0f : (x < 0.0f ? -1.0f : 1.0f); }
__device__ __forceinline__ float fwd_srgb(float x)
{
    return x > 0.0031308f ? powf(fmaxf(x, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f : 12.92f * fmaxf(x, 0.0f);
}
__device__ __forceinline__ void bwd_srgb(float x, float &d_x, float d_out)
{
    if (x > 0.0031308f)
        d_x += d_out * 0.439583f / powf(x, 0.583333f);
    else if (x > 0.0f)
        d_x += d_out * 12.92f;
}
__device__ __forceinline__ float tonemap_fwd(float x) { return fwd_srgb(logf(x + 1.0f)); }
__device__ __forceinline__ float tonemap_bwd(float x, float d_out)
{
    float d = 0.0f;
    if (x > 0.0f && x < 65535.0f) {
        bwd_srgb(logf(x + 1.0f), d, d_out);
        d *= 1 / (x + 1.0f);
    }
    return d;
}
enum { LOSS_L1 = 0, LOSS_MSE = 1, LOSS_RELMSE = 2, LOSS_SMAPE = 3, LOSS_N2N = 4 };

__device__ __forceinline__ float loss_fwd1(int loss, float img, float target)
{
    const float eps = 0.01f;
    switch (loss) {
    case LOSS_MSE: return (img - target) * (img - target);
    case LOSS_RELMSE: return (img - target) * (img - target) / (img * img + target * target + eps);
    case LOSS_SMAPE: return fabsf(img - target) / (img + target + eps);
    case LOSS_N2N: return (img - target) * (img - target) / (img * img + eps);
    default: return fabsf(img - target);
    }
}
__device__ __forceinline__ void loss_bwd1(int loss, float img, float target, float d_out, float &d_img, float &d_target)
{
    const float eps = 0.01f;
    switch (loss) {
    case LOSS_MSE:
        d_img = d_out * 2 * (img - target);
        d_target = -d_img;
        break;
    case LOSS_RELMSE: {
        const float denom = (target * target + img * img + eps);
        d_img = d_out * 2 * (img - target) * (target * (target + img) + eps) / (denom * denom);
        d_target = -(d_out * 2 * (img - target) * (img * (target + img) + eps) / (denom * denom));
        break;
    }
    case LOSS_SMAPE: {
        const float denom = (target + img + eps);
        d_img = d_out * bwd_abs(img - target) * (2 * target + eps) / (denom * denom);
        d_target = -(d_out * bwd_abs(img - target) * (2 * img + eps) / (denom * denom));
        break;
    }
    case LOSS_N2N: {
        const float denom = (img * img + eps);
        d_img = d_out * 2 * (img - target) / denom;
        d_target = -(d_out * 2 * (img - target) / denom);
        break;
    }
    default:
        d_img = d_out * bwd_abs(img - target);
        d_target = -d_img;
    }
}

#define LOSS_BLOCK 256
#define LOSS_MAX_PARTIALS 1024

extern "C" int64_t nvdr_image_loss_num_partials(int64_t n, int64_t h, int64_t w)
{
    int64_t b = (n * h * w + LOSS_BLOCK - 1) / LOSS_BLOCK;
    if (b < 1) b = 1;
    return b > LOSS_MAX_PARTIALS ? LOSS_MAX_PARTIALS : b;
}

__global__ void __launch_bounds__(LOSS_BLOCK) image_loss_fwd_kernel(Extent e, View4 img, View4 target, int loss,
                                                                     int tonemapper, float *__restrict__ partials)
{
    __shared__ float wsum[LOSS_BLOCK / 64];
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e.total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % e.W), h = (int)((i / e.W) % e.H), n = (int)(i / ((int64_t)e.W * e.H));
        F3 a = fetch3(img, n, h, w), b = fetch3(target, n, h, w);
        float av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x = clampf(av[c], 0.0f, 65535.0f), t = clampf(bv[c], 0.0f, 65535.0f);
            if (tonemapper) {
                x = tonemap_fwd(x);
                t = tonemap_fwd(t);
            }
            s += loss_fwd1(loss, x, t);
        }
        acc += s / 3.0f;
    }
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int k = 0; k < LOSS_BLOCK / 64; ++k) t += wsum[k];
        partials[blockIdx.x] = t;
    }
}

extern "C" int nvdr_image_loss_fwd(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper,
                                   float *partials, void *stream)
{
    static const char *OP = "image_loss_fwd";
    NVDR_REQUIRE(img && target && partials, "%s: NULL argument", OP);
    NVDR_REQUIRE(loss >= 0 && loss <= 4 && (tonemapper == 0 || tonemapper == 1), "%s: bad loss/tonemapper", OP);
    const Extent e = make_extent(img, target);
    CHECK_VIEW(img, 3);
    CHECK_VIEW(target, 3);
    const unsigned blocks = (unsigned)nvdr_image_loss_num_partials(e.N, e.H, e.W);
    image_loss_fwd_kernel<<<blocks, LOSS_BLOCK, 0, (hipStream_t)stream>>>(e, make_view4(*img), make_view4(*target), loss,
                                                                           tonemapper, partials);
    NVDR_LAUNCH_CHECK();
    return 0;
}

static int image_loss_bwd_impl(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper,
                               const float *d_partials, const float *d_mean, float mean_scale, float *img_grad, float *target_grad, void *stream)
{
    static const char *OP = "image_loss_bwd";
    // target_grad may be NULL (a target that does not require a gradient: the reference image of a training iteration)
    NVDR_REQUIRE(img && target && (d_partials || d_mean) && img_grad, "%s: NULL argument", OP);
    const Extent e = make_extent(img, target);
    CHECK_VIEW(img, 3);
    CHECK_VIEW(target, 3);
    const View4 vi = make_view4(*img), vt = make_view4(*target);
    const int64_t stride = nvdr_image_loss_num_partials(e.N, e.H, e.W) * LOSS_BLOCK;
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        // pixel i was summed into partial ((i mod grid span) / block) by the forward kernel
        // (mean variant: every pixel has the same upstream gradient, d(mean) / (N H W), read from ONE device scalar)
        const float d_out = d_mean ? d_mean[0] * mean_scale : d_partials[(i % stride) / LOSS_BLOCK];
        const F3 a = fetch3(vi, n, h, w), b = fetch3(vt, n, h, w);
        const float av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
        float gi[3], gt[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x = av[c], t = bv[c];
            if (tonemapper) { // the reference tonemaps the UNclamped value in the backward kernel (loss.cu:166-171)
                x = tonemap_fwd(x);
                t = tonemap_fwd(t);
            }
            float dx, dt;
            loss_bwd1(loss, x, t, d_out / 3.0f, dx, dt);
            if (tonemapper) {
                dx = tonemap_bwd(av[c], dx);
                dt = tonemap_bwd(bv[c], dt);
            }
            if (av[c] <= 0.0f || av[c] >= 65535.0f) dx = 0;
            if (bv[c] <= 0.0f || bv[c] >= 65535.0f) dt = 0;
            gi[c] = dx;
            gt[c] = dt;
        }
        store3(img_grad, i, f3(gi[0], gi[1], gi[2]));
        if (target_grad) store3(target_grad, i, f3(gt[0], gt[1], gt[2]));
    });
}

extern "C" int nvdr_image_loss_bwd(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper,
                                   const float *d_partials, float *img_grad, float *target_grad, void *stream)
{
    return image_loss_bwd_impl(img, target, loss, tonemapper, d_partials, nullptr, 0.0f, img_grad, target_grad, stream);
}

// ---- the loss as ONE scalar (additive): what `torch.sum(partials) / (N H W)` of renderutils/ops.py:494 and its adjoint cost five
// small torch kernels for.  Forward: the partial sums + one workgroup that adds them in a fixed order; backward: the upstream
// gradient is a device scalar, nothing is expanded.
__global__ void __launch_bounds__(256) image_loss_reduce_kernel(const float *__restrict__ partials, int64_t n, float scale, float *__restrict__ out)
{
    __shared__ float red[4];
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += partials[i];
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

extern "C" int nvdr_image_loss_mean_fwd(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper, float *partials,
                                        float *out_mean, void *stream)
{
    NVDR_REQUIRE(out_mean, "image_loss_mean_fwd: NULL output");
    if (int r = nvdr_image_loss_fwd(img, target, loss, tonemapper, partials, stream)) return r;
    const Extent e = make_extent(img, target);
    const int64_t n = nvdr_image_loss_num_partials(e.N, e.H, e.W);
    image_loss_reduce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(partials, n, 1.0f / (float)((int64_t)e.N * e.H * e.W), out_mean);
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_image_loss_mean_bwd(const nvdr_tensor *img, const nvdr_tensor *target, int loss, int tonemapper,
                                        const float *d_mean, float *img_grad, float *target_grad, void *stream)
{
    NVDR_REQUIRE(d_mean, "image_loss_mean_bwd: NULL upstream gradient");
    const Extent e = make_extent(img, target);
    return image_loss_bwd_impl(img, target, loss, tonemapper, nullptr, d_mean, 1.0f / (float)((int64_t)e.N * e.H * e.W), img_grad, target_grad, stream);
}

// =============================================================================================
// prepare_shading_normal (normal.cu:17-179)

#define NVDR_NORMAL_THRESHOLD 0.1f

__device__ __forceinline__ F3 fwd_perturb_normal(F3 pn, F3 sn, F3 st, bool opengl)
{
    const F3 bit = safe_normalize(cross3(st, sn));
    const F3 sh = st * pn.x + ((opengl ? -1.0f : 1.0f) * bit) * pn.y + sn * fmaxf(pn.z, 0.0f);
    return safe_normalize(sh);
}
__device__ __forceinline__ void bwd_perturb_normal(F3 pn, F3 sn, F3 st, F3 &d_pn, F3 &d_sn, F3 &d_st, F3 d_out, bool opengl)
{
    const F3 _bit = cross3(st, sn);
    const F3 bit = safe_normalize(_bit);
    const float sg = opengl ? -1.0f : 1.0f;
    const F3 _sh = st * pn.x + (sg * bit) * pn.y + sn * fmaxf(pn.z, 0.0f);
    F3 d_sh = f3(0.0f);
    bwd_safe_normalize(_sh, d_sh, d_out);
    F3 d_bit = f3(0.0f);
    if (pn.z > 0.0f) {
        d_sn += d_sh * pn.z;
        d_pn.z += sum3(d_sh * sn);
    }
    d_bit += (sg * d_sh) * pn.y;
    d_pn.y += sg * sum3(d_sh * bit);
    d_st += d_sh * pn.x;
    d_pn.x += sum3(d_sh * st);
    F3 d__bit = f3(0.0f);
    bwd_safe_normalize(_bit, d__bit, d_bit);
    bwd_cross(st, sn, d_st, d_sn, d__bit);
}
__device__ __forceinline__ F3 fwd_bend_normal(F3 view, F3 sn, F3 gn)
{
    const float dp = dot3(view, sn);
    const float t = clampf(dp / NVDR_NORMAL_THRESHOLD, 0.0f, 1.0f);
    return gn * (1.0f - t) + sn * t;
}
__device__ __forceinline__ void bwd_bend_normal(F3 view, F3 sn, F3 gn, F3 &d_view, F3 &d_sn, F3 &d_gn, F3 d_out)
{
    const float dp = dot3(view, sn);
    const float t = clampf(dp / NVDR_NORMAL_THRESHOLD, 0.0f, 1.0f);
    if (dp > NVDR_NORMAL_THRESHOLD) {
        d_sn += d_out;
    } else {
        d_gn += d_out * (1.0f - t);
        d_sn += d_out * t;
        const float d_t = sum3(d_out * (sn - gn));
        const float d_dp = (dp < 0.0f || dp > NVDR_NORMAL_THRESHOLD) ? 0.0f : d_t / NVDR_NORMAL_THRESHOLD;
        bwd_dot(view, sn, d_view, d_sn, d_dp);
    }
}

extern "C" int nvdr_prepare_shading_normal_fwd(const nvdr_tensor *pos, const nvdr_tensor *view_pos,
                                               const nvdr_tensor *perturbed_nrm, const nvdr_tensor *smooth_nrm,
                                               const nvdr_tensor *smooth_tng, const nvdr_tensor *geom_nrm,
                                               int two_sided_shading, int opengl, float *out, void *stream)
{
    static const char *OP = "prepare_shading_normal_fwd";
    NVDR_REQUIRE(pos && view_pos && perturbed_nrm && smooth_nrm && smooth_tng && geom_nrm && out, "%s: NULL argument", OP);
    const Extent e = make_extent(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm);
    CHECK_VIEW(pos, 3); CHECK_VIEW(view_pos, 3); CHECK_VIEW(perturbed_nrm, 3);
    CHECK_VIEW(smooth_nrm, 3); CHECK_VIEW(smooth_tng, 3); CHECK_VIEW(geom_nrm, 3);
    const View4 vp = make_view4(*pos), vv = make_view4(*view_pos), vpn = make_view4(*perturbed_nrm),
                vsn = make_view4(*smooth_nrm), vst = make_view4(*smooth_tng), vgn = make_view4(*geom_nrm);
    const bool two = two_sided_shading != 0, ogl = opengl != 0;
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        const F3 p = fetch3(vp, n, h, w), vw = fetch3(vv, n, h, w), pn = fetch3(vpn, n, h, w);
        const F3 sn = safe_normalize(fetch3(vsn, n, h, w)), st = safe_normalize(fetch3(vst, n, h, w));
        const F3 gn = fetch3(vgn, n, h, w);
        const F3 view = safe_normalize(vw - p);
        const F3 sh = fwd_perturb_normal(pn, sn, st, ogl);
        F3 res;
        if (two && dot3(view, gn) < 0.0f)
            res = fwd_bend_normal(view, -sh, -gn);
        else
            res = fwd_bend_normal(view, sh, gn);
        store3(out, i, res);
    });
}

// The shading frame of shade() in one launch (additive, forward only: the harness keeps the geometry fixed): the shading normal as
// above, its normalised copy for the denoiser's guides (render/util.py safe_normalize: x / sqrt(max(dot(x, x), 1e-20))) and the
// shadow-ray origin gb_pos + normal * ro_eps (render.py:107) -- eight small torch kernels otherwise.
extern "C" int nvdr_shading_frame_fwd(const nvdr_tensor *pos, const nvdr_tensor *view_pos, const nvdr_tensor *perturbed_nrm,
                                      const nvdr_tensor *smooth_nrm, const nvdr_tensor *smooth_tng, const nvdr_tensor *geom_nrm,
                                      int two_sided_shading, int opengl, float ro_eps, float *out_nrm, float *out_unit, float *out_ro,
                                      void *stream)
{
    static const char *OP = "shading_frame_fwd";
    NVDR_REQUIRE(pos && view_pos && perturbed_nrm && smooth_nrm && smooth_tng && geom_nrm && out_nrm && out_unit && out_ro, "%s: NULL argument", OP);
    const Extent e = make_extent(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm);
    CHECK_VIEW(pos, 3); CHECK_VIEW(view_pos, 3); CHECK_VIEW(perturbed_nrm, 3);
    CHECK_VIEW(smooth_nrm, 3); CHECK_VIEW(smooth_tng, 3); CHECK_VIEW(geom_nrm, 3);
    const View4 vp = make_view4(*pos), vv = make_view4(*view_pos), vpn = make_view4(*perturbed_nrm),
                vsn = make_view4(*smooth_nrm), vst = make_view4(*smooth_tng), vgn = make_view4(*geom_nrm);
    const bool two = two_sided_shading != 0, ogl = opengl != 0;
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        const F3 p = fetch3(vp, n, h, w), vw = fetch3(vv, n, h, w), pn = fetch3(vpn, n, h, w);
        const F3 sn = safe_normalize(fetch3(vsn, n, h, w)), st = safe_normalize(fetch3(vst, n, h, w));
        const F3 gn = fetch3(vgn, n, h, w);
        const F3 view = safe_normalize(vw - p);
        const F3 sh = fwd_perturb_normal(pn, sn, st, ogl);
        F3 res;
        if (two && dot3(view, gn) < 0.0f)
            res = fwd_bend_normal(view, -sh, -gn);
        else
            res = fwd_bend_normal(view, sh, gn);
        store3(out_nrm, i, res);
        // (x^2 + z^2) + y^2: the order torch.sum takes over three elements (one per lane, shuffle-down by 2 then by 1), so that the unit
        // normal is bit-identical to the composed torch expression -- the filter raises dot products of it to the 128th power
        const float len = sqrtf(fmaxf((res.x * res.x + res.z * res.z) + res.y * res.y, 1e-20f));
        store3(out_unit, i, f3(res.x / len, res.y / len, res.z / len));
        store3(out_ro, i, f3(p.x + res.x * ro_eps, p.y + res.y * ro_eps, p.z + res.z * ro_eps));
    });
}

extern "C" int nvdr_prepare_shading_normal_bwd(const nvdr_tensor *pos, const nvdr_tensor *view_pos,
                                               const nvdr_tensor *perturbed_nrm, const nvdr_tensor *smooth_nrm,
                                               const nvdr_tensor *smooth_tng, const nvdr_tensor *geom_nrm,
                                               const nvdr_tensor *d_out, int two_sided_shading, int opengl,
                                               float *pos_grad, float *view_pos_grad, float *perturbed_nrm_grad,
                                               float *smooth_nrm_grad, float *smooth_tng_grad, float *geom_nrm_grad,
                                               void *stream)
{
    static const char *OP = "prepare_shading_normal_bwd";
    NVDR_REQUIRE(pos && view_pos && perturbed_nrm && smooth_nrm && smooth_tng && geom_nrm && d_out, "%s: NULL argument", OP);
    NVDR_REQUIRE(pos_grad && view_pos_grad && perturbed_nrm_grad && smooth_nrm_grad && smooth_tng_grad && geom_nrm_grad,
                 "%s: NULL output", OP);
    const Extent e = make_extent(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm);
    CHECK_VIEW(pos, 3); CHECK_VIEW(view_pos, 3); CHECK_VIEW(perturbed_nrm, 3);
    CHECK_VIEW(smooth_nrm, 3); CHECK_VIEW(smooth_tng, 3); CHECK_VIEW(geom_nrm, 3); CHECK_VIEW(d_out, 3);
    const View4 vp = make_view4(*pos), vv = make_view4(*view_pos), vpn = make_view4(*perturbed_nrm),
                vsn = make_view4(*smooth_nrm), vst = make_view4(*smooth_tng), vgn = make_view4(*geom_nrm),
                vdo = make_view4(*d_out);
    const bool two = two_sided_shading != 0, ogl = opengl != 0;
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        const F3 p = fetch3(vp, n, h, w), vw = fetch3(vv, n, h, w), pn = fetch3(vpn, n, h, w);
        const F3 _sn = fetch3(vsn, n, h, w), _st = fetch3(vst, n, h, w), gn = fetch3(vgn, n, h, w);
        const F3 dout = fetch3(vdo, n, h, w);
        const F3 sn = safe_normalize(_sn), st = safe_normalize(_st);
        const F3 _view = vw - p;
        const F3 view = safe_normalize(_view);
        const F3 sh = fwd_perturb_normal(pn, sn, st, ogl);
        F3 d_view = f3(0.0f), d_sh = f3(0.0f), d_gn = f3(0.0f);
        if (two && dot3(view, gn) < 0.0f) {
            bwd_bend_normal(view, -sh, -gn, d_view, d_sh, d_gn, dout);
            d_sh = -d_sh;
            d_gn = -d_gn;
        } else {
            bwd_bend_normal(view, sh, gn, d_view, d_sh, d_gn, dout);
        }
        F3 d_pn = f3(0.0f), d_sn = f3(0.0f), d_st = f3(0.0f);
        bwd_perturb_normal(pn, sn, st, d_pn, d_sn, d_st, d_sh, ogl);
        F3 d__view = f3(0.0f), d__sn = f3(0.0f), d__st = f3(0.0f);
        bwd_safe_normalize(_view, d__view, d_view);
        bwd_safe_normalize(_sn, d__sn, d_sn);
        bwd_safe_normalize(_st, d__st, d_st);
        store3(pos_grad, i, -d__view);
        store3(view_pos_grad, i, d__view);
        store3(perturbed_nrm_grad, i, d_pn);
        store3(smooth_nrm_grad, i, d__sn);
        store3(smooth_tng_grad, i, d__st);
        store3(geom_nrm_grad, i, d_gn);
    });
}

// =============================================================================================
// xfm_points / xfm_vectors (mesh.cu:19-91): out[b,v,:] = (points[b|0,v,:], 1|0) * matrix[b]^T

__global__ void __launch_bounds__(256) xfm_fwd_kernel(const float *__restrict__ points, int64_t pts_batch_stride,
                                                       int64_t n_points, const float *__restrict__ matrix, int is_points,
                                                       float *__restrict__ out)
{
    __shared__ float m[16];
    const int b = blockIdx.y;
    if (threadIdx.x < 16) m[threadIdx.x] = matrix[16 * b + threadIdx.x];
    __syncthreads();
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_points) return;
    const float *p = points + b * pts_batch_stride + 3 * v;
    const float x = p[0], y = p[1], z = p[2];
    // row r of the output = dot(matrix[r, 0:3], p) (+ matrix[r, 3] for points); same order as mesh.cu:42-53
    if (is_points) {
        float *o = out + ((int64_t)b * n_points + v) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = x * m[4 * r + 0] + y * m[4 * r + 1] + z * m[4 * r + 2] + m[4 * r + 3];
    } else {
        float *o = out + ((int64_t)b * n_points + v) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = x * m[4 * r + 0] + y * m[4 * r + 1] + z * m[4 * r + 2];
    }
}
__global__ void __launch_bounds__(256) xfm_bwd_kernel(const float *__restrict__ matrix, int64_t n_points,
                                                       const float *__restrict__ d_out, int is_points,
                                                       float *__restrict__ points_grad)
{
    __shared__ float m[16];
    const int b = blockIdx.y;
    if (threadIdx.x < 16) m[threadIdx.x] = matrix[16 * b + threadIdx.x];
    __syncthreads();
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_points) return;
    float *g = points_grad + ((int64_t)b * n_points + v) * 3;
    if (is_points) {
        const float *d = d_out + ((int64_t)b * n_points + v) * 4;
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = d[0] * m[c] + d[1] * m[4 + c] + d[2] * m[8 + c] + d[3] * m[12 + c];
    } else {
        const float *d = d_out + ((int64_t)b * n_points + v) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = d[0] * m[c] + d[1] * m[4 + c] + d[2] * m[8 + c];
    }
}

extern "C" int nvdr_xfm_fwd(const float *points, int64_t points_batch, int64_t n_points, const float *matrix,
                            int64_t batch, int is_points, float *out, void *stream)
{
    NVDR_REQUIRE(points && matrix && out, "xfm_fwd: NULL argument");
    NVDR_REQUIRE(points_batch == 1 || points_batch == batch, "xfm_fwd: points batch %lld does not broadcast to %lld",
                 (long long)points_batch, (long long)batch);
    if (n_points <= 0 || batch <= 0) return 0;
    dim3 grid(div_up(n_points, 256), (unsigned)batch);
    xfm_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(points, points_batch == 1 ? 0 : 3 * n_points, n_points, matrix,
                                                          is_points, out);
    NVDR_LAUNCH_CHECK();
    return 0;
}
extern "C" int nvdr_xfm_bwd(const float *matrix, int64_t batch, int64_t n_points, const float *d_out, int is_points,
                            float *points_grad, void *stream)
{
    NVDR_REQUIRE(matrix && d_out && points_grad, "xfm_bwd: NULL argument");
    if (n_points <= 0 || batch <= 0) return 0;
    dim3 grid(div_up(n_points, 256), (unsigned)batch);
    xfm_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(matrix, n_points, d_out, is_points, points_grad);
    NVDR_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// stand-alone BSDF operators (bsdf.cu:382-707)

extern "C" int nvdr_lambert_fwd(const nvdr_tensor *nrm, const nvdr_tensor *wi, float *out, void *stream)
{
    static const char *OP = "lambert_fwd";
    NVDR_REQUIRE(nrm && wi && out, "%s: NULL argument", OP);
    const Extent e = make_extent(nrm, wi);
    CHECK_VIEW(nrm, 3); CHECK_VIEW(wi, 3);
    const View4 a = make_view4(*nrm), b = make_view4(*wi);
    return launch_ew(e, (hipStream_t)stream,
                     [=] __device__(int n, int h, int w, int64_t i) { out[i] = fwd_lambert(fetch3(a, n, h, w), fetch3(b, n, h, w)); });
}
extern "C" int nvdr_lambert_bwd(const nvdr_tensor *nrm, const nvdr_tensor *wi, const nvdr_tensor *d_out, float *nrm_grad,
                                float *wi_grad, void *stream)
{
    static const char *OP = "lambert_bwd";
    NVDR_REQUIRE(nrm && wi && d_out && nrm_grad && wi_grad, "%s: NULL argument", OP);
    const Extent e = make_extent(nrm, wi);
    CHECK_VIEW(nrm, 3); CHECK_VIEW(wi, 3); CHECK_VIEW(d_out, 1);
    const View4 a = make_view4(*nrm), b = make_view4(*wi), g = make_view4(*d_out);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        F3 dn = f3(0.0f), dw = f3(0.0f);
        bwd_lambert(fetch3(a, n, h, w), fetch3(b, n, h, w), dn, dw, fetch1(g, n, h, w));
        store3(nrm_grad, i, dn);
        store3(wi_grad, i, dw);
    });
}

extern "C" int nvdr_frostbite_fwd(const nvdr_tensor *nrm, const nvdr_tensor *wi, const nvdr_tensor *wo,
                                  const nvdr_tensor *linear_roughness, float *out, void *stream)
{
    static const char *OP = "frostbite_fwd";
    NVDR_REQUIRE(nrm && wi && wo && linear_roughness && out, "%s: NULL argument", OP);
    const Extent e = make_extent(nrm, wi, wo, linear_roughness);
    CHECK_VIEW(nrm, 3); CHECK_VIEW(wi, 3); CHECK_VIEW(wo, 3); CHECK_VIEW(linear_roughness, 1);
    const View4 a = make_view4(*nrm), b = make_view4(*wi), c = make_view4(*wo), r = make_view4(*linear_roughness);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        out[i] = fwd_frostbite(fetch3(a, n, h, w), fetch3(b, n, h, w), fetch3(c, n, h, w), fetch1(r, n, h, w));
    });
}
extern "C" int nvdr_frostbite_bwd(const nvdr_tensor *nrm, const nvdr_tensor *wi, const nvdr_tensor *wo,
                                  const nvdr_tensor *linear_roughness, const nvdr_tensor *d_out, float *nrm_grad,
                                  float *wi_grad, float *wo_grad, float *linear_roughness_grad, void *stream)
{
    static const char *OP = "frostbite_bwd";
    NVDR_REQUIRE(nrm && wi && wo && linear_roughness && d_out && nrm_grad && wi_grad && wo_grad && linear_roughness_grad,
                 "%s: NULL argument", OP);
    const Extent e = make_extent(nrm, wi, wo, linear_roughness);
    CHECK_VIEW(nrm, 3); CHECK_VIEW(wi, 3); CHECK_VIEW(wo, 3); CHECK_VIEW(linear_roughness, 1); CHECK_VIEW(d_out, 1);
    const View4 a = make_view4(*nrm), b = make_view4(*wi), c = make_view4(*wo), r = make_view4(*linear_roughness),
                g = make_view4(*d_out);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        F3 dn = f3(0.0f), dwi = f3(0.0f), dwo = f3(0.0f);
        float dr = 0.0f;
        bwd_frostbite(fetch3(a, n, h, w), fetch3(b, n, h, w), fetch3(c, n, h, w), fetch1(r, n, h, w), dn, dwi, dwo, dr,
                      fetch1(g, n, h, w));
        store3(nrm_grad, i, dn);
        store3(wi_grad, i, dwi);
        store3(wo_grad, i, dwo);
        linear_roughness_grad[i] = dr;
    });
}

extern "C" int nvdr_fresnel_shlick_fwd(const nvdr_tensor *f0, const nvdr_tensor *f90, const nvdr_tensor *cos_theta,
                                       float *out, void *stream)
{
    static const char *OP = "fresnel_shlick_fwd";
    NVDR_REQUIRE(f0 && f90 && cos_theta && out, "%s: NULL argument", OP);
    const Extent e = make_extent(f0, f90, cos_theta);
    CHECK_VIEW(f0, 3); CHECK_VIEW(f90, 3); CHECK_VIEW(cos_theta, 1);
    const View4 a = make_view4(*f0), b = make_view4(*f90), c = make_view4(*cos_theta);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        store3(out, i, fwd_fresnel3(fetch3(a, n, h, w), fetch3(b, n, h, w), fetch1(c, n, h, w)));
    });
}
extern "C" int nvdr_fresnel_shlick_bwd(const nvdr_tensor *f0, const nvdr_tensor *f90, const nvdr_tensor *cos_theta,
                                       const nvdr_tensor *d_out, float *f0_grad, float *f90_grad, float *cos_theta_grad,
                                       void *stream)
{
    static const char *OP = "fresnel_shlick_bwd";
    NVDR_REQUIRE(f0 && f90 && cos_theta && d_out && f0_grad && f90_grad && cos_theta_grad, "%s: NULL argument", OP);
    const Extent e = make_extent(f0, f90, cos_theta);
    CHECK_VIEW(f0, 3); CHECK_VIEW(f90, 3); CHECK_VIEW(cos_theta, 1); CHECK_VIEW(d_out, 3);
    const View4 a = make_view4(*f0), b = make_view4(*f90), c = make_view4(*cos_theta), g = make_view4(*d_out);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        F3 d0 = f3(0.0f), d90 = f3(0.0f);
        float dc = 0.0f;
        bwd_fresnel3(fetch3(a, n, h, w), fetch3(b, n, h, w), fetch1(c, n, h, w), d0, d90, dc, fetch3(g, n, h, w));
        store3(f0_grad, i, d0);
        store3(f90_grad, i, d90);
        cos_theta_grad[i] = dc;
    });
}

#define SCALAR2_OP(NAME, FWD, BWD)                                                                                  \
    extern "C" int nvdr_##NAME##_fwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta, float *out, void *stream) \
    {                                                                                                               \
        static const char *OP = #NAME "_fwd";                                                                       \
        NVDR_REQUIRE(alpha_sqr && cos_theta && out, "%s: NULL argument", OP);                                       \
        const Extent e = make_extent(alpha_sqr, cos_theta);                                                         \
        CHECK_VIEW(alpha_sqr, 1); CHECK_VIEW(cos_theta, 1);                                                         \
        const View4 a = make_view4(*alpha_sqr), c = make_view4(*cos_theta);                                         \
        return launch_ew(e, (hipStream_t)stream,                                                                    \
                         [=] __device__(int n, int h, int w, int64_t i) { out[i] = FWD(fetch1(a, n, h, w), fetch1(c, n, h, w)); }); \
    }                                                                                                               \
    extern "C" int nvdr_##NAME##_bwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta, const nvdr_tensor *d_out, \
                                     float *alpha_sqr_grad, float *cos_theta_grad, void *stream)                   \
    {                                                                                                               \
        static const char *OP = #NAME "_bwd";                                                                       \
        NVDR_REQUIRE(alpha_sqr && cos_theta && d_out && alpha_sqr_grad && cos_theta_grad, "%s: NULL argument", OP); \
        const Extent e = make_extent(alpha_sqr, cos_theta);                                                         \
        CHECK_VIEW(alpha_sqr, 1); CHECK_VIEW(cos_theta, 1); CHECK_VIEW(d_out, 1);                                   \
        const View4 a = make_view4(*alpha_sqr), c = make_view4(*cos_theta), g = make_view4(*d_out);                 \
        return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {                   \
            float da = 0.0f, dc = 0.0f;                                                                             \
            BWD(fetch1(a, n, h, w), fetch1(c, n, h, w), da, dc, fetch1(g, n, h, w));                                \
            alpha_sqr_grad[i] = da;                                                                                 \
            cos_theta_grad[i] = dc;                                                                                 \
        });                                                                                                         \
    }
SCALAR2_OP(ndf_ggx, fwd_ndf_ggx, bwd_ndf_ggx)
SCALAR2_OP(lambda_ggx, fwd_lambda_ggx, bwd_lambda_ggx)

extern "C" int nvdr_masking_smith_fwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta_i,
                                      const nvdr_tensor *cos_theta_o, float *out, void *stream)
{
    static const char *OP = "masking_smith_fwd";
    NVDR_REQUIRE(alpha_sqr && cos_theta_i && cos_theta_o && out, "%s: NULL argument", OP);
    const Extent e = make_extent(alpha_sqr, cos_theta_i, cos_theta_o);
    CHECK_VIEW(alpha_sqr, 1); CHECK_VIEW(cos_theta_i, 1); CHECK_VIEW(cos_theta_o, 1);
    const View4 a = make_view4(*alpha_sqr), ci = make_view4(*cos_theta_i), co = make_view4(*cos_theta_o);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        out[i] = fwd_masking_smith(fetch1(a, n, h, w), fetch1(ci, n, h, w), fetch1(co, n, h, w));
    });
}
extern "C" int nvdr_masking_smith_bwd(const nvdr_tensor *alpha_sqr, const nvdr_tensor *cos_theta_i,
                                      const nvdr_tensor *cos_theta_o, const nvdr_tensor *d_out, float *alpha_sqr_grad,
                                      float *cos_theta_i_grad, float *cos_theta_o_grad, void *stream)
{
    static const char *OP = "masking_smith_bwd";
    NVDR_REQUIRE(alpha_sqr && cos_theta_i && cos_theta_o && d_out && alpha_sqr_grad && cos_theta_i_grad && cos_theta_o_grad,
                 "%s: NULL argument", OP);
    const Extent e = make_extent(alpha_sqr, cos_theta_i, cos_theta_o);
    CHECK_VIEW(alpha_sqr, 1); CHECK_VIEW(cos_theta_i, 1); CHECK_VIEW(cos_theta_o, 1); CHECK_VIEW(d_out, 1);
    const View4 a = make_view4(*alpha_sqr), ci = make_view4(*cos_theta_i), co = make_view4(*cos_theta_o), g = make_view4(*d_out);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        float da = 0.0f, di = 0.0f, d_o = 0.0f;
        bwd_masking_smith(fetch1(a, n, h, w), fetch1(ci, n, h, w), fetch1(co, n, h, w), da, di, d_o, fetch1(g, n, h, w));
        alpha_sqr_grad[i] = da;
        cos_theta_i_grad[i] = di;
        cos_theta_o_grad[i] = d_o;
    });
}

extern "C" int nvdr_pbr_specular_fwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *wo,
                                     const nvdr_tensor *wi, const nvdr_tensor *alpha, float min_roughness, float *out,
                                     void *stream)
{
    static const char *OP = "pbr_specular_fwd";
    NVDR_REQUIRE(col && nrm && wo && wi && alpha && out, "%s: NULL argument", OP);
    const Extent e = make_extent(col, nrm, wo, wi, alpha);
    CHECK_VIEW(col, 3); CHECK_VIEW(nrm, 3); CHECK_VIEW(wo, 3); CHECK_VIEW(wi, 3); CHECK_VIEW(alpha, 1);
    const View4 a = make_view4(*col), b = make_view4(*nrm), c = make_view4(*wo), d = make_view4(*wi), al = make_view4(*alpha);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        store3(out, i, fwd_pbr_specular(fetch3(a, n, h, w), fetch3(b, n, h, w), fetch3(c, n, h, w), fetch3(d, n, h, w),
                                        fetch1(al, n, h, w), min_roughness));
    });
}
extern "C" int nvdr_pbr_specular_bwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *wo,
                                     const nvdr_tensor *wi, const nvdr_tensor *alpha, float min_roughness,
                                     const nvdr_tensor *d_out, float *col_grad, float *nrm_grad, float *wo_grad,
                                     float *wi_grad, float *alpha_grad, void *stream)
{
    static const char *OP = "pbr_specular_bwd";
    NVDR_REQUIRE(col && nrm && wo && wi && alpha && d_out && col_grad && nrm_grad && wo_grad && wi_grad && alpha_grad,
                 "%s: NULL argument", OP);
    const Extent e = make_extent(col, nrm, wo, wi, alpha);
    CHECK_VIEW(col, 3); CHECK_VIEW(nrm, 3); CHECK_VIEW(wo, 3); CHECK_VIEW(wi, 3); CHECK_VIEW(alpha, 1); CHECK_VIEW(d_out, 3);
    const View4 a = make_view4(*col), b = make_view4(*nrm), c = make_view4(*wo), d = make_view4(*wi), al = make_view4(*alpha),
                g = make_view4(*d_out);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        F3 dc = f3(0.0f), dn = f3(0.0f), dwo = f3(0.0f), dwi = f3(0.0f);
        float da = 0.0f;
        bwd_pbr_specular(fetch3(a, n, h, w), fetch3(b, n, h, w), fetch3(c, n, h, w), fetch3(d, n, h, w), fetch1(al, n, h, w),
                         min_roughness, dc, dn, dwo, dwi, da, fetch3(g, n, h, w));
        store3(col_grad, i, dc);
        store3(nrm_grad, i, dn);
        store3(wo_grad, i, dwo);
        store3(wi_grad, i, dwi);
        alpha_grad[i] = da;
    });
}

extern "C" int nvdr_pbr_bsdf_fwd(const nvdr_tensor *kd, const nvdr_tensor *arm, const nvdr_tensor *pos,
                                 const nvdr_tensor *nrm, const nvdr_tensor *view_pos, const nvdr_tensor *light_pos,
                                 float min_roughness, int bsdf, float *out, void *stream)
{
    static const char *OP = "pbr_bsdf_fwd";
    NVDR_REQUIRE(kd && arm && pos && nrm && view_pos && light_pos && out, "%s: NULL argument", OP);
    const Extent e = make_extent(kd, arm, pos, nrm, view_pos, light_pos);
    CHECK_VIEW(kd, 3); CHECK_VIEW(arm, 3); CHECK_VIEW(pos, 3); CHECK_VIEW(nrm, 3); CHECK_VIEW(view_pos, 3); CHECK_VIEW(light_pos, 3);
    const View4 a = make_view4(*kd), b = make_view4(*arm), c = make_view4(*pos), d = make_view4(*nrm),
                v = make_view4(*view_pos), l = make_view4(*light_pos);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        store3(out, i, fwd_pbr_bsdf_ru(fetch3(a, n, h, w), fetch3(b, n, h, w), fetch3(c, n, h, w), fetch3(d, n, h, w),
                                       fetch3(v, n, h, w), fetch3(l, n, h, w), min_roughness, bsdf));
    });
}
extern "C" int nvdr_pbr_bsdf_bwd(const nvdr_tensor *kd, const nvdr_tensor *arm, const nvdr_tensor *pos,
                                 const nvdr_tensor *nrm, const nvdr_tensor *view_pos, const nvdr_tensor *light_pos,
                                 float min_roughness, int bsdf, const nvdr_tensor *d_out, float *kd_grad, float *arm_grad,
                                 float *pos_grad, float *nrm_grad, float *view_pos_grad, float *light_pos_grad, void *stream)
{
    static const char *OP = "pbr_bsdf_bwd";
    NVDR_REQUIRE(kd && arm && pos && nrm && view_pos && light_pos && d_out, "%s: NULL argument", OP);
    NVDR_REQUIRE(kd_grad && arm_grad && pos_grad && nrm_grad && view_pos_grad && light_pos_grad, "%s: NULL output", OP);
    const Extent e = make_extent(kd, arm, pos, nrm, view_pos, light_pos);
    CHECK_VIEW(kd, 3); CHECK_VIEW(arm, 3); CHECK_VIEW(pos, 3); CHECK_VIEW(nrm, 3); CHECK_VIEW(view_pos, 3);
    CHECK_VIEW(light_pos, 3); CHECK_VIEW(d_out, 3);
    const View4 a = make_view4(*kd), b = make_view4(*arm), c = make_view4(*pos), d = make_view4(*nrm),
                v = make_view4(*view_pos), l = make_view4(*light_pos), g = make_view4(*d_out);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        F3 dkd = f3(0.0f), darm = f3(0.0f), dpos = f3(0.0f), dnrm = f3(0.0f), dvp = f3(0.0f), dlp = f3(0.0f);
        bwd_pbr_bsdf_ru(fetch3(a, n, h, w), fetch3(b, n, h, w), fetch3(c, n, h, w), fetch3(d, n, h, w), fetch3(v, n, h, w),
                        fetch3(l, n, h, w), min_roughness, bsdf, dkd, darm, dpos, dnrm, dvp, dlp, fetch3(g, n, h, w));
        store3(kd_grad, i, dkd);
        store3(arm_grad, i, darm);
        store3(pos_grad, i, dpos);
        store3(nrm_grad, i, dnrm);
        store3(view_pos_grad, i, dvp);
        store3(light_pos_grad, i, dlp);
    });
}

// ---- shade composite (render/render.py:119-127 + the division of optixutils/ops.py:139-141) ------------------------
// The reference composes the final colour with ~10 small torch kernels per direction:
//     diffuse  = diff_w.rgb / diff_w.w          (bilateral_denoiser's Python epilogue; skipped for 3-channel inputs)
//     specular = spec_w.rgb / spec_w.w
//     shaded   = diffuse * kd * (1 - ks.z) + specular         ('pbr')       or   diffuse * kd   ('diffuse', 'white')
// Fused here into one streaming kernel per direction.  Additive API: nothing in the reference's plugin.
__device__ __forceinline__ F3 fetch_rgb_over_w(const View4 &v, int n, int h, int w, float &inv_w)
{
    const float *q = v.p + n * v.s0 + h * v.s1 + w * v.s2;
    const float wt = v.c == 4 ? q[3 * v.s3] : 1.0f;
    inv_w = 1.0f / wt;
    return f3(q[0] / wt, q[v.s3] / wt, q[2 * v.s3] / wt);   // true division, as torch does
}
static int check_accum(const nvdr_tensor *t, const Extent &e, const char *op, const char *name)
{
    NVDR_REQUIRE(t && t->data, "%s: %s is NULL", op, name);
    NVDR_REQUIRE(t->size[0] == e.N && t->size[1] == e.H && t->size[2] == e.W && (t->size[3] == 3 || t->size[3] == 4),
                 "%s: %s must be [%d,%d,%d,3] or [%d,%d,%d,4] (colour sum and weight), got [%lld,%lld,%lld,%lld]", op, name,
                 e.N, e.H, e.W, e.N, e.H, e.W, (long long)t->size[0], (long long)t->size[1], (long long)t->size[2],
                 (long long)t->size[3]);
    return 0;
}
extern "C" int nvdr_shade_composite_fwd(const nvdr_tensor *diff, const nvdr_tensor *spec, const nvdr_tensor *kd,
                                        const nvdr_tensor *ks, int bsdf, float *out, void *stream)
{
    static const char *OP = "shade_composite_fwd";
    NVDR_REQUIRE(diff && spec && kd && ks && out, "%s: NULL argument", OP);
    NVDR_REQUIRE(bsdf == 0 || bsdf == 1, "%s: bsdf must be 0 (pbr) or 1 (diffuse only)", OP);
    const Extent e = make_extent(diff, spec, kd, ks);
    int r;
    if ((r = check_accum(diff, e, OP, "diff"))) return r;
    if ((r = check_accum(spec, e, OP, "spec"))) return r;
    CHECK_VIEW(kd, 3); CHECK_VIEW(ks, 3);
    const View4 a = make_view4(*diff), b = make_view4(*spec), c = make_view4(*kd), d = make_view4(*ks);
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        float iwd, iws;
        const F3 dn = fetch_rgb_over_w(a, n, h, w, iwd), sn = fetch_rgb_over_w(b, n, h, w, iws);
        const F3 k = fetch3(c, n, h, w), arm = fetch3(d, n, h, w);
        store3(out, i, bsdf == 0 ? dn * (k * (1.0f - arm.z)) + sn : dn * k);
    });
}
extern "C" int nvdr_shade_composite_bwd(const nvdr_tensor *diff, const nvdr_tensor *spec, const nvdr_tensor *kd,
                                        const nvdr_tensor *ks, int bsdf, const nvdr_tensor *d_out, float *diff_grad,
                                        float *spec_grad, float *kd_grad, float *ks_grad, void *stream)
{
    static const char *OP = "shade_composite_bwd";
    NVDR_REQUIRE(diff && spec && kd && ks && d_out && diff_grad && spec_grad && kd_grad && ks_grad, "%s: NULL argument", OP);
    NVDR_REQUIRE(bsdf == 0 || bsdf == 1, "%s: bsdf must be 0 (pbr) or 1 (diffuse only)", OP);
    const Extent e = make_extent(diff, spec, kd, ks);
    int r;
    if ((r = check_accum(diff, e, OP, "diff"))) return r;
    if ((r = check_accum(spec, e, OP, "spec"))) return r;
    CHECK_VIEW(kd, 3); CHECK_VIEW(ks, 3); CHECK_VIEW(d_out, 3);
    const View4 a = make_view4(*diff), b = make_view4(*spec), c = make_view4(*kd), d = make_view4(*ks), g = make_view4(*d_out);
    const int cd = (int)diff->size[3], cs = (int)spec->size[3];
    return launch_ew(e, (hipStream_t)stream, [=] __device__(int n, int h, int w, int64_t i) {
        float iwd, iws;
        const F3 dn = fetch_rgb_over_w(a, n, h, w, iwd), sn = fetch_rgb_over_w(b, n, h, w, iws);
        const F3 k = fetch3(c, n, h, w), arm = fetch3(d, n, h, w), go = fetch3(g, n, h, w);
        const float om = bsdf == 0 ? 1.0f - arm.z : 1.0f;
        const F3 d_dn = go * (k * om);                       // gradient of the normalised diffuse term
        const F3 d_sn = bsdf == 0 ? go : f3(0.0f);
        float *pd = diff_grad + i * cd, *ps = spec_grad + i * cs;
        pd[0] = d_dn.x * iwd; pd[1] = d_dn.y * iwd; pd[2] = d_dn.z * iwd;
        if (cd == 4) pd[3] = -sum3(d_dn * dn) * iwd;         // d(rgb / w) / dw = -rgb / w^2
        ps[0] = d_sn.x * iws; ps[1] = d_sn.y * iws; ps[2] = d_sn.z * iws;
        if (cs == 4) ps[3] = -sum3(d_sn * sn) * iws;
        store3(kd_grad, i, go * dn * om);
        store3(ks_grad, i, f3(0.0f, 0.0f, bsdf == 0 ? -sum3(go * dn * k) : 0.0f));
    });
}

// ---- composite + mean image loss, forward AND backward, in one launch (additive, round 6).  The tail of shade() and the loss of a training
// iteration are five small launches per iteration -- composite, loss partials, their reduction, loss adjoint, composite adjoint: 40 us of
// kernels and as many dependency gaps in a 1.8 ms one-view iteration -- and the adjoint of a MEAN does not need the mean: d loss / d pixel is
// d_mean / (N H W) times a per-pixel factor.  One pass computes the composite, the pixel's loss term (block partial sums, reduced by the
// launch behind it) and, with the upstream gradient of the mean read from a device scalar, the gradients of the four composite inputs.
// The arithmetic is the separate kernels' statement for statement (same values bit for bit: tests/test_gpu_renderutils.py).
__global__ void __launch_bounds__(LOSS_BLOCK) shade_loss_fused_kernel(Extent e, View4 a, View4 b, View4 c, View4 d, View4 tg, int bsdf, int loss, int tonemapper,
                                                                       int cd, int cs, const float *__restrict__ d_mean, float mean_scale,
                                                                       float *__restrict__ partials, float *__restrict__ diff_grad,
                                                                       float *__restrict__ spec_grad, float *__restrict__ kd_grad, float *__restrict__ ks_grad)
{
    __shared__ float wsum[LOSS_BLOCK / 64];
    const float d_out = d_mean[0] * mean_scale;
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e.total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % e.W), h = (int)((i / e.W) % e.H), n = (int)(i / ((int64_t)e.W * e.H));
        float iwd, iws;
        const F3 dn = fetch_rgb_over_w(a, n, h, w, iwd), sn = fetch_rgb_over_w(b, n, h, w, iws);
        const F3 k = fetch3(c, n, h, w), arm = fetch3(d, n, h, w);
        const F3 img = bsdf == 0 ? dn * (k * (1.0f - arm.z)) + sn : dn * k;                  // nvdr_shade_composite_fwd
        const F3 tgt = fetch3(tg, n, h, w);
        const float av[3] = {img.x, img.y, img.z}, bv[3] = {tgt.x, tgt.y, tgt.z};
        float s = 0.0f, gi[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float x = clampf(av[ch], 0.0f, 65535.0f), t = clampf(bv[ch], 0.0f, 65535.0f);  // image_loss_fwd_kernel
            if (tonemapper) {
                x = tonemap_fwd(x);
                t = tonemap_fwd(t);
            }
            s += loss_fwd1(loss, x, t);
            float xu = av[ch], tu = bv[ch];                                                 // image_loss_bwd_impl (tonemaps the UNclamped value)
            if (tonemapper) {
                xu = tonemap_fwd(xu);
                tu = tonemap_fwd(tu);
            }
            float dx, dt;
            loss_bwd1(loss, xu, tu, d_out / 3.0f, dx, dt);
            if (tonemapper) dx = tonemap_bwd(av[ch], dx);
            if (av[ch] <= 0.0f || av[ch] >= 65535.0f) dx = 0;
            gi[ch] = dx;
        }
        acc += s / 3.0f;
        const F3 go = f3(gi[0], gi[1], gi[2]);                                              // nvdr_shade_composite_bwd
        const float om = bsdf == 0 ? 1.0f - arm.z : 1.0f;
        const F3 d_dn = go * (k * om);
        const F3 d_sn = bsdf == 0 ? go : f3(0.0f);
        float *pd = diff_grad + i * cd, *ps = spec_grad + i * cs;
        pd[0] = d_dn.x * iwd; pd[1] = d_dn.y * iwd; pd[2] = d_dn.z * iwd;
        if (cd == 4) pd[3] = -sum3(d_dn * dn) * iwd;
        ps[0] = d_sn.x * iws; ps[1] = d_sn.y * iws; ps[2] = d_sn.z * iws;
        if (cs == 4) ps[3] = -sum3(d_sn * sn) * iws;
        store3(kd_grad, i, go * dn * om);
        store3(ks_grad, i, f3(0.0f, 0.0f, bsdf == 0 ? -sum3(go * dn * k) : 0.0f));
    }
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int q = 0; q < LOSS_BLOCK / 64; ++q) t += wsum[q];
        partials[blockIdx.x] = t;
    }
}

extern "C" int nvdr_shade_loss_fused(const nvdr_tensor *diff, const nvdr_tensor *spec, const nvdr_tensor *kd, const nvdr_tensor *ks, int bsdf,
                                     const nvdr_tensor *target, int loss, int tonemapper, const float *d_mean, float *partials, float *out_mean,
                                     float *diff_grad, float *spec_grad, float *kd_grad, float *ks_grad, void *stream)
{
    static const char *OP = "shade_loss_fused";
    NVDR_REQUIRE(diff && spec && kd && ks && target && d_mean && partials && out_mean && diff_grad && spec_grad && kd_grad && ks_grad, "%s: NULL argument", OP);
    NVDR_REQUIRE(bsdf == 0 || bsdf == 1, "%s: bsdf must be 0 (pbr) or 1 (diffuse only)", OP);
    NVDR_REQUIRE(loss >= 0 && loss <= 4 && (tonemapper == 0 || tonemapper == 1), "%s: bad loss/tonemapper", OP);
    const Extent e = make_extent(diff, spec, kd, ks, target);
    int r;
    if ((r = check_accum(diff, e, OP, "diff"))) return r;
    if ((r = check_accum(spec, e, OP, "spec"))) return r;
    CHECK_VIEW(kd, 3); CHECK_VIEW(ks, 3); CHECK_VIEW(target, 3);
    NVDR_REQUIRE(kd->size[0] == e.N && kd->size[1] == e.H && kd->size[2] == e.W && ks->size[0] == e.N && ks->size[1] == e.H && ks->size[2] == e.W,
                 "%s: kd / ks must have the full extent (their gradients are written at it)", OP);
    const int64_t n = nvdr_image_loss_num_partials(e.N, e.H, e.W);
    shade_loss_fused_kernel<<<(unsigned)n, LOSS_BLOCK, 0, (hipStream_t)stream>>>(e, make_view4(*diff), make_view4(*spec), make_view4(*kd), make_view4(*ks),
                                                                                   make_view4(*target), bsdf, loss, tonemapper, (int)diff->size[3], (int)spec->size[3],
                                                                                   d_mean, 1.0f / (float)((int64_t)e.N * e.H * e.W), partials, diff_grad, spec_grad,
                                                                                   kd_grad, ks_grad);
    image_loss_reduce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(partials, n, 1.0f / (float)((int64_t)e.N * e.H * e.W), out_mean);
    NVDR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Rows of a table by index (additive): out[i] = index[i] >= 0 ? table[index[i]] : 0 -- the nearest-texel lookup of the trained kd
// texture in the iteration harness (trainer.py), which torch composes from zeros + index_select + index_copy (forward) and
// zeros + index_select + index_add (backward).

__global__ void __launch_bounds__(256) gather_rows_fwd_kernel(const float *__restrict__ table, const int *__restrict__ index, int64_t n, int c,
                                                              float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    const int64_t r = i / c;
    const int k = (int)(i - r * c), j = index[r];
    out[i] = j >= 0 ? table[(int64_t)j * c + k] : 0.0f;
}

__global__ void __launch_bounds__(256) gather_rows_bwd_kernel(const float *__restrict__ dout, const int *__restrict__ index, int64_t n, int c,
                                                              float *__restrict__ dtable)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    const int64_t r = i / c;
    const int k = (int)(i - r * c), j = index[r];
    if (j >= 0) atomicAdd(dtable + (int64_t)j * c + k, dout[i]);
}

extern "C" int nvdr_gather_rows_fwd(const float *table, const int *index, int64_t n, int channels, float *out, void *stream)
{
    NVDR_REQUIRE(table && index && out && n >= 0 && channels >= 1 && channels <= 16, "gather_rows_fwd: bad argument");
    if (n == 0) return 0;
    gather_rows_fwd_kernel<<<div_up(n * channels, 256), 256, 0, (hipStream_t)stream>>>(table, index, n, channels, out);
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_gather_rows_bwd(const float *dout, const int *index, int64_t n, int channels, int64_t table_rows, float *dtable, void *stream)
{
    NVDR_REQUIRE(dout && index && dtable && n >= 0 && channels >= 1 && channels <= 16 && table_rows >= 0, "gather_rows_bwd: bad argument");
    NVDR_HIP_TRY(hipMemsetAsync(dtable, 0, sizeof(float) * (size_t)table_rows * channels, (hipStream_t)stream));
    if (n == 0) return 0;
    gather_rows_bwd_kernel<<<div_up(n * channels, 256), 256, 0, (hipStream_t)stream>>>(dout, index, n, channels, dtable);
    NVDR_LAUNCH_CHECK();
    return 0;
}
