// ref_driver.cpp -- runs the REFERENCE's own env-shade raygen program and bilateral-denoiser kernels
// on the CPU.  The reference sources are #included from where they lie (NVDR_REF_ROOT, set by the
// Makefile); nothing from them is copied into this repository.  Output: oracle/_ref/libnvdr_ref.so,
// used by tests/test_oracle_pins.py to pin oracle/nvdr_oracle.c, and optionally by bench.py as the
// "reference" CPU baseline.
#include "cuda_shim.h"
#include "optix.h"

thread_local uint3 ref_launch_index;
uint3 ref_launch_dim;
thread_local uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
thread_local unsigned int *ref_payload0;

extern "C" void __miss__ms();

#include "nvdr_hip.h"
#include "nvdr_raytri.h"

static const float *g_trirec = nullptr; // [T,9] v0,e1,e2
static long g_ntris = 0;
static thread_local const unsigned char *g_vis_in = nullptr; // optional injected visibility for the current pixel
static thread_local unsigned char *g_vis_out = nullptr;
static thread_local int g_ray_counter = 0;

void optixTrace(OptixTraversableHandle, float3 o, float3 d, float, float, float, OptixVisibilityMask, unsigned int,
                unsigned int, unsigned int, unsigned int, unsigned int &p0)
{
    bool hit = false;
    if (g_vis_in) {
        hit = g_vis_in[g_ray_counter] == 0;
    } else {
        const float *t = g_trirec;
        for (long k = 0; k < g_ntris; ++k, t += 9) {
            float tn, un, vn, det;
            if (nvdr_ray_tri(o.x, o.y, o.z, d.x, d.y, d.z, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], &tn, &un, &vn, &det)) {
                hit = true;
                break;
            }
        }
    }
    if (g_vis_out) g_vis_out[g_ray_counter] = hit ? 0 : 1;
    g_ray_counter++;
    ref_payload0 = &p0;
    if (!hit) __miss__ms();
}

// ---- the reference device code, compiled in place -------------------------------------------
#include "envsampling/kernel.cu"
#include "denoising.cu"

// ---- glue: fill the reference's accessor structs from nvdr_tensor views ----------------------
template <int N> struct RawAcc { void *p; int32_t sizes[N]; int32_t strides[N]; };

template <class T, int N> static void fill(PackedTensorAccessor32<T, N> &acc, const nvdr_tensor &t, int first_dim)
{
    static_assert(sizeof(PackedTensorAccessor32<T, N>) == sizeof(RawAcc<N>), "accessor layout");
    RawAcc<N> r;
    r.p = t.data;
    for (int i = 0; i < N; ++i) {
        r.sizes[i] = (int32_t)t.size[first_dim + i];
        r.strides[i] = (int32_t)t.stride[first_dim + i];
    }
    memcpy((void *)&acc, &r, sizeof(r));
}
template <class T, int N> static void fill_contig(PackedTensorAccessor32<T, N> &acc, T *p, const long *sizes)
{
    RawAcc<N> r;
    r.p = p;
    long s = 1;
    for (int i = N - 1; i >= 0; --i) {
        r.sizes[i] = (int32_t)sizes[i];
        r.strides[i] = (int32_t)s;
        s *= sizes[i];
    }
    memcpy((void *)&acc, &r, sizeof(r));
}

extern "C" void oracle_make_trirec_ref(const float *verts, const int32_t *tris, long n_tris, float *out9)
{
    for (long k = 0; k < n_tris; ++k) {
        const float *a = verts + 3 * tris[3 * k], *b = verts + 3 * tris[3 * k + 1], *c = verts + 3 * tris[3 * k + 2];
        float *o = out9 + 9 * k;
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
        o[3] = b[0] - a[0]; o[4] = b[1] - a[1]; o[5] = b[2] - a[2];
        o[6] = c[0] - a[0]; o[7] = c[1] - a[1]; o[8] = c[2] - a[2];
    }
}

// Same calling convention as oracle_env_shade() in nvdr_oracle.c (host pointers; mask is [N,H,W] in
// dims 0..2 of the view, 4-d tensors in dims 0..3, light in dims 0..2, pdf/cols in 0..1, rows in 0).
extern "C" long ref_env_shade(const nvdr_env_shade_args *a, const float *verts, const int32_t *tris, long n_tris,
                              int backward, int n_threads, const unsigned char *vis_in, unsigned char *vis_out)
{
    const long N = a->ro.size[0], H = a->ro.size[1], W = a->ro.size[2];
    const long npix = N * H * W;
    const unsigned S = a->n_samples_x * a->n_samples_x;
    float *rec = nullptr;
    if (!vis_in) {
        rec = (float *)malloc(sizeof(float) * 9 * (size_t)n_tris);
        oracle_make_trirec_ref(verts, tris, n_tris, rec);
    }
    g_trirec = rec;
    g_ntris = n_tris;

    EnvSamplingParams &p = params;
    memset((void *)&p, 0, sizeof(p));
    fill<float, 3>(p.mask, a->mask, 0);
    fill<float, 4>(p.ro, a->ro, 0);
    fill<float, 4>(p.gb_pos, a->gb_pos, 0);
    fill<float, 4>(p.gb_normal, a->gb_normal, 0);
    fill<float, 4>(p.gb_view_pos, a->gb_view_pos, 0);
    fill<float, 4>(p.gb_kd, a->gb_kd, 0);
    fill<float, 4>(p.gb_ks, a->gb_ks, 0);
    fill<float, 3>(p.light, a->light, 0);
    fill<float, 2>(p.pdf, a->pdf, 0);
    fill<float, 1>(p.rows, a->rows, 0);
    fill<float, 2>(p.cols, a->cols, 0);
    fill<int, 2>(p.perms, a->perms, 0);
    const long s4[4] = {N, H, W, 3};
    const long sl[3] = {a->light.size[0], a->light.size[1], 3};
    if (!backward) {
        memset(a->diff, 0, sizeof(float) * 3 * npix);
        memset(a->spec, 0, sizeof(float) * 3 * npix);
        fill_contig<float, 4>(p.diff, a->diff, s4);
        fill_contig<float, 4>(p.spec, a->spec, s4);
    } else {
        fill<float, 4>(p.diff_grad, a->diff_grad, 0);
        fill<float, 4>(p.spec_grad, a->spec_grad, 0);
        memset(a->gb_pos_grad, 0, sizeof(float) * 3 * npix);
        memset(a->gb_normal_grad, 0, sizeof(float) * 3 * npix);
        memset(a->gb_kd_grad, 0, sizeof(float) * 3 * npix);
        memset(a->gb_ks_grad, 0, sizeof(float) * 3 * npix);
        memset(a->light_grad, 0, sizeof(float) * sl[0] * sl[1] * 3);
        fill_contig<float, 4>(p.gb_pos_grad, a->gb_pos_grad, s4);
        fill_contig<float, 4>(p.gb_normal_grad, a->gb_normal_grad, s4);
        fill_contig<float, 4>(p.gb_kd_grad, a->gb_kd_grad, s4);
        fill_contig<float, 4>(p.gb_ks_grad, a->gb_ks_grad, s4);
        fill_contig<float, 3>(p.light_grad, a->light_grad, sl);
    }
    p.handle = 0;
    p.BSDF = a->bsdf;
    p.n_samples_x = a->n_samples_x;
    p.rnd_seed = a->rnd_seed;
    p.backward = backward ? 1 : 0;
    p.shadow_scale = a->shadow_scale;
    ref_launch_dim = make_uint3((unsigned)W, (unsigned)H, (unsigned)N);
    long covered = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : covered) num_threads(n_threads > 0 ? n_threads : 1)
    for (long lin = 0; lin < npix; ++lin) {
        ref_launch_index = make_uint3((unsigned)(lin % W), (unsigned)((lin / W) % H), (unsigned)(lin / (H * W)));
        g_ray_counter = 0;
        g_vis_in = vis_in ? vis_in + lin * 2 * S : nullptr;
        g_vis_out = vis_out ? vis_out + lin * 2 * S : nullptr;
        __raygen__rg();
        covered += g_ray_counter > 0;
    }
    free(rec);
    g_trirec = nullptr;
    return covered;
}

static void run_grid(void (*kernel)(BilateralDenoiserParams), BilateralDenoiserParams &bp, long N, long H, long W, int n_threads)
{
    blockDim.x = 8; blockDim.y = 8; blockDim.z = 1; // torch_bindings.cpp:281
    const long gx = (W - 1) / 8 + 1, gy = (H - 1) / 8 + 1;
#pragma omp parallel for collapse(2) schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
    for (long bz = 0; bz < N; ++bz)
        for (long by = 0; by < gy; ++by)
            for (long bx = 0; bx < gx; ++bx)
                for (unsigned ty = 0; ty < 8; ++ty)
                    for (unsigned tx = 0; tx < 8; ++tx) {
                        blockIdx = make_uint3((unsigned)bx, (unsigned)by, (unsigned)bz);
                        threadIdx = make_uint3(tx, ty, 0);
                        kernel(bp);
                    }
}

extern "C" void ref_bilateral_fwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz, float sigma,
                                  float *out, int n_threads)
{
    BilateralDenoiserParams bp;
    memset((void *)&bp, 0, sizeof(bp));
    fill<float, 4>(bp.col, *col, 0);
    fill<float, 4>(bp.nrm, *nrm, 0);
    fill<float, 4>(bp.zdz, *zdz, 0);
    const long s4[4] = {col->size[0], col->size[1], col->size[2], 4};
    fill_contig<float, 4>(bp.out, out, s4);
    bp.sigma = sigma;
    run_grid(bilateral_denoiser_fwd_kernel, bp, s4[0], s4[1], s4[2], n_threads);
}

extern "C" void ref_bilateral_bwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz, float sigma,
                                  const nvdr_tensor *out_grad, float *col_grad, int n_threads)
{
    BilateralDenoiserParams bp;
    memset((void *)&bp, 0, sizeof(bp));
    fill<float, 4>(bp.col, *col, 0);
    fill<float, 4>(bp.nrm, *nrm, 0);
    fill<float, 4>(bp.zdz, *zdz, 0);
    fill<float, 4>(bp.out_grad, *out_grad, 0);
    const long s3[4] = {col->size[0], col->size[1], col->size[2], 3};
    fill_contig<float, 4>(bp.col_grad, col_grad, s3);
    bp.sigma = sigma;
    run_grid(bilateral_denoiser_bwd_kernel, bp, s3[0], s3[1], s3[2], n_threads);
}
