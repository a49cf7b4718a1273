// denoise.hip -- cross-bilateral denoiser, forward and backward, LDS-tiled for gfx950.
//
// Replaces bilateral_denoiser_fwd_kernel / _bwd_kernel (render/optixutils/c_src/denoising.cu:14-130) and
// their launchers (render/optixutils/c_src/torch_bindings.cpp:274-319).
//
// The reference reads 8 floats per tap straight from global memory in 8x8 blocks (529 taps at
// sigma = 2).  Here a 32x16 workgroup (32x8 in the two-image kernels) first stages its (32+2R) x (rows+2R) halo tile in LDS as two
// float4 planes -- (col.rgb | out_grad.rgb, z) and (nrm.xyz, dz); a third with the second image -- so that a tap costs two (three)
// conflict-free 16-byte LDS reads (half-wave = one row of 32 consecutive pixels).  Taps that fall
// outside the image are zero-filled: a zero normal gives clamp(dot,1e-4,1)^128 == 0 exactly, which
// reproduces the reference's `continue` (denoising.cu:39-40).
// The same identity gives the background early-out: a centre pixel whose normal is exactly zero (every pixel the
// rasteriser did not cover, render.py:99-104) has w == 0 on all taps, forward and backward, so its result is
// (0,0,0,1e-4) / (0,0,0) without looking at a single tap.  Workgroups and wavefronts made only of such pixels skip the
// tile load / the tap loop (bob covers 23 % of the frame).  (Only non-finite colours would tell the difference: 0*inf.)
// (Tried: one kernel for the diffuse and the specular image -- same guides, so one weight evaluation -- with the second
// image's taps read from global memory because two colour planes do not fit the 64 KB tile budget: forward 0.75 ms
// instead of 2 x 0.47, but backward 1.33 ms instead of 2 x 0.43 at 8 x 512^2; dropped.)
// (Tried in round 3 and dropped: the radius as a template parameter + TWO vertically adjacent pixels per thread, so that every
// tile element read from LDS serves two (pixel, tap) pairs -- half the LDS reads per tap, bit-identical output: 0.474 / 0.451 ms
// instead of 0.468 / 0.434 ms per 8-view launch.  The filter is bound by its ~28 VALU instructions per tap, not by LDS reads,
// and the coarser four-row background early-out costs more than the reads saved.)
// The per-tap constants exp(-d^2/2s^2) and d are wave-uniform; gfx950 has no scalar float unit, so they are tabulated
// once per workgroup in LDS (one quadrant: they depend on |fx|, |fy|) and fetched as broadcast reads instead of being recomputed
// (v_sqrt + v_exp per tap).
//   forward : w = w_xy * w_n * exp(-|z_t - z_c| / max(dz_c * dist, 1e-4)),  out = (sum w*col_t, max(sum w, 1e-4))
//   backward: the transposed gather with the TAP's dz in the denominator (denoising.cu:118).
#include "common.h"

#define DN_BX 32
#ifndef DN_BY
#define DN_BY 16
#endif
#ifndef DN_LDS_KB
#define DN_LDS_KB 80            // tile + tap table of one workgroup: two workgroups per CU (160 KB of LDS)
#endif
#define DN_EPS 0.0001f

struct DnView {
    View4 col, nrm, zdz;   // col is out_grad in the backward pass
    View4 col2;            // PAIR kernels: the second image filtered with the same guides
    int N, H, W;
};

// tile height of the PAIR kernels: 32 x 8 pixels, three float4 planes = 78 KB of LDS at sigma = 2, two workgroups per CU.  (At eight
// 512x512 views the height does not matter -- 8 rows 0.605 / 0.525 ms forward / backward, 32 rows 0.610 / 0.538, 16 rows (one
// workgroup per CU) 0.70 / 0.64 -- but ONE view has only ~80 live 32x32 tiles for 256 CUs: 0.181 / 0.128 ms against 0.133 / 0.099.)
// (Tried and dropped: issuing the LDS reads of the next 1, 2 or 4 taps ahead of the arithmetic, with whole ds_read_b128 instead of
// the ds_read_b96 the compiler narrows two of the three reads to: +10 % at every depth, one view or eight.  The tap loop is bound
// by its ~36 VALU instructions, which the compiler's own two-tap unrolling already overlaps with the reads.)
#ifndef DN_BY_PAIR
#define DN_BY_PAIR 8
#endif
#ifndef DN_LDS_KB_PAIR
#define DN_LDS_KB_PAIR 80
#endif

__device__ __forceinline__ float pow128(float x)
{
#pragma unroll
    for (int i = 0; i < 7; ++i) x *= x;
    return x;
}

// PAIR: two images (the diffuse and the specular light of shade(), render.py:120-121) filtered in one pass: the weights depend on the
// guides only, so they are evaluated once per tap and applied to both -- ~32 instead of 2 x 28 VALU instructions per tap.  Same
// arithmetic per image, in the same order: bit-identical to two single calls.
template <bool BACKWARD, bool TILED, bool PAIR>
__global__ void __launch_bounds__(DN_BX * (PAIR ? DN_BY_PAIR : DN_BY)) bilateral_kernel(DnView v, float sigma, int rad, float *__restrict__ out,
                                                                                        float *__restrict__ out2)
{
    constexpr int BY = PAIR ? DN_BY_PAIR : DN_BY;
    extern __shared__ __attribute__((aligned(16))) float4 tile[];
    const int TW = DN_BX + 2 * rad, TH = BY + 2 * rad;
    float4 *tA = tile, *tB = tile + (TILED ? TW * TH : 0), *tC = tile + (TILED ? 2 * TW * TH : 0);
    float2 *tap_tab = (float2 *)(tile + (TILED ? (PAIR ? 3 : 2) * TW * TH : 0));   // (w_xy, dist) per tap
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * DN_BX, y0 = blockIdx.y * BY;
    const int lx = threadIdx.x & (DN_BX - 1), ly = threadIdx.x / DN_BX;
    const int x = x0 + lx, y = y0 + ly;
    const bool inside = x < v.W && y < v.H;
    const int64_t o = ((int64_t)n * v.H + y) * v.W + x;
    // background early-out, workgroup level (before the tile is staged)
    F3 cn = f3(0.0f);
    if (inside) cn = fetch3(v.nrm, n, y, x);
    const bool live = cn.x != 0.0f || cn.y != 0.0f || cn.z != 0.0f;
    auto write_background = [&]() {
        if (BACKWARD) {
            out[3 * o + 0] = 0.f; out[3 * o + 1] = 0.f; out[3 * o + 2] = 0.f;
            if (PAIR) { out2[3 * o + 0] = 0.f; out2[3 * o + 1] = 0.f; out2[3 * o + 2] = 0.f; }
        } else {
            out[4 * o + 0] = 0.f; out[4 * o + 1] = 0.f; out[4 * o + 2] = 0.f; out[4 * o + 3] = DN_EPS;
            if (PAIR) { out2[4 * o + 0] = 0.f; out2[4 * o + 1] = 0.f; out2[4 * o + 2] = 0.f; out2[4 * o + 3] = DN_EPS; }
        }
    };
    if (!__syncthreads_or(live)) {
        if (inside) write_background();
        return;
    }
    const float inv2var = 1.0f / (2.0f * sigma * sigma);
    const int side = rad + 1;                               // the table is symmetric in fx and fy: one quadrant
    for (int t = threadIdx.x; t < side * side; t += DN_BX * BY) {
        const int fx = t % side, fy = t / side;
        const float dist_sqr = (float)(fx * fx + fy * fy);
        tap_tab[t] = make_float2(__expf(-dist_sqr * inv2var), sqrtf(dist_sqr));
    }
    if (!TILED) __syncthreads();
    if (TILED) {
        for (int t = threadIdx.x; t < TW * TH; t += DN_BX * BY) {
            const int tx = t % TW, ty = t / TW;
            const int gx = x0 + tx - rad, gy = y0 + ty - rad;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c2 = a;
            if (gx >= 0 && gy >= 0 && gx < v.W && gy < v.H) {
                const F3 c = fetch3(v.col, n, gy, gx), nn = fetch3(v.nrm, n, gy, gx);
                const float *zp = v.zdz.p + n * v.zdz.s0 + gy * v.zdz.s1 + gx * v.zdz.s2;
                a = make_float4(c.x, c.y, c.z, zp[0]);
                b = make_float4(nn.x, nn.y, nn.z, zp[v.zdz.s3]);
                if (PAIR) {
                    const F3 d = fetch3(v.col2, n, gy, gx);
                    c2 = make_float4(d.x, d.y, d.z, 0.f);
                }
            }
            tA[t] = a;
            tB[t] = b;
            if (PAIR) tC[t] = c2;
        }
        __syncthreads();
    }
    if (!inside) return;
    // wavefront-level early-out (two rows of 32 pixels)
    if (__ballot(live) == 0ull) {
        write_background();
        return;
    }
    float4 cA, cB;
    if (TILED) {
        cA = tA[(ly + rad) * TW + lx + rad];
        cB = tB[(ly + rad) * TW + lx + rad];
    } else {
        const F3 c = fetch3(v.col, n, y, x), nn = fetch3(v.nrm, n, y, x);
        const float *zp = v.zdz.p + n * v.zdz.s0 + y * v.zdz.s1 + x * v.zdz.s2;
        cA = make_float4(c.x, c.y, c.z, zp[0]);
        cB = make_float4(nn.x, nn.y, nn.z, zp[v.zdz.s3]);
    }
    float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
    float bx = 0.f, by = 0.f, bz = 0.f;                     // PAIR: the second image's sums (the weight sum is shared)
    // one tap: its weight from the guides, applied to the colour(s).  The sums run in tap order (rows of the window, left to right).
    auto tap = [&](const float4 &tAv, const float4 &tBv, const float4 &tCv, const float2 &tt) {
        const float w_xy = tt.x, dist = tt.y;
        // explicit fused multiply-adds (the library is built with -ffp-contract=off for the sampling code; this filter is compared at 2e-5 --
        // it already takes the hardware's exp and reciprocal -- and the reference's nvcc build contracts the same expressions): nine VALU
        // instructions less per tap, of ~36
        const float d = fmaf(tBv.z, cB.z, fmaf(tBv.y, cB.y, tBv.x * cB.x));
        const float w_normal = pow128(fminf(fmaxf(d, DN_EPS), 1.0f));
        const float dz = BACKWARD ? tBv.w : cB.w;
        const float w_depth = __expf(-(fabsf(tAv.w - cA.w) * __builtin_amdgcn_rcpf(fmaxf(dz * dist, DN_EPS))));
        const float w = w_xy * w_normal * w_depth;
        ax = fmaf(tAv.x, w, ax);
        ay = fmaf(tAv.y, w, ay);
        az = fmaf(tAv.z, w, az);
        aw += w;
        if (PAIR) {
            bx = fmaf(tCv.x, w, bx);
            by = fmaf(tCv.y, w, by);
            bz = fmaf(tCv.z, w, bz);
        }
    };
    if (TILED) {
        for (int fy = -rad; fy <= rad; ++fy)
            for (int fx = -rad; fx <= rad; ++fx) {
                const int t = (ly + rad + fy) * TW + lx + rad + fx;
                tap(tA[t], tB[t], PAIR ? tC[t] : make_float4(0.f, 0.f, 0.f, 0.f), tap_tab[(fy < 0 ? -fy : fy) * side + (fx < 0 ? -fx : fx)]);
            }
    } else {
        for (int fy = -rad; fy <= rad; ++fy) {
            for (int fx = -rad; fx <= rad; ++fx) {
                const int gx = x + fx, gy = y + fy;
                if (gx < 0 || gy < 0 || gx >= v.W || gy >= v.H) continue;
                const F3 c = fetch3(v.col, n, gy, gx), nn = fetch3(v.nrm, n, gy, gx);
                const float *zp = v.zdz.p + n * v.zdz.s0 + gy * v.zdz.s1 + gx * v.zdz.s2;
                float4 tCv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (PAIR) {
                    const F3 d = fetch3(v.col2, n, gy, gx);
                    tCv = make_float4(d.x, d.y, d.z, 0.f);
                }
                tap(make_float4(c.x, c.y, c.z, zp[0]), make_float4(nn.x, nn.y, nn.z, zp[v.zdz.s3]), tCv,
                    tap_tab[(fy < 0 ? -fy : fy) * side + (fx < 0 ? -fx : fx)]);
            }
        }
    }
    if (BACKWARD) {
        out[3 * o + 0] = ax; out[3 * o + 1] = ay; out[3 * o + 2] = az;
        if (PAIR) { out2[3 * o + 0] = bx; out2[3 * o + 1] = by; out2[3 * o + 2] = bz; }
    } else {
        out[4 * o + 0] = ax; out[4 * o + 1] = ay; out[4 * o + 2] = az;
        out[4 * o + 3] = fmaxf(aw, DN_EPS);
        if (PAIR) { out2[4 * o + 0] = bx; out2[4 * o + 1] = by; out2[4 * o + 2] = bz; out2[4 * o + 3] = fmaxf(aw, DN_EPS); }
    }
}


static int check_dn(const nvdr_tensor *t, int64_t N, int64_t H, int64_t W, int c, const char *op, const char *name)
{
    NVDR_REQUIRE(t && t->data, "%s: %s is NULL", op, name);
    NVDR_REQUIRE((t->size[0] == N || t->size[0] == 1) && (t->size[1] == H || t->size[1] == 1) &&
                     (t->size[2] == W || t->size[2] == 1) && (t->size[3] >= c || t->size[3] == 1),
                 "%s: %s has shape [%lld,%lld,%lld,%lld], expected [%lld,%lld,%lld,%d]", op, name, (long long)t->size[0],
                 (long long)t->size[1], (long long)t->size[2], (long long)t->size[3], (long long)N, (long long)H,
                 (long long)W, c);
    return 0;
}

static int launch_bilateral(const nvdr_tensor *col_or_grad, const nvdr_tensor *col2_or_grad2, const nvdr_tensor *col_shape, const nvdr_tensor *nrm,
                            const nvdr_tensor *zdz, float sigma, bool backward, float *out, float *out2, hipStream_t stream,
                            const char *op)
{
    NvdrRange range(op);
    const bool pair = col2_or_grad2 != nullptr;
    const int64_t N = col_shape->size[0], H = col_shape->size[1], W = col_shape->size[2];
    NVDR_REQUIRE(sigma > 0.0f, "%s: sigma must be positive", op);
    int r;
    if ((r = check_dn(col_or_grad, N, H, W, 3, op, backward ? "out_grad" : "col"))) return r;
    if (pair && (r = check_dn(col2_or_grad2, N, H, W, 3, op, backward ? "out_grad2" : "col2"))) return r;
    if ((r = check_dn(nrm, N, H, W, 3, op, "nrm"))) return r;
    if ((r = check_dn(zdz, N, H, W, 2, op, "zdz"))) return r;
    if (N * H * W == 0) return 0;
    DnView v;
    v.col = make_view4(*col_or_grad);
    v.col2 = make_view4(pair ? *col2_or_grad2 : *col_or_grad);
    v.nrm = make_view4(*nrm);
    v.zdz = make_view4(*zdz);
    v.N = (int)N; v.H = (int)H; v.W = (int)W;
    const int rad = 2 * (int)ceil((double)sigma * 2.5) + 1; // denoising.cu:27
    const int by = pair ? DN_BY_PAIR : DN_BY;
    const size_t lds_tab = (size_t)(rad + 1) * (rad + 1) * sizeof(float2);
    const size_t lds_tile = (size_t)(DN_BX + 2 * rad) * (by + 2 * rad) * (pair ? 3 : 2) * sizeof(float4);
    dim3 grid(div_up(W, DN_BX), div_up(H, by), (unsigned)N);
    // beyond 64 KB of dynamic LDS a kernel needs the attribute
    // ... per DEVICE (a process that drives several GPUs sets it on each; the first version remembered one process-wide flag)
    static bool big_lds_dev[64] = {}, big_lds_tried_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    dev_id = dev_id < 0 || dev_id >= 64 ? 0 : dev_id;
    bool &big_lds = big_lds_dev[dev_id], &big_lds_tried = big_lds_tried_dev[dev_id];
    if (!big_lds_tried) {
        big_lds_tried = true;
        big_lds = hipFuncSetAttribute((const void *)bilateral_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DN_LDS_KB * 1024) == hipSuccess &&
                  hipFuncSetAttribute((const void *)bilateral_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DN_LDS_KB * 1024) == hipSuccess &&
                  hipFuncSetAttribute((const void *)bilateral_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DN_LDS_KB_PAIR * 1024) == hipSuccess &&
                  hipFuncSetAttribute((const void *)bilateral_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DN_LDS_KB_PAIR * 1024) == hipSuccess;
        if (!big_lds) (void)hipGetLastError();
    }
    const bool tiled = lds_tile + lds_tab <= (size_t)(big_lds ? (pair ? DN_LDS_KB_PAIR : DN_LDS_KB) : 64) * 1024;
    NVDR_REQUIRE(lds_tab <= 64 * 1024, "%s: sigma %g needs a %d-wide window, more than fits", op, (double)sigma, 2 * rad + 1);
    const size_t lds = (tiled ? lds_tile : 0) + lds_tab;
    const unsigned threads = DN_BX * by;
    if (pair) {
        if (backward) {
            if (tiled) bilateral_kernel<true, true, true><<<grid, threads, lds, stream>>>(v, sigma, rad, out, out2);
            else bilateral_kernel<true, false, true><<<grid, threads, lds, stream>>>(v, sigma, rad, out, out2);
        } else {
            if (tiled) bilateral_kernel<false, true, true><<<grid, threads, lds, stream>>>(v, sigma, rad, out, out2);
            else bilateral_kernel<false, false, true><<<grid, threads, lds, stream>>>(v, sigma, rad, out, out2);
        }
    } else if (backward) {
        if (tiled) bilateral_kernel<true, true, false><<<grid, threads, lds, stream>>>(v, sigma, rad, out, nullptr);
        else bilateral_kernel<true, false, false><<<grid, threads, lds, stream>>>(v, sigma, rad, out, nullptr);
    } else {
        if (tiled) bilateral_kernel<false, true, false><<<grid, threads, lds, stream>>>(v, sigma, rad, out, nullptr);
        else bilateral_kernel<false, false, false><<<grid, threads, lds, stream>>>(v, sigma, rad, out, nullptr);
    }
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_bilateral_denoiser_fwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz,
                                           float sigma, float *out, void *stream)
{
    NVDR_REQUIRE(col && nrm && zdz && out, "bilateral_denoiser_fwd: NULL argument");
    return launch_bilateral(col, nullptr, col, nrm, zdz, sigma, false, out, nullptr, (hipStream_t)stream, "bilateral_denoiser_fwd");
}

extern "C" int nvdr_bilateral_denoiser_bwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz,
                                           float sigma, const nvdr_tensor *out_grad, float *col_grad, void *stream)
{
    NVDR_REQUIRE(col && nrm && zdz && out_grad && col_grad, "bilateral_denoiser_bwd: NULL argument");
    return launch_bilateral(out_grad, nullptr, col, nrm, zdz, sigma, true, col_grad, nullptr, (hipStream_t)stream, "bilateral_denoiser_bwd");
}

// two images with the same guides in one pass (additive: the diffuse and the specular light of shade(), render.py:120-121)
extern "C" int nvdr_bilateral_denoiser_pair_fwd(const nvdr_tensor *col_a, const nvdr_tensor *col_b, const nvdr_tensor *nrm, const nvdr_tensor *zdz,
                                                float sigma, float *out_a, float *out_b, void *stream)
{
    NVDR_REQUIRE(col_a && col_b && nrm && zdz && out_a && out_b, "bilateral_denoiser_pair_fwd: NULL argument");
    return launch_bilateral(col_a, col_b, col_a, nrm, zdz, sigma, false, out_a, out_b, (hipStream_t)stream, "bilateral_denoiser_pair_fwd");
}

extern "C" int nvdr_bilateral_denoiser_pair_bwd(const nvdr_tensor *col, const nvdr_tensor *nrm, const nvdr_tensor *zdz, float sigma,
                                                const nvdr_tensor *out_grad_a, const nvdr_tensor *out_grad_b, float *col_grad_a,
                                                float *col_grad_b, void *stream)
{
    NVDR_REQUIRE(col && nrm && zdz && out_grad_a && out_grad_b && col_grad_a && col_grad_b, "bilateral_denoiser_pair_bwd: NULL argument");
    return launch_bilateral(out_grad_a, out_grad_b, col, nrm, zdz, sigma, true, col_grad_a, col_grad_b, (hipStream_t)stream, "bilateral_denoiser_pair_bwd");
}
