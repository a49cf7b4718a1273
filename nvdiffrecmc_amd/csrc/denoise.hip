// denoise.hip -- cross-bilateral denoiser, forward and backward, LDS-tiled for gfx950.
//
// Replaces bilateral_denoiser_fwd_kernel / _bwd_kernel (render/optixutils/c_src/denoising.cu:14-130) and
// their launchers (render/optixutils/c_src/torch_bindings.cpp:274-319).
//
// The reference reads 8 floats per tap straight from global memory in 8x8 blocks (529 taps at
// sigma = 2).  Here a 32x16 workgroup (32x8 in the two-image kernels) first stages its (32+2R) x (rows+2R) halo tile in LDS -- the guides
// (normal, depth, depth slope) as five rows of floats per tile row, the colour(s) as 4 / 6 floats per pixel; rounds 1-4: float4 planes
// (col.rgb, z), (nrm.xyz, dz) and a third with the second image -- and walks the window TWO TAPS AT A TIME with their weights in the two
// halves of packed fp32 registers (round 5, below).  Taps that fall
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
// once per workgroup in LDS (per |fy| a row of w_xy and a row of distances over fx) and fetched as broadcast reads instead of being
// recomputed (v_sqrt + v_exp per tap).
// Round 5 (profiles/r05_ab_filter_packed_taps.md): with two wavefronts per SIMD -- the tile's 72 KB of LDS -- the filter is bound by how
// often ONE wavefront can issue an instruction: every ~8.5 cycles whatever its independence, ~10.5 for a packed one (v_pk_mul_f32,
// v_pk_fma_f32, v_pk_add_f32) that does the work of two.  The tap loop therefore computes the weights of taps (fx, fx + 1) in register
// PAIRS -- the seven squarings of pow128, the normal's dot product, the depth difference and the weight products: 37 instead of 56
// vector instructions per two taps, 20 of them packed -- and the guides' layout makes a pair ONE ds_read2_b32 (component k of two
// neighbouring tile pixels), no copies; the colours' pairs (r, g), (b, r2), (g2, b2) take one packed fma per tap each, as the
// compiler already had it.  With the radius a constant (DN_RAD_FIXED = 11, sigma = 2) every tap of a window row is an immediate offset
// from one address register.  Element for element the same operations in the same order: outputs bit-identical (tools/dn_probe.py
// asserts it against the previous kernel).  Two-image kernel at 8 x 512^2: 0.548 / 0.471 -> 0.451 / 0.428 ms forward / backward.  The
// loop over the pairs stays rolled: unrolled by 2 it is 14 % slower, fully 42 % (0.64 / 0.65 ms: the reads of the unrolled row end up
// waited for one by one).
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
// how far the loop over the tap pairs of a window row is unrolled when the radius is a constant (A/B: 0 = the compiler's choice, i.e. all 11)
#ifndef DN_PAIR_UNROLL
#define DN_PAIR_UNROLL 1
#endif
#if DN_PAIR_UNROLL == 0
#define DN_PAIR_LOOP_PRAGMA
#elif DN_PAIR_UNROLL == 1
#define DN_PAIR_LOOP_PRAGMA _Pragma("nounroll")
#elif DN_PAIR_UNROLL == 2
#define DN_PAIR_LOOP_PRAGMA _Pragma("unroll 2")
#else
#define DN_PAIR_LOOP_PRAGMA _Pragma("unroll 4")
#endif
#define DN_RAD_FIXED 11         // 2 * ceil(2.5 sigma) + 1 at sigma = 2 (BilateralDenoiser at full influence): the radius the kernels are also compiled for as a constant

struct DnView {
    View4 col, nrm, zdz;   // col is out_grad in the backward pass
    View4 col2;            // PAIR kernels: the second image filtered with the same guides
    int N, H, W;
};

// tile height of the PAIR kernels: 32 x 8 pixels, 44 bytes per tile pixel = 72 KB of LDS at sigma = 2, two workgroups per CU.  (At eight
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

typedef float dn_v2f __attribute__((ext_vector_type(2)));
#define DN_COL(pair) ((pair) ? 6 : 4)       // colour floats per tile pixel
#define DN_TILE_FLOATS(pair) (5 + DN_COL(pair))

__device__ __forceinline__ float pow128(float x)
{
#pragma unroll
    for (int i = 0; i < 7; ++i) x *= x;
    return x;
}

// PAIR: two images (the diffuse and the specular light of shade(), render.py:120-121) filtered in one pass: the weights depend on the
// guides only, so they are evaluated once per tap and applied to both -- ~32 instead of 2 x 28 VALU instructions per tap.  Same
// arithmetic per image, in the same order: bit-identical to two single calls.
// RAD: the window radius as a compile-time constant (11 = sigma 2, what BilateralDenoiser starts with: the five guide planes, the colours and
// the tap table then sit at immediate offsets from one address register each), 0 = the run-time `rad_`.
template <bool BACKWARD, bool TILED, bool PAIR, int RAD>
__global__ void __launch_bounds__(DN_BX * (PAIR ? DN_BY_PAIR : DN_BY)) bilateral_kernel(DnView v, float sigma, int rad_, float *__restrict__ out,
                                                                                        float *__restrict__ out2)
{
    constexpr int BY = PAIR ? DN_BY_PAIR : DN_BY;
    const int rad = RAD > 0 ? RAD : rad_;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int TW = DN_BX + 2 * rad, TH = BY + 2 * rad, TT = TILED ? TW * TH : 0;
    // The halo tile: the GUIDES as five rows of floats per tile row (normal x / y / z, depth, depth slope: value k of tile pixel (r, c) at
    // (r * 5 + k) * TW + c) -- component k of two neighbouring taps is one ds_read2_b32 into a register PAIR, what the packed instructions
    // of the tap loop want, and with the radius a constant every tap of a window row is an immediate offset (< 256 dwords) from ONE address
    // register -- and the colour(s) as DN_COL(PAIR) floats per pixel (rgb + pad, or rgb | rgb of the second image: the pairs (r, g),
    // (b, r2), (g2, b2) take one packed fma each).  The tap table: per |fy| a row of w_xy and a row of distances (fx = -rad ... rad).
    constexpr int NC = DN_COL(PAIR);
    float *gG = tile, *gCol = tile + 5 * TT;
    float *tap_tab = tile + (5 + NC) * TT;
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
    // the table is symmetric in fy (rows |fy|); a row holds all 2 * rad + 1 values of fx so that the taps fx, fx + 1 sit side by side
    const int side = 2 * rad + 1;
    for (int t = threadIdx.x; t < (rad + 1) * side; t += DN_BX * BY) {
        const int j = t % side, fy = t / side, fx = j - rad;
        const float dist_sqr = (float)(fx * fx + fy * fy);
        tap_tab[fy * 2 * side + j] = __expf(-dist_sqr * inv2var);
        tap_tab[fy * 2 * side + side + j] = sqrtf(dist_sqr);
    }
    if (!TILED) __syncthreads();
    if (TILED) {
        for (int t = threadIdx.x; t < TW * TH; t += DN_BX * BY) {
            const int tx = t % TW, ty = t / TW;
            const int gx = x0 + tx - rad, gy = y0 + ty - rad;
            F3 c = f3(0.0f), nn = f3(0.0f), d = f3(0.0f);
            float z = 0.0f, dz = 0.0f;
            if (gx >= 0 && gy >= 0 && gx < v.W && gy < v.H) {
                c = fetch3(v.col, n, gy, gx);
                nn = fetch3(v.nrm, n, gy, gx);
                const float *zp = v.zdz.p + n * v.zdz.s0 + gy * v.zdz.s1 + gx * v.zdz.s2;
                z = zp[0];
                dz = zp[v.zdz.s3];
                if (PAIR) d = fetch3(v.col2, n, gy, gx);
            }
            float *gq = gG + ty * 5 * TW + tx;
            gq[0] = nn.x; gq[TW] = nn.y; gq[2 * TW] = nn.z; gq[3 * TW] = z; gq[4 * TW] = dz;
            if (PAIR) {
                float2 *q = (float2 *)(gCol + 6 * t);
                q[0] = make_float2(c.x, c.y); q[1] = make_float2(c.z, d.x); q[2] = make_float2(d.y, d.z);
            } else {
                *(float4 *)(gCol + 4 * t) = make_float4(c.x, c.y, c.z, 0.0f);
            }
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
        const float *gq = gG + (ly + rad) * 5 * TW + lx + rad;
        cA = make_float4(0.0f, 0.0f, 0.0f, gq[3 * TW]);     // (the centre's colour is not used: the centre is a tap like the others)
        cB = make_float4(gq[0], gq[TW], gq[2 * TW], gq[4 * TW]);
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
    // TWO taps of a row at a time, their weights in the two halves of packed fp32 registers (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32): with
    // two wavefronts per SIMD (the tile's 78 KB of LDS) the filter is bound by how often ONE wavefront can issue -- every ~8.5 cycles, ~10.5
    // for a packed instruction that does two taps' worth -- so the seven squarings of pow128, the normal dot product, the depth difference
    // and the weight products cost half as many issues.  The guide values of the two taps are read from LDS as PAIRS (component k of tap t
    // and of tap t + 1: ds_read2_b32 with the two offsets), which is what puts them into the register pairs the packed instructions want
    // without a copy.  Element for element the same operations in the same order as tap(); the sums still run in tap order.  (A window
    // row has 2 * rad + 1 taps: pairs, then one single.)
    if (TILED) {
        const dn_v2f cBx = {cB.x, cB.x}, cBy = {cB.y, cB.y}, cBz = {cB.z, cB.z}, cBw = {cB.w, cB.w}, ncAw = {-cA.w, -cA.w};
        dn_v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, s45 = {0.f, 0.f};       // (ax, ay), (az, bx), (by, bz)  [single image: (ax, ay), (az, -)]
        for (int fy = -rad; fy <= rad; ++fy) {
            const float *g = gG + (ly + rad + fy) * 5 * TW + lx;           // tap j = fx + rad of this row: value k at g[k * TW + j]
            const float *c = gCol + NC * ((ly + rad + fy) * TW + lx);      // its colour(s) at c[NC * j ...]
            const float *tw = tap_tab + (fy < 0 ? -fy : fy) * 2 * side;    // w_xy at tw[j], the distance at tw[side + j]
            int j = 0;
            DN_PAIR_LOOP_PRAGMA
            for (; j + 1 < side; j += 2) {
                dn_v2f d = dn_v2f{g[j], g[j + 1]} * cBx;
                d = __builtin_elementwise_fma(dn_v2f{g[TW + j], g[TW + j + 1]}, cBy, d);
                d = __builtin_elementwise_fma(dn_v2f{g[2 * TW + j], g[2 * TW + j + 1]}, cBz, d);
                dn_v2f wn = {fminf(fmaxf(d.x, DN_EPS), 1.0f), fminf(fmaxf(d.y, DN_EPS), 1.0f)};
#pragma unroll
                for (int i = 0; i < 7; ++i) wn = wn * wn;
                const dn_v2f dist = {tw[side + j], tw[side + j + 1]}, wxy = {tw[j], tw[j + 1]};
                const dn_v2f dzd = (BACKWARD ? dn_v2f{g[4 * TW + j], g[4 * TW + j + 1]} : cBw) * dist;
                const dn_v2f dif = dn_v2f{g[3 * TW + j], g[3 * TW + j + 1]} + ncAw;          // tap depth - centre depth
                const float e0 = -(fabsf(dif.x) * __builtin_amdgcn_rcpf(fmaxf(dzd.x, DN_EPS)));
                const float e1 = -(fabsf(dif.y) * __builtin_amdgcn_rcpf(fmaxf(dzd.y, DN_EPS)));
                const dn_v2f w = (wxy * wn) * dn_v2f{__expf(e0), __expf(e1)};        // (__expf: v_exp_f32 of the argument times log2 e)
                const dn_v2f w0 = {w.x, w.x}, w1 = {w.y, w.y};
                const float *cj = c + NC * j;
                s01 = __builtin_elementwise_fma(dn_v2f{cj[0], cj[1]}, w0, s01);
                s23 = __builtin_elementwise_fma(dn_v2f{cj[2], cj[3]}, w0, s23);
                if (PAIR) s45 = __builtin_elementwise_fma(dn_v2f{cj[4], cj[5]}, w0, s45);
                aw += w.x;
                s01 = __builtin_elementwise_fma(dn_v2f{cj[NC], cj[NC + 1]}, w1, s01);
                s23 = __builtin_elementwise_fma(dn_v2f{cj[NC + 2], cj[NC + 3]}, w1, s23);
                if (PAIR) s45 = __builtin_elementwise_fma(dn_v2f{cj[NC + 4], cj[NC + 5]}, w1, s45);
                aw += w.y;
            }
            {   // the row's last tap (j == 2 * rad) on its own
                const float *cj = c + NC * j;
                ax = s01.x; ay = s01.y; az = s23.x;
                if (PAIR) { bx = s23.y; by = s45.x; bz = s45.y; }
                tap(make_float4(cj[0], cj[1], cj[2], g[3 * TW + j]), make_float4(g[j], g[TW + j], g[2 * TW + j], g[4 * TW + j]),
                    PAIR ? make_float4(cj[3], cj[4], cj[5], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f), make_float2(tw[j], tw[side + j]));
                s01 = dn_v2f{ax, ay};
                s23 = dn_v2f{az, PAIR ? bx : 0.f};
                if (PAIR) s45 = dn_v2f{by, bz};
            }
        }
        ax = s01.x; ay = s01.y; az = s23.x;
        if (PAIR) { bx = s23.y; by = s45.x; bz = s45.y; }
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
                    make_float2(tap_tab[(fy < 0 ? -fy : fy) * 2 * side + rad + fx], tap_tab[(fy < 0 ? -fy : fy) * 2 * side + side + rad + fx]));
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
    const size_t lds_tab = (size_t)(rad + 1) * (2 * rad + 1) * sizeof(float2);
    const size_t lds_tile = (size_t)(DN_BX + 2 * rad) * (by + 2 * rad) * DN_TILE_FLOATS(pair) * sizeof(float);
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
        auto raise = [](const void *f, int kb) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024) == hipSuccess; };
        big_lds = raise((const void *)bilateral_kernel<false, true, false, 0>, DN_LDS_KB) && raise((const void *)bilateral_kernel<true, true, false, 0>, DN_LDS_KB) &&
                  raise((const void *)bilateral_kernel<false, true, false, DN_RAD_FIXED>, DN_LDS_KB) && raise((const void *)bilateral_kernel<true, true, false, DN_RAD_FIXED>, DN_LDS_KB) &&
                  raise((const void *)bilateral_kernel<false, true, true, 0>, DN_LDS_KB_PAIR) && raise((const void *)bilateral_kernel<true, true, true, 0>, DN_LDS_KB_PAIR) &&
                  raise((const void *)bilateral_kernel<false, true, true, DN_RAD_FIXED>, DN_LDS_KB_PAIR) && raise((const void *)bilateral_kernel<true, true, true, DN_RAD_FIXED>, DN_LDS_KB_PAIR);
        if (!big_lds) (void)hipGetLastError();
    }
    const bool tiled = lds_tile + lds_tab <= (size_t)(big_lds ? (pair ? DN_LDS_KB_PAIR : DN_LDS_KB) : 64) * 1024;
    NVDR_REQUIRE(lds_tab <= 64 * 1024, "%s: sigma %g needs a %d-wide window, more than fits", op, (double)sigma, 2 * rad + 1);
    const size_t lds = (tiled ? lds_tile : 0) + lds_tab;
    const unsigned threads = DN_BX * by;
    const bool fixed = tiled && rad == DN_RAD_FIXED;        // the instantiation with the radius as a constant
#define DN_LAUNCH(B, T, P, R) bilateral_kernel<B, T, P, R><<<grid, threads, lds, stream>>>(v, sigma, rad, out, P ? out2 : nullptr)
#define DN_LAUNCH_T(B, P) do { if (fixed) DN_LAUNCH(B, true, P, DN_RAD_FIXED); else if (tiled) DN_LAUNCH(B, true, P, 0); else DN_LAUNCH(B, false, P, 0); } while (0)
    if (pair) {
        if (backward) DN_LAUNCH_T(true, true); else DN_LAUNCH_T(false, true);
    } else {
        if (backward) DN_LAUNCH_T(true, false); else DN_LAUNCH_T(false, false);
    }
#undef DN_LAUNCH_T
#undef DN_LAUNCH
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
