// env_shade.hip -- Monte-Carlo direct lighting (forward and backward) as three kernels around a ray stream.
//
// Replaces the OptiX raygen program __raygen__rg (render/optixutils/c_src/envsampling/kernel.cu:463-542,
// helpers :30-461) and its host launchers env_shade_fwd / env_shade_bwd
// (render/optixutils/c_src/torch_bindings.cpp:123-272).
//
// MI355X design (the reference runs ONE thread per pixel with a serial 2*S-ray loop inside one OptiX program):
//   * covered pixels are compacted into a work list first (bob covers ~20 % of the frame);
//   * the raygen program is cut at its two natural seams into THREE lean kernels that hand a ray stream
//     through HBM (16 B + 4 B + 1 B per slot -- ~170 MB per pass and view at 512^2 x 64 spp):
//       1. env_gen_kernel    sample generation: a pixel is owned by L = min(64, pow2ceil(S)) lanes, every lane
//                            owns whole strata and jumps the pixel's LCG stream ahead to its stratum (5 draws per
//                            stratum, kernel.cu:513-524): bit-identical random sequence to the serial loop.  Samples
//                            under the shading horizon contribute exactly zero and are left out of the LIVE-RAY LIST
//                            the next stage walks;
//       2. env_trace_kernel  (trace_kernel.h) PERSISTENT wavefronts over the live-ray list, chunks of 256 rays claimed from 64
//                            device counters: a lane that finishes its ray takes the next one of its wave (wave-uniform
//                            cursor), so lanes never idle behind the slowest ray of a pixel; eight-wide compressed nodes,
//                            unordered any-hit descent over (child group, hit mask) pairs stacked in LDS, triangle tests
//                            deferred to a per-wavefront queue and run 64 at a time;
//       3. env_shade_kernel  BSDF evaluation (forward) or hand-derived gradients (backward) per live sample, reduced
//                            across the L lanes with shuffle butterflies; the light gradient leaves (texel, rgb) records
//                            in the stream that an LDS band gather reduces (no global atomics; kernel.cu:208-210).
//     Why not one fused kernel (the first version, 2.1 ms forward): rocprofv3 showed the traversal VALU-bound at ~40 %
//     active lanes and the fused kernel spilling 67 VGPRs + 146 SGPRs; the split keeps each stage in registers and
//     lets the traversal refill lanes across pixel boundaries.
// The sampling math mirrors the evaluation order and the fp32/fp64 promotions of the reference
// source and uses include/nvdr_detmath.h for sin/cos/acos/atan2, which makes every
// discrete decision (texel, lobe, visibility) bit-identical to the CPU oracle.
#include "trace_kernel.h"
#include "bsdf_device.h"

#define NVDR_PI_DBL 3.14159265358979323846
// Division of a FLOAT by pi or 2 pi in double precision, rounded back to float (the reference's `x / M_PI` with float x): the double
// product with the rounded reciprocal gives the same float for EVERY finite float x -- checked exhaustively, all 4 278 190 080 of them,
// also under the `+ 0.5` of dir_to_tc and the fmax(1e-6f, .) of cosine_sample (tools/div_by_constant_check.c,
// profiles/r03_div_by_constant_check.txt).  One v_mul_f64 instead of the ~14 fp64 instructions of an IEEE division.
#define NVDR_INV_PI_DBL (1.0 / NVDR_PI_DBL)
#define NVDR_INV_2PI_DBL (1.0 / (2.0 * NVDR_PI_DBL))

// minimum waves per SIMD the backward shading kernel is compiled for (register budget = 512 / this).  Measured in one GPU
// session (8-view launch, backward shading + light-gradient gather): unrolled sample loop, 192 VGPRs, 2 waves/SIMD 4.85 ms;
// rolled loop 170 VGPRs 4.75 ms; rolled + 3 waves/SIMD (168 VGPRs, 3 dwords spilled) 4.07 ms; rolled + 4 waves/SIMD (128
// VGPRs, 43 spilled) 5.30 ms.
#ifndef NVDR_SHADE_OCC
#define NVDR_SHADE_OCC 3
#endif
// 1: the two samples of a stratum are shaded by a rolled loop in the backward kernel (half the code, fewer live registers)
#ifndef NVDR_BWD_ROLL
#define NVDR_BWD_ROLL 1
#endif
// measurements only (variants): 1 = no light-gradient records at all, 2 = none + round-robin groups, 3 = unsorted in-place records + round-robin
#ifndef NVDR_CDF_GUIDE
#define NVDR_CDF_GUIDE 1          // CDF inversion through guide tables (0: the reference's bisection; A/B)
#endif
#ifndef NVDR_LG_EXPERIMENT
#define NVDR_LG_EXPERIMENT 0
#endif
#ifndef NVDR_GEN_OCC
#define NVDR_GEN_OCC 4   // 128 VGPRs (14 dwords spilled) instead of 150: 4 waves per SIMD, -3 % time
#endif


struct Tab {          // small strided table (light, pdf, rows, cols)
    const float *p;
    int s0, s1, s2;
    int n0, n1, n2;
};

// A strided [N,H,W,3] view with 32-BIT element strides (the launcher checks that they fit): six scalar registers instead of the eleven of
// common.h's View4.  The parameter block below is kept in scalar registers by every env-shade kernel, and at 88 of them for the eight views
// alone the compiler spilled scalars into vector lanes -- 267 of the 1 965 vector instructions of the backward shading kernel were
// v_readlane / v_writelane of that spill traffic (round 5).
struct View4s {
    const float *p;
    int s0, s1, s2, s3;     // element strides, already zeroed for broadcast dims
};
__device__ __forceinline__ F3 fetch3(const View4s &v, int n, int h, int w)
{
    const float *q = v.p + ((int64_t)n * v.s0 + (int64_t)h * v.s1 + (int64_t)w * v.s2);
    return f3(q[0], q[v.s3], q[2 * v.s3]);
}

struct ShadeParams {
    View4s ro, pos, nrm, view_pos, kd, ks, dgrad, sgrad;
    const float *mask; int ms0, ms1, ms2;
    Tab light, pdf, rows, cols;
    const int *perms; int perm_s0, perm_s1; unsigned n_perms;
    int N, H, W;
    unsigned bsdf, n, S, seed, pix_offset;
    float shadow_scale;
    int L, log2L;             // lanes per pixel
    float *diff, *spec;
    float *g_pos, *g_nrm, *g_kd, *g_ks, *g_light;
    uint32_t *vis_cache; int vis_words; // caller's bit planes (optional)
    const int *pix_list;
    const unsigned *pix_count;
    // ray stream between the three stages (context scratch, indexed by compacted pixel * 2S + 2*stratum + r)
    float4 *rays;             // (dir.xyz, pdf_light + pdf_bsdf)
    int *texel;               // ty * Wl + tx of the radiance lookup
    float4 *pix_origin;       // shadow-ray origin per compacted pixel
    float4 *pix_grad;         // backward: (diff_grad, spec_grad) of a compacted pixel (2 x float4), packed by pack_grads_kernel
    float4 *pix_setup;        // the G-buffer values of a compacted pixel every sample of it is shaded with, packed by stage 1 (4 x float4: see load_setup)
    uint8_t *vis;             // 1 = unoccluded
    uint32_t *live;           // compacted list of the stream slots stage 2 has to traverse: NVDR_LIVE_SEGS dense segments (trace_kernel.h)
    unsigned *ray_count;      // the chunk's block of counters: the segments' lengths, [NVDR_LIVE_SEGS] their capacity
    float *g_light_xcd;       // [8][Hl*Wl*3] per-XCD private light-gradient accumulators (backward, atomics mode only)
    int light_elems;          // Hl*Wl*3
    unsigned debug;           // NVDR_DEBUG bits (read once per context): 1 skip tracing, 2 skip the light gradient
    // the chunk of the covered-pixel list this launch of the three stages works on: pixels [pix_begin, pix_begin + pix_cap)
    unsigned pix_begin, pix_cap;
    const unsigned *seed_dev; // optional: a seed offset read from device memory (captured HIP graphs replay with fresh seeds)
    int reuse;                // backward: the forward's stream is still in the context IF the whole launch fitted one chunk
    int lg_records;           // backward: 1 = write (texel, rgb) records for the band gather, 0 = global atomics
    int lg_shift;             // band of a texel = texel >> lg_shift (bands hold a power-of-two number of texels)
    const uint16_t *cdf_guide; // [Hl + 1][1 << guide_log] first CDF entry above k / 2^guide_log: row y of the column CDFs, last row = the row CDF
    int guide_log_rows, guide_log_cols;
    uint16_t *lg_tags;        // per block of 128 stream slots: band | fill << 8 of the light-gradient records it holds (0xFFFF: none)
    unsigned lg_spare_base;   // first spare block (behind the chunk's own slots); wavefront w owns lg_spw of them from lg_spare_base + w * lg_spw
    unsigned lg_spw;
    unsigned *queues;         // chunk counters of the traversal kernel: stages 1 and 3 leave them zeroed for the next stage-2 launch
};

// stages 1 and 3 (large grids) reset the 64 chunk counters the NEXT traversal launch on this stream starts from
__device__ __forceinline__ void reset_trace_queues(const ShadeParams &p)
{
    if (blockIdx.x == 0 && threadIdx.x < NVDR_TRACE_QUEUES) p.queues[threadIdx.x * 32u] = 0u;
}

__device__ __forceinline__ unsigned launch_seed(const ShadeParams &p) { return p.seed_dev ? p.seed + *p.seed_dev : p.seed; }

// pixels of this chunk (the covered-pixel count lives on the device; chunks behind it are empty launches)
// (Written with explicit branches: the one-liner `total > begin ? min(total - begin, cap) : 0` followed by `if (P == 0) return`
// was compiled WITHOUT the guard in light_grad_band_kernel -- s_sub, s_min_u32, s_cmp_eq -- so the empty chunks of a launch
// scanned `cap` pixels of stale records: 3 x 13.4 ms per 8-view backward pass, found in the round-2 kernel trace.)
__device__ __forceinline__ unsigned chunk_span(unsigned total, unsigned begin, unsigned cap)
{
    if (total <= begin) return 0u;
    const unsigned rest = total - begin;
    return rest < cap ? rest : cap;
}
__device__ __forceinline__ unsigned chunk_pixels(const ShadeParams &p) { return chunk_span(*p.pix_count, p.pix_begin, p.pix_cap); }

// ---------------------------------------------------------------------------------------------
// work-list compaction: covered pixels (mask > 0, kernel.cu:478) in raster order per wave

// A wavefront looks at NVDR_COMPACT_ROUNDS x 64 consecutive pixels and claims list space for all of them with ONE atomic (the
// counter is a single address: with one claim per 64 pixels the 32 k claims of an 8-view launch serialised into 113 us).
#define NVDR_COMPACT_ROUNDS 16
// zero_a / zero_b (forward launches): the two output images are zero-filled HERE at the pixels that are not covered -- the shading kernel stores
// every covered pixel -- instead of by a memset of their own (one node less in front of the generation kernel of a launch-bound iteration)
__global__ void __launch_bounds__(256) compact_pixels_kernel(const float *__restrict__ mask, int64_t ms0, int64_t ms1, int64_t ms2, int N, int H,
                                                             int W, int *__restrict__ list, unsigned *count, float *__restrict__ zero_a,
                                                             float *__restrict__ zero_b)
{
    const unsigned total = (unsigned)(N * H * W);           // < 2^31 (checked by the launcher)
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned first = wave * (64u * NVDR_COMPACT_ROUNDS);
    const bool dense = ms2 == 1 && ms1 == W && ms0 == (int64_t)W * H;
    unsigned long long bits[NVDR_COMPACT_ROUNDS];
    unsigned n = 0;
#pragma unroll
    for (int k = 0; k < NVDR_COMPACT_ROUNDS; ++k) {
        const unsigned i = first + k * 64 + lane;
        bool on = false;
        if (i < total) {
            if (dense) {
                on = mask[i] > 0.0f;
            } else {
                const unsigned x = i % (unsigned)W, y = (i / (unsigned)W) % (unsigned)H, z = i / ((unsigned)W * (unsigned)H);
                on = mask[z * ms0 + y * ms1 + x * ms2] > 0.0f;
            }
        }
        bits[k] = __ballot(on);
        n += (unsigned)__popcll(bits[k]);
        if (zero_a && i < total && !on) {
            float *za = zero_a + 3 * (int64_t)i, *zb = zero_b + 3 * (int64_t)i;
            za[0] = 0.0f; za[1] = 0.0f; za[2] = 0.0f;
            zb[0] = 0.0f; zb[1] = 0.0f; zb[2] = 0.0f;
        }
    }
    if (n == 0) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(count, n);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
    for (int k = 0; k < NVDR_COMPACT_ROUNDS; ++k) {
        if ((bits[k] >> lane) & 1ull) list[base + (unsigned)__popcll(bits[k] & ((1ull << lane) - 1ull))] = (int)(first + k * 64 + lane);
        base += (unsigned)__popcll(bits[k]);
    }
}

// start of an env-shade launch: reset the covered-pixel counter (not when the forward's work list is reused) and the
// live-ray counters of the chunks -- unless the stored ray stream is going to be reused, which is decided HERE, on the
// device, because only the device knows whether the forward's covered pixels fitted one chunk
// zero-fill of up to four output images in ONE launch (the reference's torch::zeros, torch_bindings.cpp:148-149; the caller's four
// gradient tensors are separate storages: four memset launches of ~5 us each in a one-view iteration)
__global__ void __launch_bounds__(256) zero_outputs_kernel(float *b0, float *b1, float *b2, float *b3, int64_t n)
{
    float *const b = blockIdx.y == 0 ? b0 : (blockIdx.y == 1 ? b1 : (blockIdx.y == 2 ? b2 : b3));
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
        b[i] = 0.0f;
        if (i + 256 < n) b[i + 256] = 0.0f;
        if (i + 512 < n) b[i + 512] = 0.0f;
        if (i + 768 < n) b[i + 768] = 0.0f;
    }
}

__global__ void begin_launch_kernel(unsigned *pix_count, unsigned *chunk_counts, int n_chunks, int reuse, unsigned cap, unsigned seg_cap,
                                    unsigned *seed_counter, unsigned *seed_snapshot, unsigned seed_advance)
{
    // the device-resident seed counter of shade() (render.py:112-116): this launch uses the value it finds (kept in the snapshot
    // for the backward pass), the next one the advanced value
    if (seed_snapshot && threadIdx.x == 0) {
        const unsigned s = *seed_counter;
        *seed_snapshot = s;
        *seed_counter = s + seed_advance;
    }
    const bool keep = reuse && *pix_count <= cap;
    __syncthreads();
    if (!reuse && threadIdx.x == 0) *pix_count = 0;
    if (!keep)
        for (int i = threadIdx.x; i < n_chunks * (NVDR_LIVE_SEGS + 1); i += blockDim.x) {
            const int k = i / (NVDR_LIVE_SEGS + 1), w = i % (NVDR_LIVE_SEGS + 1);      // word 32 w of chunk k's block: a segment's length, or the capacity
            chunk_counts[(size_t)k * NVDR_LIVE_WORDS + 32 * w] = w == NVDR_LIVE_SEGS ? seg_cap : 0u;
        }
}

// ---------------------------------------------------------------------------------------------
// RNG (kernel.cu:30-45)

__device__ __forceinline__ unsigned rand_pcg(unsigned &s)
{
    const unsigned st = s;
    const unsigned word = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    s = st * 747796405u + 2891336453u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float uniform_pcg(unsigned &s) { return (float)(rand_pcg(s) & 0xFFFFFF) / (float)0x1000000; }
// advance the LCG by k steps in O(log k)
// ... as an affine map: k steps of s -> a*s + c are s -> am*s + ap, and (am, ap) depend on k only
__device__ __forceinline__ void lcg_skip_coeff(unsigned k, unsigned &am, unsigned &ap)
{
    am = 1u; ap = 0u;
    unsigned cm = 747796405u, cp = 2891336453u;
    while (k) {
        if (k & 1u) { am *= cm; ap = ap * cm + cp; }
        cp = (cm + 1u) * cp;
        cm *= cm;
        k >>= 1;
    }
}

// ---------------------------------------------------------------------------------------------
// sampling helpers (device transcription of kernel.cu:47-397; same order of operations as the oracle)

__device__ __forceinline__ void branchless_onb(F3 n, F3 &b1, F3 &b2)
{
    const float sign = copysignf(1.0f, n.z);
    const float a = nvdr_div(-1.0f, sign + n.z);         // n is a unit vector here: 1 <= |sign + n.z| <= 2
    const float b = n.x * n.y * a;
    b1 = f3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    b2 = f3(b, sign + n.y * n.y * a, -n.y);
}
__device__ __forceinline__ F3 tolocal(F3 a, F3 u, F3 v, F3 w) { return f3(dot3(a, u), dot3(a, v), dot3(a, w)); }
__device__ __forceinline__ F3 toworld(F3 a, F3 u, F3 v, F3 w) { return (u * a.x + v * a.y) + w * a.z; }

__device__ __forceinline__ F3 cosine_sample(F3 N, float u, float v, float &pdf)
{
    N = safe_normalize(N);
    F3 dx, dy;
    branchless_onb(N, dx, dy);
    const float phi = (float)(2.0 * NVDR_PI_DBL * (double)u);
    const float costheta = nvdr_sqrt(v);                // v = (stratum + k 2^-24) / n: zero or above 2^-27
    const float sintheta = (float)nvdr_dsqrt(1.0 - (double)v);
    float sp, cp;
    nvdr_sincosf(phi, &sp, &cp);
    const float x = cp * sintheta, y = sp * sintheta, z = costheta;
    pdf = (float)fmax((double)0.000001f, (double)costheta * NVDR_INV_PI_DBL);       // == costheta / pi (see NVDR_INV_PI_DBL)
    const F3 vec = (dx * x + dy * y) + N * z;
    return safe_normalize(vec);
}
__device__ __forceinline__ float albedo(F3 baseColor, F3 wo, F3 N)
{
    const F3 W = safe_normalize(N);
    F3 U, V;
    branchless_onb(W, U, V);
    const F3 wo_l = safe_normalize(tolocal(wo, U, V, W));
    const float cosNO = wo_l.z;
    if (!(cosNO > 0)) return 0.0f;
    return luminance(fwd_fresnel3(baseColor, f3(1.0f), cosNO));
}
__device__ __forceinline__ void dir_to_tc(F3 dir, float &u, float &v)
{
    u = (float)((double)nvdr_atan2f(dir.x, -dir.z) * NVDR_INV_2PI_DBL + 0.5);          // == atan2 / (2 pi) + 0.5 (see NVDR_INV_PI_DBL)
    v = (float)((double)nvdr_acosf(clampf(dir.y, -1.0f, 1.0f)) * NVDR_INV_PI_DBL);     // == acos / pi
}
__device__ __forceinline__ F3 tc_to_dir(float u, float v)
{
    float sinphi, cosphi, sintheta, costheta;
    nvdr_sincosf((float)((double)(u * 2.0f - 1.0f) * NVDR_PI_DBL), &sinphi, &cosphi);
    nvdr_sincosf((float)((double)v * NVDR_PI_DBL), &sintheta, &costheta);
    return f3(sintheta * sinphi, costheta, -sintheta * cosphi);
}
__device__ __forceinline__ unsigned cdf_iterations(unsigned hi)
{
    // int(ceil(log2(float(hi)))) + 1  (kernel.cu:147), in integers
    return hi >= 2u ? (unsigned)(32 - __clz((int)(hi - 1u))) + 1u : hi;
}
// CDF inversion, kernel.cu:141-166: m steps of the reference's bisection.  (Measured and dropped, session V of round 2: running 2 or
// 3 steps per round trip by loading the mids of the following steps for either outcome -- same comparisons, bit-identical, parity
// tests green -- makes the kernel 1.4 % / 10 % SLOWER: the kernel is not waiting for these loads, the extra lookups cost more.)
__device__ __forceinline__ float sample_cdf(const float *__restrict__ cdf, int stride, int size, float x, unsigned &idx)
{
    x = fminf(x, 0.99999994f);
    unsigned lo = 0, hi = (unsigned)(size - 1);
    const unsigned m = cdf_iterations(hi);
    for (unsigned i = 0; i < m; ++i) {
        const unsigned mid = (lo + hi) >> 1;
        const float c = cdf[mid * stride];
        lo = x >= c ? mid : lo;
        hi = x < c ? mid : hi;
    }
    idx = hi;
    float pdf, sample;
    if (hi == 0) {
        pdf = cdf[0];
        sample = x;
    } else {
        const float d0 = cdf[hi * stride], d1 = cdf[(hi - 1) * stride];
        pdf = d0 - d1;
        sample = x - d1;
    }
    return fminf(nvdr_div(sample, pdf), 0.99999994f);       // pdf: a step of a normalised cumulative sum -- zero (the fix-up) or far above the denormals
}
// The same inversion through a GUIDE TABLE (cdf_guide_kernel): the bisection above returns the first entry above x (or the last
// entry); guide[k] is the first entry above k / K for the cell k = floor(x * K) of x (K a power of two: the product is exact), which
// cannot lie behind it, so a short linear walk from there ends on the same entry -- two or three dependent loads instead of nine.
__device__ __forceinline__ float sample_cdf_guided(const float *__restrict__ cdf, int stride, int size, float x, const uint16_t *__restrict__ guide,
                                                   int guide_log, unsigned &idx)
{
    x = fminf(x, 0.99999994f);
    unsigned hi = guide[(unsigned)(x * (float)(1u << guide_log))];
    float d0 = cdf[hi * stride];
    while (hi + 1u < (unsigned)size && !(x < d0)) {
        ++hi;
        d0 = cdf[hi * stride];
    }
    idx = hi;
    float pdf, sample;
    if (hi == 0) {
        pdf = cdf[0];
        sample = x;
    } else {
        const float d1 = cdf[(hi - 1) * stride];
        pdf = d0 - d1;
        sample = x - d1;
    }
    return fminf(nvdr_div(sample, pdf), 0.99999994f);       // pdf: a step of a normalised cumulative sum -- zero (the fix-up) or far above the denormals
}
// one thread per (table row, cell): lower bound of cell / K in the row's CDF (rows < n_rows: the column CDF of that row; row n_rows:
// the row CDF), cells K_cols resp. K_rows wide rows of `guide` (row pitch = max of the two)
__global__ void __launch_bounds__(256) cdf_guide_kernel(Tab rows, Tab cols, int log_rows, int log_cols, uint16_t *__restrict__ guide)
{
    const int pitch = 1 << max(log_rows, log_cols);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = i / pitch, k = i - y * pitch;
    if (y > cols.n0) return;
    const bool is_rows = y == cols.n0;
    const int lg = is_rows ? log_rows : log_cols, size = is_rows ? rows.n0 : cols.n1;
    if (k >= (1 << lg)) return;
    const float *cdf = is_rows ? rows.p : cols.p + (int64_t)y * cols.s0;
    const int stride = is_rows ? rows.s0 : cols.s1;
    const float t = (float)k / (float)(1 << lg);
    int lo = 0, hi = size;                      // first entry with cdf > t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid * stride] > t) hi = mid; else lo = mid + 1;
    }
    guide[(int64_t)y * pitch + k] = (uint16_t)min(lo, size - 1);
}
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

__device__ __forceinline__ float light_pdf(const ShadeParams &p, F3 dir, int &tx, int &ty)
{
    float cu, cv;
    dir_to_tc(dir, cu, cv);
    const int Wl = p.pdf.n1, Hl = p.pdf.n0;
    const int x = clampi((int)(cu * (float)Wl), 0, Wl - 1);
    const int y = clampi((int)(cv * (float)Hl), 0, Hl - 1);
    // the radiance texel (eval_light_fwd, kernel.cu:195-199): same mapping at the light's own resolution
    tx = clampi((int)(cu * (float)p.light.n1), 0, p.light.n1 - 1);
    ty = clampi((int)(cv * (float)p.light.n0), 0, p.light.n0 - 1);
    float s, c;
    nvdr_sincosf((float)((double)cv * NVDR_PI_DBL), &s, &c);
    const float pdf_weight = (float)nvdr_ddiv((double)(Hl * Wl), 2.0 * NVDR_PI_DBL * NVDR_PI_DBL * (double)fmaxf(s, 0.0001f));
    return p.pdf.p[y * p.pdf.s0 + x * p.pdf.s1] * pdf_weight;
}
// the direction of a light sample alone (light_sample without its pdf: the generation kernel asks for the pdf of live samples only)
__device__ __forceinline__ F3 light_sample_dir(const ShadeParams &p, float u, float v)
{
    unsigned x, y;
    const int pitch = 1 << max(p.guide_log_rows, p.guide_log_cols);
    const float ry = p.cdf_guide ? sample_cdf_guided(p.rows.p, p.rows.s0, p.rows.n0, v, p.cdf_guide + (int64_t)p.cols.n0 * pitch, p.guide_log_rows, y)
                                 : sample_cdf(p.rows.p, p.rows.s0, p.rows.n0, v, y);
    const float rx = p.cdf_guide ? sample_cdf_guided(p.cols.p + (int64_t)y * p.cols.s0, p.cols.s1, p.cols.n1, u, p.cdf_guide + (int64_t)y * pitch, p.guide_log_cols, x)
                                 : sample_cdf(p.cols.p + (int64_t)y * p.cols.s0, p.cols.s1, p.cols.n1, u, x);
    return tc_to_dir(nvdr_div((float)x + rx, (float)p.pdf.n1), nvdr_div((float)y + ry, (float)p.pdf.n0));
}
__device__ __forceinline__ F3 light_sample(const ShadeParams &p, float u, float v, float &pdf, int &tx, int &ty)
{
    unsigned x, y;
    const float ry = sample_cdf(p.rows.p, p.rows.s0, p.rows.n0, v, y);
    const float rx = sample_cdf(p.cols.p + (int64_t)y * p.cols.s0, p.cols.s1, p.cols.n1, u, x);
    const F3 d = tc_to_dir(nvdr_div((float)x + rx, (float)p.pdf.n1), nvdr_div((float)y + ry, (float)p.pdf.n0));
    pdf = light_pdf(p, d, tx, ty);
    return d;
}
__device__ __forceinline__ float eval_ndf_ggx(float alpha, float cosTheta)
{
    const float a2 = alpha * alpha;
    const float d = ((cosTheta * a2 - cosTheta) * cosTheta + 1);
    return (float)nvdr_ddiv((double)a2, (double)(d * d) * NVDR_PI_DBL);
}
__device__ __forceinline__ float eval_g1_ggx(float alphaSqr, float cosTheta)
{
    if (cosTheta <= 0) return 0;
    const float c2 = cosTheta * cosTheta;
    const float tan2 = fmaxf(1.0f - c2, 0.0f) / c2;         // (plain `/`: the square of an unclamped cosine may be denormal)
    return nvdr_div(2.0f, 1 + nvdr_sqrt(1 + alphaSqr * tan2));
}
__device__ __forceinline__ float eval_pdf_ggx_vndf(float alpha, F3 wo, F3 h)
{
    const float G1 = eval_g1_ggx(alpha * alpha, wo.z);
    const float D = eval_ndf_ggx(alpha, h.z);
    return G1 * D * fmaxf(0.f, dot3(wo, h)) / wo.z;         // (plain `/` here and for `/ (4 woDotH)` below: unclamped cosines -- ieee_arith.h)
}
__device__ __forceinline__ F3 sample_ggx_vndf(float alpha, F3 wo, float ux, float uy, float &pdf)
{
    const F3 Vh = safe_normalize(f3(alpha * wo.x, alpha * wo.y, wo.z));
    const F3 T1 = (Vh.z < 0.9999f) ? safe_normalize(cross3(f3(0.f, 0.f, 1.f), Vh)) : f3(1.f, 0.f, 0.f);
    const F3 T2 = cross3(Vh, T1);
    const float r = nvdr_sqrt(ux);
    const float phi = (2.f * NVDR_PI_FLT) * uy;
    float sp, cp;
    nvdr_sincosf(phi, &sp, &cp);
    const float t1 = r * cp;
    float t2 = r * sp;
    const float s = 0.5f * (1.f + Vh.z);
    t2 = (1.f - s) * nvdr_sqrt(1.f - t1 * t1) + s * t2;        // (differences of numbers of order one: zero or above 2^-25)
    const F3 Nh = (T1 * t1 + T2 * t2) + Vh * nvdr_sqrt(fmaxf(0.f, 1.f - t1 * t1 - t2 * t2));
    const F3 h = safe_normalize(f3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.f, Nh.z)));
    pdf = eval_pdf_ggx_vndf(alpha, wo, h);
    return h;
}
__device__ __forceinline__ F3 ggx_sample(F3 N, F3 wo, float u, float v, float alpha, float &pdf)
{
    const F3 W = safe_normalize(N);
    F3 U, V;
    branchless_onb(W, U, V);
    const F3 wo_l = safe_normalize(tolocal(wo, U, V, W));
    const float cosNO = wo_l.z;
    if (!(cosNO > 0)) {
        pdf = 0.f;
        return f3(0.f);
    }
    const F3 h = sample_ggx_vndf(alpha, wo_l, u, v, pdf);
    const float woDotH = dot3(wo_l, h);
    const F3 wi_l = (h * woDotH) * 2.0f - wo_l;
    pdf /= (4.0f * woDotH);
    return safe_normalize(toworld(wi_l, U, V, W));
}
__device__ __forceinline__ float ggx_pdf(F3 N, F3 wo, F3 wi, float alpha)
{
    const F3 W = safe_normalize(N);
    F3 U, V;
    branchless_onb(W, U, V);
    const F3 wo_l = tolocal(wo, U, V, W);
    const F3 wi_l = tolocal(wi, U, V, W);
    float pdf = 0.0f;
    if (wo_l.z > 0 && wi_l.z > 0) {
        const F3 m = safe_normalize(wi_l + wo_l);
        const float woDotH = dot3(m, wo_l);
        const float D = eval_ndf_ggx(alpha, m.z);
        const float G1 = eval_g1_ggx(alpha * alpha, wo_l.z);
        pdf = G1 * D * fmaxf(0.f, dot3(wo_l, m)) / wo_l.z;
        pdf /= (4 * woDotH);
    }
    return pdf;
}
__device__ __forceinline__ void update_pdf(float &pdf, float opdf, float b)
{
    if (b > 0.000001f) {
        opdf *= b;
        pdf += opdf;
    }
}
__device__ __forceinline__ F3 bsdf_sample(float pDiffuse, float pSpecular, F3 N, F3 wo, float sx, float sy, float sz,
                                          float alpha, float &pdf)
{
    pdf = 0.0f;
    F3 wi_o;
    if (sz < pDiffuse) {
        if (pDiffuse < 0.0001f) {
            pdf = 1.0f;
            return N;
        }
        wi_o = cosine_sample(N, sx, sy, pdf);
        pdf *= pDiffuse;
        if (pSpecular > 0) update_pdf(pdf, ggx_pdf(N, wo, wi_o, alpha), 1.0f - pDiffuse);
    } else {
        wi_o = ggx_sample(N, wo, sx, sy, alpha, pdf);
        pdf *= 1.f - pDiffuse;
        if (pDiffuse > 0) update_pdf(pdf, (float)(fmax((double)dot3(N, wi_o), 0.0) * NVDR_INV_PI_DBL), pDiffuse);
    }
    return wi_o;
}
__device__ __forceinline__ float bsdf_pdf(float pDiffuse, float pSpecular, F3 N, F3 wo, F3 wi, float alpha)
{
    const float NdotL = dot3(N, wi), NdotV = dot3(N, wo);
    float pdf = 0.0f;
    if (fminf(NdotV, NdotL) < 1e-6f) return 1.0f;
    if (pDiffuse > 0) update_pdf(pdf, (float)(fmax((double)dot3(N, wi), 0.0) * NVDR_INV_PI_DBL), pDiffuse);
    if (pSpecular > 0) update_pdf(pdf, ggx_pdf(N, wo, wi, alpha), 1.0f - pDiffuse);
    return pdf;
}

// butterfly sum over the L lanes that share a pixel (L a power of two <= 64)
// (Round 6, session 37, measured and dropped: the same sums as DPP modifiers on the adds -- quad permutes, row_half_mirror, row_mirror, row_bcast15 / 31, the
// last lane writing -- instead of __shfl_xor, which this compiler turns into ds_bpermute_b32.  No LDS round trips, and 8 % faster at 16 spp -- but at the 64 spp
// of the benchmark the forward shading kernel got 3.6 % SLOWER (the compiler folds only a quarter of the permutes into their adds; the rest are v_mov_dpp + add
// with hazard nops between dependent steps, where the six values' permutes used to pipeline), and the results were not the butterfly's bit for bit.)
__device__ __forceinline__ float group_sum(float v, int L)
{
    for (int o = L >> 1; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ F3 group_sum3(F3 v, int L) { return f3(group_sum(v.x, L), group_sum(v.y, L), group_sum(v.z, L)); }

__device__ __forceinline__ F3 fetch_light_texel(const Tab &t, int texel)
{
    // texel = y * n1 + x (stage 1).  Rows without padding (s0 == n1 * s1: every probe the operators build) make the address texel * s1;
    // the integer division by the run-time width is ~25 vector instructions per SAMPLE otherwise.  (Wave-uniform test.)
    const float *q;
    if (t.s0 == t.n1 * t.s1) {
        q = t.p + (int64_t)texel * t.s1;
    } else {
        const int y = texel / t.n1, x = texel - y * t.n1;
        q = t.p + (int64_t)y * t.s0 + (int64_t)x * t.s1;
    }
    return t.n2 == 1 ? f3(q[0]) : f3(q[0], q[t.s2], q[2 * t.s2]);
}

// ---------------------------------------------------------------------------------------------
// stage 1: sample generation (kernel.cu:463-526 minus process_sample)

#define NVDR_GEN_STAGE 480u         // (with the ring's 11 KB and the queues' 20 KB: 40 448 bytes per workgroup, four of them in a CU's 160 KB)
#define NVDR_GEN_RING 64u          // pixels whose set-up the queued samples of a wavefront may still refer to
#define NVDR_GEN_QCAP 128u         // entries of one lobe queue (< 64 waiting + <= 64 pushed per round)

// one list-space claim for `staged` slots of a wavefront, then a coalesced copy out of LDS; returns the new fill (0)
// capacity of one segment of the live-ray list: generation wavefront w appends to segment w % NVDR_LIVE_SEGS, so a segment takes at most the
// slots of the groups of ceil(waves / SEGS) wavefronts (the launcher sizes the list with the same formula)
__host__ __device__ static inline unsigned long long live_segment_capacity(unsigned long long gen_waves, unsigned long long max_groups, unsigned long long group_slots)
{
    const unsigned long long v = ((gen_waves + NVDR_LIVE_SEGS - 1) / NVDR_LIVE_SEGS) * ((max_groups + gen_waves - 1) / gen_waves) * group_slots;
    return (v + 127ull) & ~127ull;
}
__device__ __forceinline__ unsigned flush_live(const unsigned *stage, unsigned staged, int lane, const ShadeParams &p, unsigned seg_cap)
{
    if (staged == 0) return 0;
    __builtin_amdgcn_wave_barrier();        // the staged entries were written by other lanes of this wavefront
    // this wavefront's segment of the list (trace_kernel.h: one claim counter per segment; one counter for all was the kernel's bound)
    const unsigned seg = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % NVDR_LIVE_SEGS;
    unsigned at = 0;
    if (lane == 0) at = atomicAdd(p.ray_count + 32u * seg, staged);
    at = (unsigned)__builtin_amdgcn_readfirstlane((int)at) + seg * seg_cap;
    for (unsigned k = lane; k < staged; k += 64) p.live[at + k] = stage[k];
    __builtin_amdgcn_wave_barrier();
    return 0;
}


// (Round 4 measured and dropped a pass between stage 1 and stage 2 that ordered every window of 2 048 entries of the live-ray list by
// the direction octant of its rays -- the "bucket the claim windows by direction" lead of round 3.  Bit-identical outputs, and the
// traversal kernel 2-11 % SLOWER on bob, 8-15 % slower on 684 k triangles (profiles/r04_ab_live_sort.md): a wavefront's 64 consecutive
// entries are the rays of ONE pixel -- one origin -- and walk the same nodes around that origin before they part; 64 rays of one
// octant from sixteen pixels do not.  The list order the generation kernel produces is the coherent one.)

// DBG: the NVDR_DEBUG experiment switches are compiled INTO a second instantiation only; the production kernels (DBG = false) carry none
template <bool DBG>
__global__ void __launch_bounds__(256, NVDR_GEN_OCC) env_gen_kernel(ShadeParams p)
{
    const unsigned dbg = DBG ? p.debug : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = p.L, G = 64 >> p.log2L;
    const int slot = lane >> p.log2L, sub = lane & (L - 1);
    reset_trace_queues(p);
    if (p.reuse && *p.pix_count <= p.pix_cap) return;     // backward pass: the forward's stream is still valid (one chunk)
    const unsigned P = chunk_pixels(p);
    const unsigned n_groups = (P + G - 1) / G;
    const unsigned waves_total = gridDim.x * (blockDim.x >> 6);
    const unsigned S = p.S, n = p.n;
    const unsigned seg_cap = live_segment_capacity(waves_total, (p.pix_cap + (unsigned)G - 1u) / (unsigned)G, (unsigned)G * 2u * S);
    const float strata_frac = 1.0f / (float)n;
    // stratum / n as a multiplication: n_magic = ceil(2^32 / n) gives the exact quotient of every stratum < 2^16 (the error term
    // stratum * (n_magic * n - 2^32) stays below 2^32); n = 1 wraps to 0, and its only stratum is 0
    const unsigned n_magic = 0xFFFFFFFFu / n + 1u;
    // Live-ray list: a device-scope atomic per wavefront and round serialises on its one address (61 k of them cost
    // 0.4 ms); each wavefront therefore stages slots in LDS and claims list space once per NVDR_GEN_STAGE-64 entries -- and the list is
    // NVDR_LIVE_SEGS segments with a counter each (flush_live; trace_kernel.h): even the 109 k staged claims of an 8-view launch were
    // 1.3 ms on one counter.
    __shared__ unsigned stage_all[4][NVDR_GEN_STAGE];
    unsigned *stage = stage_all[wave];
    unsigned staged = 0;
    // BSDF samples by lobe.  bsdf_sample (kernel.cu:330-352) picks the diffuse or the specular lobe per SAMPLE (sz < pDiffuse), so
    // the 64 strata of a pixel split between two long branches and a wavefront that samples in place executes both, each with part
    // of its lanes.  Here a lane only draws its numbers and queues a task (sx, sy, stratum, pixel) for the lobe it picked; a lobe is
    // evaluated when 64 of its tasks wait -- every lane busy, one branch -- together with everything that follows from the direction
    // (light pdf, culling, stream slot).  The same arithmetic per sample, in another order.  The pixel's set-up a task needs
    // (normal, wo, alpha, pDiffuse) waits in a small ring; before a ring entry is reused the queues are drained, full or not.
    // A third queue does the same for the LIGHT samples: half of them point under the horizon -- known as soon as the direction is --
    // and need neither the light pdf (atan2, acos, sincos, pdf lookup) nor the BSDF pdf; the live ones are queued (direction, stratum,
    // pixel) and get both 64 at a time.
    // (LDS of a workgroup: 39.5 KB, four workgroups per CU.)
    __shared__ float4 ring_a_all[4][NVDR_GEN_RING], ring_b_all[4][NVDR_GEN_RING];      // (normal, alpha), (wo, pDiffuse)
    __shared__ unsigned ring_pi_all[4][NVDR_GEN_RING];                                  // the pixel's index in the chunk
    __shared__ unsigned ring_rng_all[4][3][NVDR_GEN_RING];                              // the pixel's random state and its two permutation rows
    __shared__ float2 lobe_xy_all[4][2][NVDR_GEN_QCAP];                                 // BSDF tasks: (sx, sy) ...
    __shared__ unsigned lobe_key_all[4][2][NVDR_GEN_QCAP];                              // ... and stratum | ring entry << 16; [0] diffuse, [1] specular lobe
    __shared__ float4 lightq_all[4][NVDR_GEN_QCAP];                                     // live light samples: (direction, stratum | ring entry << 16)
    float4 *ring_a = ring_a_all[wave], *ring_b = ring_b_all[wave];
    unsigned *ring_pi = ring_pi_all[wave];
    unsigned *ring_rng[3] = {ring_rng_all[wave][0], ring_rng_all[wave][1], ring_rng_all[wave][2]};
    float2 *lobe_xy[2] = {lobe_xy_all[wave][0], lobe_xy_all[wave][1]};
    unsigned *lobe_key[2] = {lobe_key_all[wave][0], lobe_key_all[wave][1]};
    float4 *lightq = lightq_all[wave];
    unsigned q_head[3] = {0u, 0u, 0u}, q_count[3] = {0u, 0u, 0u};       // wave-uniform; [2]: the light-sample queue
    const unsigned ring_groups = NVDR_GEN_RING / (unsigned)G;           // groups of pixels the ring holds (G <= 64)
    unsigned groups_done = 0;
    // the lane's jump (5 draws per stratum, kernel.cu:513-524) is the same for every pixel: computed once when one round
    // of L lanes covers all S strata, per round otherwise
    unsigned jump_m, jump_a;
    lcg_skip_coeff(5u * (unsigned)sub, jump_m, jump_a);

    // live slots of this round / batch -> the wavefront's staging buffer (ballot ranks; `staged` is wave-uniform)
    auto stage_live = [&](bool live, unsigned r) {
        const unsigned long long m = __ballot(live);
        if (live) stage[staged + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = r;
        staged += (unsigned)__popcll(m);
        if (staged > NVDR_GEN_STAGE - 64u) staged = flush_live(stage, staged, lane, p, seg_cap);      // (room for the next append of <= 64)
    };
    // one batch of `cnt` (<= 64) tasks of one lobe (wave-uniform arguments)
    auto run_batch = [&](int lobe, unsigned cnt) {
        __builtin_amdgcn_wave_barrier();                    // tasks and ring entries were written by other lanes of this wavefront
        const bool has = (unsigned)lane < cnt;
        const unsigned qi = (q_head[lobe] + (unsigned)lane) & (NVDR_GEN_QCAP - 1u);
        float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        unsigned key;
        if (lobe == 2) {
            t = lightq[qi];
            key = __float_as_uint(t.w);
        } else {
            const float2 xy = lobe_xy[lobe][qi];
            t.x = xy.x; t.y = xy.y;
            key = lobe_key[lobe][qi];
        }
        q_head[lobe] += cnt;
        q_count[lobe] -= cnt;
        if (has) {
            const unsigned pb = key & 0xffffu, ri = key >> 16;
            const float4 s0 = ring_a[ri], s1 = ring_b[ri];
            const unsigned pix = ring_pi[ri];
            const F3 N = f3(s0.x, s0.y, s0.z), wo = f3(s1.x, s1.y, s1.z);
            const float alpha = s0.w, pDiffuse = s1.w, sx = t.x, sy = t.y;
            if (lobe == 2) {                                // a live light sample: both pdfs, the ray, the texel (kernel.cu:513-520)
                const F3 dirA = f3(t.x, t.y, t.z);
                int txA, tyA;
                const float pdfA_light = light_pdf(p, dirA, txA, tyA);
                const float pdfA_bsdf = bsdf_pdf(pDiffuse, 1.0f - pDiffuse, N, wo, dirA, alpha);
                const unsigned rA = pix * 2u * S + pb;      // (pb holds the light stratum here)
                p.rays[rA] = make_float4(dirA.x, dirA.y, dirA.z, pdfA_light + pdfA_bsdf);
                p.texel[rA] = tyA * p.light.n1 + txA;
                return;
            }
            float pdfB_bsdf = 0.0f;
            F3 dirB;
            if (lobe == 0) {                                // bsdf_sample, the branch sz < pDiffuse
                if (pDiffuse < 0.0001f) {
                    pdfB_bsdf = 1.0f;
                    dirB = N;
                } else {
                    dirB = cosine_sample(N, sx, sy, pdfB_bsdf);
                    pdfB_bsdf *= pDiffuse;
                    if (1.0f - pDiffuse > 0) update_pdf(pdfB_bsdf, ggx_pdf(N, wo, dirB, alpha), 1.0f - pDiffuse);
                }
            } else {                                        // ... and the other one
                dirB = ggx_sample(N, wo, sx, sy, alpha, pdfB_bsdf);
                pdfB_bsdf *= 1.f - pDiffuse;
                if (pDiffuse > 0) update_pdf(pdfB_bsdf, (float)(fmax((double)dot3(N, dirB), 0.0) * NVDR_INV_PI_DBL), pDiffuse);
            }
            int txB, tyB;
            const float pdfB_light = light_pdf(p, dirB, txB, tyB);
            const unsigned rB = pix * 2u * S + S + pb;
            const unsigned deadB = (!(dbg & 8u) && !(dot3(N, dirB) > 0.0f)) ? 0x80000000u : 0u;
            p.rays[rB] = make_float4(dirB.x, dirB.y, dirB.z, __uint_as_float(__float_as_uint(pdfB_light + pdfB_bsdf) | deadB));
            p.texel[rB] = tyB * p.light.n1 + txB;
        }
    };

    // The per-pixel set-up (kernel.cu:490-505: five strided fetches, wo, the lobe probabilities, the pixel's random state) is the same for
    // all lanes of a pixel, and a wavefront that ran it pixel by pixel spent a fifth of its instructions on it -- ~400 vector instructions
    // with every lane computing the same numbers, and the fetches' latency in front of every pixel.  It now runs ONE PIXEL PER LANE for
    // all the pixels of a ring cycle at once (the wavefront's next 64 >> log2(G) groups, when the ring wraps and the queues are empty)
    // and leaves what the rounds need in the ring; a round then starts with four LDS reads.  Same arithmetic per pixel, same results.
    const int log2G = 6 - p.log2L;
    for (unsigned grp = blockIdx.x * (blockDim.x >> 6) + wave; grp < n_groups; grp += waves_total) {
        const unsigned in_cycle = groups_done % ring_groups;
        if (in_cycle == 0u) {
            const unsigned long long a_grp = (unsigned long long)grp + (unsigned long long)(lane >> log2G) * waves_total;
            const unsigned a_pi = (unsigned)a_grp * (unsigned)G + ((unsigned)lane & (unsigned)(G - 1));
            const bool a_valid = a_grp < n_groups && a_pi < P;
            const int lin = p.pix_list[p.pix_begin + (a_valid ? a_pi : 0u)];
            const int x = lin % p.W, y = (lin / p.W) % p.H, z = lin / (p.W * p.H);
            const F3 pos = fetch3(p.pos, z, y, x), nrm = fetch3(p.nrm, z, y, x);
            const F3 view_pos = fetch3(p.view_pos, z, y, x), kd = fetch3(p.kd, z, y, x), ks = fetch3(p.ks, z, y, x);
            const float alpha = ks.y * ks.y;
            const F3 wo = safe_normalize(view_pos - pos);
            const float metallic = ks.z;
            const F3 specColor = f3(0.04f) * (1.0f - metallic) + kd * metallic;
            const float diffuseWeight = (1.f - metallic) * luminance(kd);
            const float specularWeight = albedo(specColor, wo, nrm);
            const float pDiffuse = (diffuseWeight + specularWeight) > 0.f ? diffuseWeight / (diffuseWeight + specularWeight) : 1.f;
            unsigned a_seed = launch_seed(p), b_seed = (unsigned)lin + p.pix_offset;
            unsigned rng0 = rand_pcg(a_seed) ^ rand_pcg(b_seed);
            const unsigned lightIdx = rand_pcg(rng0) % p.n_perms;
            const unsigned bsdfIdx = rand_pcg(rng0) % p.n_perms;
            __builtin_amdgcn_wave_barrier();                // (the batches that read the previous cycle's entries have run: the queues are empty here)
            if (a_valid) {
                const F3 ro = fetch3(p.ro, z, y, x);
                p.pix_origin[a_pi] = make_float4(ro.x, ro.y, ro.z, 0.0f);
                // what stage 3 shades this pixel's samples with, in one 64-byte record behind ONE pointer: its kernels then keep two strided
                // views (the incoming gradients) in scalar registers instead of seven (round 5: the parameter block no longer spills)
                float4 *su = p.pix_setup + 4 * (int64_t)a_pi;
                su[0] = make_float4(pos.x, pos.y, pos.z, nrm.x);
                su[1] = make_float4(nrm.y, nrm.z, view_pos.x, view_pos.y);
                su[2] = make_float4(view_pos.z, kd.x, kd.y, kd.z);
                su[3] = make_float4(ks.x, ks.y, ks.z, 0.0f);
            }
            ring_a[lane] = make_float4(nrm.x, nrm.y, nrm.z, alpha);
            ring_b[lane] = make_float4(wo.x, wo.y, wo.z, pDiffuse);
            ring_pi[lane] = a_valid ? a_pi : 0xFFFFFFFFu;
            ring_rng[0][lane] = rng0; ring_rng[1][lane] = lightIdx; ring_rng[2][lane] = bsdfIdx;
            __builtin_amdgcn_wave_barrier();
        }
        const unsigned ring_at = in_cycle * (unsigned)G + (unsigned)slot;
        const unsigned pi = ring_pi[ring_at];               // index inside the chunk
        const bool valid = pi != 0xFFFFFFFFu;
        const float4 ra = ring_a[ring_at];
        const F3 nrm = f3(ra.x, ra.y, ra.z);
        const float pDiffuse = ring_b[ring_at].w;
        const unsigned rng0 = ring_rng[0][ring_at], lightIdx = ring_rng[1][ring_at], bsdfIdx = ring_rng[2][ring_at];

        for (unsigned base = 0; base < S; base += L) {
            const unsigned i = base + sub;
            const bool has = valid && i < S;
            bool liveA = false, cosine = false;
            unsigned rA = 0;
            float4 task = make_float4(0.0f, 0.0f, 0.0f, 0.0f), taskA = task;
            if (has) {
                unsigned jm = jump_m, ja = jump_a;
                if (base != 0u) lcg_skip_coeff(5u * i, jm, ja);
                unsigned rng = jm * rng0 + ja;
                // light importance sample (kernel.cu:513-516)
                const unsigned pl = (unsigned)p.perms[(int64_t)lightIdx * p.perm_s0 + (int64_t)i * p.perm_s1];
                const unsigned pl_y = __umulhi(pl, n_magic), pl_x = pl - pl_y * n;        // pl / n, pl % n
                float sx = ((float)pl_x + uniform_pcg(rng)) * strata_frac;
                float sy = ((float)pl_y + uniform_pcg(rng)) * strata_frac;
                const F3 dirA = light_sample_dir(p, sx, sy);
                // BSDF importance sample (kernel.cu:522-526): the numbers are drawn here, the lobe is sampled in run_batch
                const unsigned pb = (unsigned)p.perms[(int64_t)bsdfIdx * p.perm_s0 + (int64_t)i * p.perm_s1];
                const unsigned pb_y = __umulhi(pb, n_magic), pb_x = pb - pb_y * n;
                sx = ((float)pb_x + uniform_pcg(rng)) * strata_frac;
                sy = ((float)pb_y + uniform_pcg(rng)) * strata_frac;
                const float sz = uniform_pcg(rng);
                cosine = sz < pDiffuse;
                task = make_float4(sx, sy, __uint_as_float(pb | (ring_at << 16)), 0.0f);
                // Stream order inside a pixel: the S light-sampled rays by THEIR STRATUM (pl), then the S BSDF-sampled rays by
                // theirs (pb).  The permutation tables scramble which sample draws which stratum; ordering by stratum puts
                // neighbouring cells of the CDF / hemisphere grid -- i.e. nearby directions -- into neighbouring lanes of the
                // traversal kernel, which keeps its wavefronts coherent (both rows are permutations of 0..S-1: a bijection).
                // (The other grouping -- one stratum of 64 neighbouring pixels per wave, i.e. near-parallel rays from spread-out
                // origins -- was measured 3-12 % slower than this one: shared origins matter more than shared directions.)
                rA = pi * 2u * S + pl;
                // Dead samples: with dot(n, wi) <= 0 the Lambert term is max(.,0) = 0 and the GGX lobe fails its front-facing
                // gate (bsdf.h:121,165 -- the same dot product), forward AND backward, so the sample contributes exactly
                // zero whatever its visibility.  Such rays (about half of the light-sampled ones: the probe covers the whole
                // sphere) are flagged in the sign bit of the pdf sum (never negative) for stage 3, which skips them, and are
                // left out of the list of stream slots stage 2 traverses (appended per wavefront; the list order varies from
                // run to run, the visibility of a slot does not).
                // NVDR_DEBUG bit 8 switches the culling off (traces every ray like the reference).
                liveA = (dbg & 8u) || dot3(nrm, dirA) > 0.0f;
                // a dead sample's slot only says so (nobody reads its pdfs or its texel); a live one gets them in run_batch
                if (!liveA) p.rays[rA] = make_float4(dirA.x, dirA.y, dirA.z, __uint_as_float(0x80000000u));
                taskA = make_float4(dirA.x, dirA.y, dirA.z, __uint_as_float(pl | (ring_at << 16)));
            }
            stage_live(liveA, rA);
            {
                const unsigned long long m2 = __ballot(liveA);
                if (liveA) lightq[(q_head[2] + q_count[2] + __builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0u))) & (NVDR_GEN_QCAP - 1u)] = taskA;
                q_count[2] += (unsigned)__popcll(m2);
                if (q_count[2] >= 64u) run_batch(2, 64u);
            }
            // The BSDF-sampled ray goes on the list of rays to traverse NOW, in the order of the round, whether it turns out dead or
            // not: its direction is not known yet, dead BSDF samples are 0.3 % of the rays (the lobes point away from the surface), and
            // listing them by batch instead -- a pixel's rays in three places of the list -- costs the traversal kernel 1.8 %.
            stage_live(has, pi * 2u * S + S + (__float_as_uint(task.z) & 0xffffu));
            // queue the BSDF tasks by lobe, run the lobes that have a full wavefront of them
            {
                const unsigned long long m0 = __ballot(has && cosine), m1 = __ballot(has && !cosine);
                const unsigned long long mine = cosine ? m0 : m1;
                if (has) {
                    const unsigned at = (cosine ? q_head[0] + q_count[0] : q_head[1] + q_count[1]) +
                                        __builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0u));
                    (cosine ? lobe_xy[0] : lobe_xy[1])[at & (NVDR_GEN_QCAP - 1u)] = make_float2(task.x, task.y);
                    (cosine ? lobe_key[0] : lobe_key[1])[at & (NVDR_GEN_QCAP - 1u)] = __float_as_uint(task.z);
                }
                q_count[0] += (unsigned)__popcll(m0);
                q_count[1] += (unsigned)__popcll(m1);
                if (q_count[0] >= 64u) run_batch(0, 64u);
                if (q_count[1] >= 64u) run_batch(1, 64u);
            }
        }
        // the ring is about to wrap: nothing may wait on the entries the next groups overwrite
        if (++groups_done % ring_groups == 0u) {
            if (q_count[0]) run_batch(0, q_count[0]);
            if (q_count[1]) run_batch(1, q_count[1]);
            if (q_count[2]) run_batch(2, q_count[2]);
        }
    }
    if (q_count[0]) run_batch(0, q_count[0]);
    if (q_count[1]) run_batch(1, q_count[1]);
    if (q_count[2]) run_batch(2, q_count[2]);
    flush_live(stage, staged, lane, p, seg_cap);
}

// ---------------------------------------------------------------------------------------------
// stage 2: persistent-wavefront any-hit traversal of the ray stream -- trace_kernel.h

template <bool COUNT>
__global__ void __launch_bounds__(NVDR_QUERY_BLOCK, NVDR_TRACE_OCC) env_trace_kernel(TraceLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    env_trace_body<COUNT, 0, false, COUNT>(a, smem);      // (the counting build knows the treetop table, if the context has one)
}

// the build with the treetop table in LDS (bvh.h; contexts with NVDR_TRACE_TOP_NODES set)
__global__ void __launch_bounds__(NVDR_QUERY_BLOCK, NVDR_TRACE_OCC) env_trace_top_kernel(TraceLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    env_trace_body<false, 0, false, true>(a, smem);
}

// the build with split walks in the drain (trace_kernel.h SPLIT): small launches
__global__ void __launch_bounds__(NVDR_QUERY_BLOCK, NVDR_TRACE_OCC) env_trace_split_kernel(TraceLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    env_trace_body<false, 0, true>(a, smem);
}

// the phase-clock builds of the same kernel (trace_kernel.h PH; counting launches only)
template <int PH>
__global__ void __launch_bounds__(NVDR_QUERY_BLOCK, NVDR_TRACE_OCC) env_trace_phase_kernel(TraceLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    env_trace_body<false, PH>(a, smem);
}

// ---------------------------------------------------------------------------------------------
// stage 3: shading (process_sample, kernel.cu:403-461) forward or backward

// backward: the incoming gradients of the chunk's covered pixels, gathered through their strided views ONCE into two float4 per compacted
// pixel; the shading kernels then hold no strided view at all in their scalar registers (a 5-us launch per chunk)
__global__ void __launch_bounds__(256) pack_grads_kernel(ShadeParams p)
{
    const unsigned P = chunk_pixels(p);
    for (unsigned pi = blockIdx.x * blockDim.x + threadIdx.x; pi < P; pi += gridDim.x * blockDim.x) {
        const int lin = p.pix_list[p.pix_begin + pi];
        const int x = lin % p.W, y = (lin / p.W) % p.H, z = lin / (p.W * p.H);
        const F3 d = fetch3(p.dgrad, z, y, x), s = fetch3(p.sgrad, z, y, x);
        p.pix_grad[2 * (int64_t)pi] = make_float4(d.x, d.y, d.z, 0.0f);
        p.pix_grad[2 * (int64_t)pi + 1] = make_float4(s.x, s.y, s.z, 0.0f);
    }
}

// The G-buffer values of one pixel that every sample of it is shaded with (, and the incoming gradients of the backward pass)
struct PixelSetup {
    F3 pos, nrm, view_pos, kd, ks, dgrad, sgrad;
};
__device__ __forceinline__ void load_grads(const ShadeParams &p, unsigned pi, F3 &dgrad, F3 &sgrad)
{
    const float4 a = p.pix_grad[2 * (int64_t)pi], b = p.pix_grad[2 * (int64_t)pi + 1];
    dgrad = f3(a.x, a.y, a.z); sgrad = f3(b.x, b.y, b.z);
}
// the record stage 1 packed for compacted pixel pi (env_gen_kernel)
__device__ __forceinline__ void load_setup(const ShadeParams &p, unsigned pi, F3 &pos, F3 &nrm, F3 &view_pos, F3 &kd, F3 &ks)
{
    const float4 *su = p.pix_setup + 4 * (int64_t)pi;
    const float4 a = su[0], b = su[1], c = su[2], d = su[3];
    pos = f3(a.x, a.y, a.z); nrm = f3(a.w, b.x, b.y); view_pos = f3(b.z, b.w, c.x); kd = f3(c.y, c.z, c.w); ks = f3(d.x, d.y, d.z);
}

// One sample of process_sample (kernel.cu:403-461) -- ray direction and pdf sum `rd`, light texel, visibility -- shaded for pixel `px`.
// Forward: its contribution is ADDED to (diff, spec).  Backward: its gradient terms are ADDED to (g_kd, g_ks, g_pos, g_nrm) in the order
// the reference's adjoint code forms them, and its light-gradient addend (eval_light_bwd, kernel.cu:203-211) is returned in lg.
// The three stage-3 kernels differ in which lane shades which sample and where the sums live, not in this arithmetic.
template <bool BACKWARD>
__device__ __forceinline__ void shade_sample(const ShadeParams &p, const PixelSetup &px, const float4 &rd, int texel, bool occluded, float sample_frac,
                                             F3 &diff, F3 &spec, F3 &g_kd, F3 &g_ks, F3 &g_pos, F3 &g_nrm, F3 &lg)
{
    const F3 dir = f3(rd.x, rd.y, rd.z);
    const float pdfSum = rd.w;
    const F3 light_col = fetch_light_texel(p.light, texel);
    // (float)(1.0 / (double)f) of the reference == the IEEE float quotient 1.0f / f: rounding a double quotient of two floats
    // to float is innocuous double rounding (53 >= 2 * 24 + 2 bits)
    const float mis_weight = nvdr_div(1.0f, fmaxf(pdfSum, 0.0001f));
    F3 _diff = f3(0.0f), _spec = f3(0.0f);
    if (p.bsdf == 1 || p.bsdf == 2)
        _diff = f3(fwd_lambert(px.nrm, dir));
    else
        fwd_pbr_bsdf_shader(px.kd, px.ks, px.pos, px.nrm, px.view_pos, dir, 0.08f, _diff, _spec);
    const float vis = occluded ? 0.0f : 1.0f;
    const float V = vis * p.shadow_scale + (1 - p.shadow_scale);
    if (BACKWARD) {
        lg = (((px.dgrad * _diff + px.sgrad * _spec) * V) * mis_weight) * sample_frac;
        const F3 _dg = (((px.dgrad * light_col) * V) * mis_weight) * sample_frac;
        const F3 _sg = (((px.sgrad * light_col) * V) * mis_weight) * sample_frac;
        if (p.bsdf == 1 || p.bsdf == 2) {
            F3 d_wi = f3(0.0f);
            bwd_lambert(px.nrm, dir, g_nrm, d_wi, sum3(_dg));
        } else {
            bwd_pbr_bsdf_shader(px.kd, px.ks, px.pos, px.nrm, px.view_pos, dir, 0.08f, g_kd, g_ks, g_pos, g_nrm, _dg, _sg);
        }
    } else {
        diff += (((_diff * light_col) * V) * mis_weight) * sample_frac;
        spec += (((_spec * light_col) * V) * mis_weight) * sample_frac;
    }
}

// Light-gradient records of ONE WAVEFRONT of a backward shading kernel, sorted by band as they are written (light_grad_block_kernel reads
// them).  A record = (rgb addend, texel); records of one band are packed into BLOCKS of 128 stream slots, and a block is tagged with
// (band, fill) once it is complete.  Where the blocks come from: the ray stream itself.  The rays of a group of pixels are dead once the
// wavefront has read them, and a group emits at most as many records as it has slots, so the wavefront carves blocks out of the range
// it has already consumed (free_ptr .. free_end); until enough is consumed -- the first group, the up to n_bands blocks that are open at
// any time, the alignment of the range to 128 slots -- it draws on lg_spw spare blocks of its own behind the chunk.  Per-band state
// lives in the lanes of two registers (lane b: next slot to write / slots left in the open block of band b).
#ifndef NVDR_LG_EMIT_ROUNDS
#define NVDR_LG_EMIT_ROUNDS 0           // 1: rounds 3-5's placement, one round per band present among a pass pair's records (A/B)
#endif
#define NVDR_LG_NO_BLOCK 0xFFFFFF80u    // bpos of a band without an open block (aligned: "no slot left")
struct RecordBlocks {
    // bpos: lane b holds the next slot to write in the open block of band b.  The slots left in that block are (0 - bpos) & 127: a block ends at a
    // multiple of 128, and an aligned bpos means "full" (the next record opens a new block).
    unsigned free_ptr, free_end, spare_next, bpos;
#if NVDR_LG_EMIT_ROUNDS
    unsigned bleft;
#endif

    // first_slot: the first stream slot this wavefront is going to consume; spare_first: its first spare block
    __device__ __forceinline__ void init(unsigned first_slot, unsigned spare_first)
    {
        free_ptr = (first_slot + 127u) & ~127u;
        free_end = first_slot;
        spare_next = spare_first;
#if NVDR_LG_EMIT_ROUNDS
        bpos = 0xFFFFFFFFu;
        bleft = 0u;
#else
        bpos = NVDR_LG_NO_BLOCK;
#endif
    }
#if NVDR_LG_EMIT_ROUNDS
    // both record sets of a pass pair at once (A first: the light-sampled ones): one placement round per band present in either.
    // Called in converged control flow.
    __device__ __forceinline__ void emit(const ShadeParams &p, int lane, bool hasA, const float4 &recA, bool hasB, const float4 &recB, unsigned *)
    {
        const int bandA = hasA ? (__float_as_int(recA.w) >> p.lg_shift) : -1;
        const int bandB = hasB ? (__float_as_int(recB.w) >> p.lg_shift) : -1;
        unsigned long long remA = __ballot(hasA), remB = __ballot(hasB);
        while (remA | remB) {
            // the band of the first record still to place
            const int b = remA ? __builtin_amdgcn_readlane(bandA, __builtin_ctzll(remA)) : __builtin_amdgcn_readlane(bandB, __builtin_ctzll(remB));
            const unsigned long long mA = __ballot(bandA == b), mB = __ballot(bandB == b);
            remA &= ~mA;
            remB &= ~mB;
            const unsigned nA = (unsigned)__popcll(mA), cnt = nA + (unsigned)__popcll(mB);
            const unsigned at_slot = (unsigned)__builtin_amdgcn_readlane((int)bpos, b), left = (unsigned)__builtin_amdgcn_readlane((int)bleft, b);
            unsigned fresh = 0u;
            if (cnt > left) {                                           // the open block fills up: `left` records complete it, the rest open a new one
                unsigned blk;
                if (free_ptr + 128u <= free_end) { blk = free_ptr >> 7; free_ptr += 128u; }
                else blk = spare_next++;
                fresh = blk << 7;
                if (at_slot != 0xFFFFFFFFu && lane == 0) p.lg_tags[(at_slot + left - 1u) >> 7] = (uint16_t)((unsigned)b | (128u << 8));
            }
            const unsigned to_fresh = fresh - left;                     // (wraps; only used by ranks >= left)
            if (bandA == b) {
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, 0u));
                p.rays[rank + (rank < left ? at_slot : to_fresh)] = recA;
            }
            if (bandB == b) {
                const unsigned rank = nA + __builtin_amdgcn_mbcnt_hi((unsigned)(mB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mB, 0u));
                p.rays[rank + (rank < left ? at_slot : to_fresh)] = recB;
            }
            const unsigned at2 = cnt > left ? fresh + (cnt - left) : at_slot + cnt;
            const unsigned left2 = cnt > left ? 128u - (cnt - left) : left - cnt;
            bpos = lane == b ? at2 : bpos;
            bleft = lane == b ? left2 : bleft;
        }
    }
    // the blocks still open when the wavefront runs out of pixels: lane b tags the one of band b with its fill
    __device__ __forceinline__ void finish(const ShadeParams &p, int lane) const
    {
        if (bpos != 0xFFFFFFFFu) p.lg_tags[(bpos + bleft - 1u) >> 7] = (uint16_t)((unsigned)lane | ((128u - bleft) << 8));
    }
#else
    // Both record sets of a pass pair at once, in a FIXED number of steps (round 6; rounds 3-5 took one placement round of ~40 instructions per band
    // present among the records, five or six of the eight on the benchmark probe -- a tenth of the backward shading kernel's instructions):
    //   1. a record's rank inside its band = what an LDS counter of that band returns (the B records behind the A records: the light-sampled first);
    //   2. lane b reads band b's count and does the block bookkeeping of that band in its own registers: does the open block overflow, and if so
    //      which fresh block follows (the bands that need one are ranked by a ballot: consumed stream range first, then the spare blocks);
    //   3. every record fetches its band's (slot, fresh block) from lane `band` with two ds_bpermute and is stored.
    // cnt: 16 LDS words of this wavefront, all zero between calls.  Called in converged control flow.
    __device__ __forceinline__ void emit(const ShadeParams &p, int lane, bool hasA, const float4 &recA, bool hasB, const float4 &recB, unsigned *cnt)
    {
        const int bandA = hasA ? (__float_as_int(recA.w) >> p.lg_shift) : 0;
        const int bandB = hasB ? (__float_as_int(recB.w) >> p.lg_shift) : 0;
        unsigned rankA = 0u, rankB = 0u;
        if (hasA) rankA = __hip_atomic_fetch_add(cnt + bandA, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();                    // (LDS operations of a wavefront execute in order: the B ranks continue the A ranks)
        if (hasB) rankB = __hip_atomic_fetch_add(cnt + bandB, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        unsigned c = 0u;
        if (lane < 16) {
            c = __hip_atomic_load(cnt + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(cnt + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_wave_barrier();
        // lane b: band b
        const unsigned at_slot = bpos, left = (0u - bpos) & 127u;
        const bool need = c > left;                           // the open block fills up: `left` records complete it, the rest open a new one (c = 0 for lanes >= 16)
        const unsigned long long nm = __ballot(need);
        unsigned fresh = 0u;
        if (nm != 0ull) {
            const unsigned k = __builtin_amdgcn_mbcnt_hi((unsigned)(nm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm, 0u));
            const unsigned n_need = (unsigned)__popcll(nm);
            const unsigned avail = free_end > free_ptr ? (free_end - free_ptr) >> 7 : 0u;      // whole blocks of the consumed stream range
            const unsigned from_free = min(n_need, avail);
            const unsigned blk = k < from_free ? (free_ptr >> 7) + k : spare_next + (k - from_free);
            free_ptr += from_free << 7;
            spare_next += n_need - from_free;
            if (need) {
                fresh = blk << 7;
                if (at_slot != NVDR_LG_NO_BLOCK) p.lg_tags[(at_slot + left - 1u) >> 7] = (uint16_t)((unsigned)lane | (128u << 8));
            }
        }
        if (c != 0u) bpos = need ? fresh + (c - left) : at_slot + c;
        // the records: their band's slot and fresh block from lane `band` (all lanes execute the permutes: a source lane must be active to be read)
        const unsigned atA = (unsigned)__builtin_amdgcn_ds_bpermute(bandA << 2, (int)at_slot), frA = (unsigned)__builtin_amdgcn_ds_bpermute(bandA << 2, (int)fresh);
        const unsigned atB = (unsigned)__builtin_amdgcn_ds_bpermute(bandB << 2, (int)at_slot), frB = (unsigned)__builtin_amdgcn_ds_bpermute(bandB << 2, (int)fresh);
        if (hasA) {
            const unsigned l = (0u - atA) & 127u;
            p.rays[rankA + (rankA < l ? atA : frA - l)] = recA;
        }
        if (hasB) {
            const unsigned l = (0u - atB) & 127u;
            p.rays[rankB + (rankB < l ? atB : frB - l)] = recB;
        }
    }
    // the blocks still open when the wavefront runs out of pixels: lane b tags the one of band b with its fill
    __device__ __forceinline__ void finish(const ShadeParams &p, int lane) const
    {
        const unsigned left = (0u - bpos) & 127u;
        if (bpos != NVDR_LG_NO_BLOCK) p.lg_tags[(bpos + left - 1u) >> 7] = (uint16_t)((unsigned)lane | ((128u - left) << 8));
    }
#endif
};

// Tried in round 3 and dropped (session 29): fetching the next pixel's list entry one iteration ahead, and starting both samples'
// texel -> radiance load chains before the arithmetic: forward +1 ... +2.6 %, backward +5 ... +8 % (the registers they hold cost more than
// the latency they hide at 3 waves per SIMD).
// Tried on the backward instantiation and dropped (no gain, same GPU session): issuing the light-gradient atomics after both
// samples / 3-4 waves per SIMD with spills (launch bounds) / fast division + contraction + fp32 islands (-23 % VALU
// instructions, but gradient errors of 1e-2) -- all 0.87-0.93 ms.  0.37 ms of it is the 8 M float atomics: every one leaves
// the XCD as a 64-byte write (rocprofv3 WRITE_SIZE = 506 MB per launch), i.e. it is executed at the memory side whatever the
// scope bits say; non-temporal loads of the ray stream (to keep the accumulators in L2) change nothing either.
template <bool BACKWARD, bool DBG>
__global__ void __launch_bounds__(256, BACKWARD ? NVDR_SHADE_OCC : 1) env_shade_kernel(ShadeParams p)
{
    const unsigned dbg = DBG ? p.debug : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = p.L, G = 64 >> p.log2L;
    const int slot = lane >> p.log2L, sub = lane & (L - 1);
    reset_trace_queues(p);
    const unsigned P = chunk_pixels(p);
    const unsigned n_groups = (P + G - 1) / G;
    const unsigned waves_total = gridDim.x * (blockDim.x >> 6);
    const unsigned S = p.S, n = p.n;
    const float sample_frac = 1.0f / (float)(n * n);
    // HW_REG_XCC_ID (hwreg 20), bits [3:0]: the XCD this wave runs on
    const unsigned xcc = BACKWARD ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : 0u;
    float *xcd_light = (BACKWARD && !p.lg_records) ? p.g_light_xcd + (int64_t)xcc * p.light_elems : nullptr;
    const bool use_bits = BACKWARD && p.vis_cache != nullptr;    // replay the caller's cached forward bits
    const bool save_bits = !BACKWARD && p.vis_cache != nullptr;

    // Work split.  Forward: groups of pixels dealt round-robin.  Backward: every wavefront takes ONE CONTIGUOUS run of groups, so
    // that the stream slots it has consumed form one growing range -- the space its light-gradient records go to (below).
    const unsigned wave_id = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + wave));
    unsigned grp_first = wave_id, grp_last = n_groups, grp_step = waves_total;
    if (BACKWARD && NVDR_LG_EXPERIMENT != 2 && NVDR_LG_EXPERIMENT != 3) {
        const unsigned per_wave = (n_groups + waves_total - 1) / waves_total;
        grp_first = min(wave_id * per_wave, n_groups);
        grp_last = min(grp_first + per_wave, n_groups);
        grp_step = 1;
    }
    // light-gradient records of this wavefront (RecordBlocks above)
    const unsigned gs = (unsigned)G * 2u * S;                           // stream slots of one group
    // per-band record counters of this wavefront (RecordBlocks::emit): zero between calls
    __shared__ unsigned lg_cnt_all[4][16];
    unsigned *const lg_cnt = lg_cnt_all[wave];
    if (BACKWARD && lane < 16) lg_cnt[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    RecordBlocks rb;
    rb.init(grp_first * gs, p.lg_spare_base + wave_id * p.lg_spw);

    for (unsigned grp = grp_first; grp < grp_last; grp += grp_step) {
        const unsigned pi = grp * G + slot;
        const bool valid = pi < P;
        const int lin = p.pix_list[p.pix_begin + (valid ? pi : 0u)];
        F3 pos, nrm, view_pos, kd, ks;
        load_setup(p, valid ? pi : 0u, pos, nrm, view_pos, kd, ks);
        F3 dgrad = f3(0.0f), sgrad = f3(0.0f);
        if (BACKWARD) load_grads(p, valid ? pi : 0u, dgrad, sgrad);
        const PixelSetup px = {pos, nrm, view_pos, kd, ks, dgrad, sgrad};
        F3 diffAccum = f3(0.0f), specAccum = f3(0.0f);
        F3 g_pos = f3(0.0f), g_nrm = f3(0.0f), g_kd = f3(0.0f), g_ks = f3(0.0f);
        // Stage 1 stored the pixel's samples BY STRATUM (slot s = the light-sampled ray of stratum s, slot S + s the
        // BSDF-sampled one): shading only sums over them, so this stage walks the slots in order -- no permutation rows, no
        // RNG, two dependent loads less per sample than walking them by sample index as round 1 did.  Bit s of a cached
        // visibility plane is therefore the ray of stratum s, for the forward pass that writes it and the backward that replays it.
        for (unsigned base = 0; base < S; base += L) {
            const unsigned i = base + sub;                  // stratum
            const bool active = valid && i < S;
            const unsigned ii = active ? i : 0;
            const int64_t rbase = (int64_t)(valid ? pi : 0) * 2 * S;
            const int64_t rA = rbase + ii, rB = rbase + S + ii;
            // the two rays of this stratum; a set sign bit on the pdf sum marks a dead sample (stage 1)
            // (Streaming / non-temporal loads here -- to keep the half-written record lines of the backward pass in L2 until the next
            // round completes them -- change nothing: 3.219 vs 3.229 ms backward shading + gather per 8-view launch, session 15.)
            const float4 rdA = p.rays[rA], rdB = p.rays[rB];
            const unsigned dead = active ? ((__float_as_uint(rdA.w) >> 31) | ((__float_as_uint(rdB.w) >> 31) << 1)) : 3u;
            unsigned occ = 0;
            if (use_bits) {
                const uint32_t *vc = p.vis_cache + (int64_t)lin * 2 * p.vis_words;
                if (active) {
                    occ |= ((vc[i >> 5] >> (i & 31u)) & 1u);
                    occ |= ((vc[p.vis_words + (i >> 5)] >> (i & 31u)) & 1u) << 1;
                }
            } else {
                if (!(dead & 1u)) occ |= p.vis[rA] ? 0u : 1u;
                if (!(dead & 2u)) occ |= p.vis[rB] ? 0u : 2u;
            }
            if (save_bits) {
                const unsigned long long ba = __ballot(occ & 1u), bb = __ballot((occ >> 1) & 1u);
                if (sub == 0 && valid) {
                    uint32_t *vc = p.vis_cache + (int64_t)lin * 2 * p.vis_words;
                    if (L == 64) {
                        const int w0 = 2 * (int)(base >> 6);
                        vc[w0] = (uint32_t)ba;
                        vc[p.vis_words + w0] = (uint32_t)bb;
                        if (w0 + 1 < p.vis_words) {
                            vc[w0 + 1] = (uint32_t)(ba >> 32);
                            vc[p.vis_words + w0 + 1] = (uint32_t)(bb >> 32);
                        }
                    } else {
                        const unsigned long long m = (1ull << L) - 1ull;
                        vc[0] = (uint32_t)((ba >> (slot * L)) & m);
                        vc[p.vis_words] = (uint32_t)((bb >> (slot * L)) & m);
                    }
                }
            }
            // (no `continue` for lanes without a sample: the record placement below votes across the whole wavefront)
            // light-gradient addend of one sample, atomics mode (the fallback): three global atomics
            auto emit_light_grad = [&](F3 lg, int at) {
                float *g = xcd_light + (int64_t)at * 3;
                __hip_atomic_fetch_add(g + 0, lg.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(g + 1, lg.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(g + 2, lg.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            float4 lg_recA = make_float4(0.0f, 0.0f, 0.0f, 0.0f), lg_recB = lg_recA;
            bool lg_hasA = false, lg_hasB = false;
#if NVDR_BWD_ROLL
#pragma unroll 1
#else
#pragma unroll
#endif
            for (int r = 0; r < 2; ++r) {
              if (!((dead >> r) & 1u)) {                // a dead sample contributes exactly zero to every output
                const int64_t ri = r == 0 ? rA : rB;
                const float4 rd = r == 0 ? rdA : rdB;
                const int texel = p.texel[ri];
                F3 lg = f3(0.0f);
                shade_sample<BACKWARD>(p, px, rd, texel, (occ >> r) & 1u, sample_frac, diffAccum, specAccum, g_kd, g_ks, g_pos, g_nrm, lg);
                if (BACKWARD) {
                    // light gradient (eval_light_bwd, kernel.cu:203-211).  fp32 atomics are executed at the memory side on
                    // MI355X (rocprofv3: one 64-byte write leaves the XCD per atomic, whatever the scope bits say), ~25 G
                    // of them per second.  Each XCD accumulates into ITS OWN copy (picked by the XCC id the wave actually
                    // runs on, so the result does not depend on workgroup placement) and a tiny kernel sums the 8 copies:
                    // that spreads hot texels over 8 addresses (measured: a few per cent); what really helped was not
                    // issuing the zero addends.
                    const int at = (dbg & 4u) ? (int)((ri * 2654435761u) % (unsigned)(p.light_elems / 3)) : texel;   // bit 4: contention experiment
                    // adding +-0 never changes an accumulator that started at +0: occluded samples leave no addend
                    const bool lg_has = !(dbg & 2u) && (lg.x != 0.0f || lg.y != 0.0f || lg.z != 0.0f);
                    if (r == 0) { lg_hasA = lg_has; lg_recA = make_float4(lg.x, lg.y, lg.z, __int_as_float(at)); }
                    else { lg_hasB = lg_has; lg_recB = make_float4(lg.x, lg.y, lg.z, __int_as_float(at)); }
                    if (lg_has && !p.lg_records) emit_light_grad(lg, at);
                }
              }
            }
            // No atomic leaves the workgroup in records mode (21 M addends per 8-view launch were 63 M memory-side fp32 atomics):
            // the addends of this round go to the band blocks, light-sampled ones first, in lane order
#if NVDR_LG_EXPERIMENT == 0
            if (BACKWARD && p.lg_records) rb.emit(p, lane, lg_hasA, lg_recA, lg_hasB, lg_recB, lg_cnt);
#elif NVDR_LG_EXPERIMENT == 3           // records written in place, unsorted (what the shading kernel cost before the band blocks)
            if (BACKWARD && p.lg_records) {
                if (lg_hasA) p.rays[rA] = lg_recA;
                if (lg_hasB) p.rays[rB] = lg_recB;
            }
#endif
        }
        if (BACKWARD) rb.free_end = (grp + 1u) * gs;    // the rays of this group have all been read: its slots may hold records now

        if (!BACKWARD) {
            diffAccum = group_sum3(diffAccum, L);
            specAccum = group_sum3(specAccum, L);
            if (valid && sub == 0) {
                float *o = p.diff + (int64_t)lin * 3;
                o[0] = diffAccum.x; o[1] = diffAccum.y; o[2] = diffAccum.z;
                o = p.spec + (int64_t)lin * 3;
                o[0] = specAccum.x; o[1] = specAccum.y; o[2] = specAccum.z;
            }
        } else {
            g_pos = group_sum3(g_pos, L);
            g_nrm = group_sum3(g_nrm, L);
            g_kd = group_sum3(g_kd, L);
            g_ks = group_sum3(g_ks, L);
            if (valid && sub == 0) {
                float *o = p.g_pos + (int64_t)lin * 3;
                o[0] = g_pos.x; o[1] = g_pos.y; o[2] = g_pos.z;
                o = p.g_nrm + (int64_t)lin * 3;
                o[0] = g_nrm.x; o[1] = g_nrm.y; o[2] = g_nrm.z;
                o = p.g_kd + (int64_t)lin * 3;
                o[0] = g_kd.x; o[1] = g_kd.y; o[2] = g_kd.z;
                o = p.g_ks + (int64_t)lin * 3;
                o[0] = g_ks.x; o[1] = g_ks.y; o[2] = g_ks.z;
            }
        }
    }
    if (BACKWARD && p.lg_records) rb.finish(p, lane);
}

// ---------------------------------------------------------------------------------------------
// stage 3 for S == 64 (one pixel per wavefront round, lane = stratum): the LIGHT-sampled rays go through a queue.
//
// A light sample under the pixel's horizon is dead (stage 1 marks it; it contributes exactly zero) -- 45 % of them on bob -- while
// nearly every BSDF sample is live, so the kernel above shades a pixel in one pass with ~35 and one with ~63 busy lanes.  Here the
// BSDF samples are shaded in place; the live light samples of consecutive pixels are queued in LDS (ray, texel, visibility, stratum,
// pixel) and shaded 64 at a time, every lane busy: 1.55 instead of 2 passes per pixel.  A queued sample's contribution is ADDED TO ITS
// OWN STRATUM'S CELL of the pixel's result row in LDS, which already holds the BSDF sample's; when all samples of a pixel are done the
// row is summed over the lanes exactly as above.  Forward: per lane (0 + A) + B = A + B, the same operands -- bit-identical images.
// Backward: each sample's gradient terms are accumulated from zero before they are added to the cell (the kernel above adds them
// term by term to the running sums): last-bit differences, and like above no dependence on which pixels share a wavefront or a chunk.
// The ring holds NVDR_SQ_RING pixels (set-up + result row); a pixel whose entry is needed again while samples of it still wait
// drains the queue with a partial batch (3 entries: 0.552 passes per pixel for the light samples against 0.547 with no limit).
// (Backward, round 4, in-process A/B: two entries = 38 KB of LDS: +4.1 % -- more partial batches; two entries AND 128 VGPRs (seven dwords
// spilled) for four workgroups per CU: +1.8 % at eight views, +7.6 % at one.  Three entries at three waves per SIMD stay.)
#ifndef NVDR_SQ_RING
#define NVDR_SQ_RING 3u
#endif
#ifndef NVDR_SQ_RING_FWD
#define NVDR_SQ_RING_FWD 2u        // ring entries of the forward instantiation: 25 instead of 31 KB of LDS per workgroup, six resident per CU (forward shading 0.818 -> 0.800 ms per 8-view launch; 3: A/B)
#endif
#define NVDR_SQ_QCAP 128u
#ifndef NVDR_SQ_ROW_ROT
#define NVDR_SQ_ROW_ROT 1          // ... lane 4 c + q takes quarter (q + c) % 4 of row c (0: quarter q; A/B)
#endif
#ifndef NVDR_SQ_ROW_SUMS
#define NVDR_SQ_ROW_SUMS 1         // backward: a pixel's twelve result rows summed by four lanes each from LDS (0: twelve 64-lane butterflies, A/B)
#endif

template <bool BACKWARD, bool DBG>
__global__ void __launch_bounds__(256, BACKWARD ? NVDR_SHADE_OCC : 1) env_shade_queue_kernel(ShadeParams p)
{
    const unsigned dbg = DBG ? p.debug : 0u;
    constexpr int NF = BACKWARD ? 12 : 6;       // floats of a result cell: (diff, spec) or (g_kd, g_ks, g_pos, g_nrm)
    constexpr int NS = BACKWARD ? 21 : 15;      // floats of a pixel's set-up: pos, nrm, view_pos, kd, ks (, dgrad, sgrad)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    reset_trace_queues(p);
    const unsigned P = chunk_pixels(p);         // S == 64: a group is one pixel
    const unsigned waves_total = gridDim.x * (blockDim.x >> 6);
    const unsigned S = 64u;
    const float sample_frac = 1.0f / (float)(p.n * p.n);
    const bool use_bits = BACKWARD && p.vis_cache != nullptr;
    const bool save_bits = !BACKWARD && p.vis_cache != nullptr;

    constexpr unsigned RING = BACKWARD ? NVDR_SQ_RING : NVDR_SQ_RING_FWD;
    __shared__ __attribute__((aligned(16))) float res_all[4][RING][NF][64];
    __shared__ float setup_all[4][RING][NS + 3];
    __shared__ float4 q_rd_all[4][NVDR_SQ_QCAP];
    __shared__ unsigned q_meta_all[4][NVDR_SQ_QCAP];    // stratum | ring entry << 6 | occluded << 8
    __shared__ int q_tex_all[4][NVDR_SQ_QCAP];
    float (*res)[NF][64] = res_all[wave];
    float (*setup)[NS + 3] = setup_all[wave];
    float4 *q_rd = q_rd_all[wave];
    unsigned *q_meta = q_meta_all[wave];
    int *q_tex = q_tex_all[wave];

    // work split as in env_shade_kernel: forward round-robin, backward one contiguous run per wavefront (its consumed slots take the records)
    const unsigned wave_id = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + wave));
    unsigned grp = wave_id, grp_last = P, grp_step = waves_total;
    if (BACKWARD) {
        const unsigned per_wave = (P + waves_total - 1) / waves_total;
        grp = min(wave_id * per_wave, P);
        grp_last = min(grp + per_wave, P);
        grp_step = 1;
    }
    // light-gradient records of this wavefront (RecordBlocks above)
    const unsigned gs = 2u * S;
    // per-band record counters of this wavefront (RecordBlocks::emit): zero between calls
    __shared__ unsigned lg_cnt_all[4][16];
    unsigned *const lg_cnt = lg_cnt_all[wave];
    if (BACKWARD && lane < 16) lg_cnt[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    RecordBlocks rb;
    rb.init(grp * gs, p.lg_spare_base + wave_id * p.lg_spw);

    // ring state (wave-uniform): pixel ordinal `it` of this wavefront lives in entry it % RING until it is finalised (in order)
    unsigned it = 0, fin = 0;
    unsigned pend0 = 0, pend1 = 0, pend2 = 0;       // queued samples of the entry's pixel that are not shaded yet
    int lin0 = 0, lin1 = 0, lin2 = 0;               // its index in the frame(s)
    unsigned q_head = 0, q_count = 0;
    static_assert(RING == 2u || RING == 3u, "the ring state is spelled out for up to three entries");

    while (true) {
        const bool ring_full = it - fin == RING;
        const bool do_home = grp < grp_last && !ring_full;
        if (!do_home && q_count == 0u && fin == it) break;
        float4 lg_rec0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), lg_rec1 = lg_rec0;    // [0]: the in-place pass (BSDF samples), [1]: the queue pass
        bool lg_has0 = false, lg_has1 = false;
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {
            // what this lane shades in this pass
            bool has = false;
            unsigned ent = 0, stratum = (unsigned)lane;
            float4 rd = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int texel = 0;
            bool occluded = false;
            F3 pos, nrm, view_pos, kd, ks, dgrad = f3(0.0f), sgrad = f3(0.0f);     // the set-up of the sample's pixel
            if (k == 0) {
                if (!do_home) continue;
                ent = it % RING;
                const unsigned pi = grp;
                const int lin = p.pix_list[p.pix_begin + pi];
                // the pixel's set-up: in registers for this pass (every lane reads the same addresses), in LDS for the lanes that will
                // shade its queued samples
                load_setup(p, pi, pos, nrm, view_pos, kd, ks);
                if (BACKWARD) load_grads(p, pi, dgrad, sgrad);
                if (lane == 0) {
                    float *su = setup[ent];
                    su[0] = pos.x; su[1] = pos.y; su[2] = pos.z; su[3] = nrm.x; su[4] = nrm.y; su[5] = nrm.z;
                    su[6] = view_pos.x; su[7] = view_pos.y; su[8] = view_pos.z; su[9] = kd.x; su[10] = kd.y; su[11] = kd.z;
                    su[12] = ks.x; su[13] = ks.y; su[14] = ks.z;
                    if (BACKWARD) { su[15] = dgrad.x; su[16] = dgrad.y; su[17] = dgrad.z; su[18] = sgrad.x; su[19] = sgrad.y; su[20] = sgrad.z; }
                }
                const int64_t rA = (int64_t)pi * 2 * S + lane, rB = rA + S;
                const float4 rdA = p.rays[rA], rdB = p.rays[rB];
                const unsigned dead = (__float_as_uint(rdA.w) >> 31) | ((__float_as_uint(rdB.w) >> 31) << 1);
                unsigned occ = 0;
                if (use_bits) {
                    const uint32_t *vc = p.vis_cache + (int64_t)lin * 2 * p.vis_words;
                    occ |= ((vc[lane >> 5] >> (lane & 31)) & 1u);
                    occ |= ((vc[p.vis_words + (lane >> 5)] >> (lane & 31)) & 1u) << 1;
                } else {
                    if (!(dead & 1u)) occ |= p.vis[rA] ? 0u : 1u;
                    if (!(dead & 2u)) occ |= p.vis[rB] ? 0u : 2u;
                }
                if (save_bits) {
                    const unsigned long long ba = __ballot(occ & 1u), bb = __ballot((occ >> 1) & 1u);
                    if (lane == 0) {
                        uint32_t *vc = p.vis_cache + (int64_t)lin * 2 * p.vis_words;
                        vc[0] = (uint32_t)ba;
                        vc[p.vis_words] = (uint32_t)bb;
                        if (1 < p.vis_words) {
                            vc[1] = (uint32_t)(ba >> 32);
                            vc[p.vis_words + 1] = (uint32_t)(bb >> 32);
                        }
                    }
                }
                // the live light samples wait in the queue
                const bool liveA = !(dead & 1u);
                const unsigned long long mq = __ballot(liveA);
                if (liveA) {
                    const unsigned at = (q_head + q_count + __builtin_amdgcn_mbcnt_hi((unsigned)(mq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mq, 0u))) & (NVDR_SQ_QCAP - 1u);
                    q_rd[at] = rdA;
                    q_meta[at] = (unsigned)lane | (ent << 6) | ((occ & 1u) << 8);
                    q_tex[at] = p.texel[rA];
                }
                const unsigned pushed = (unsigned)__popcll(mq);
                q_count += pushed;
                if (ent == 0u) { pend0 = pushed; lin0 = lin; } else if (ent == 1u) { pend1 = pushed; lin1 = lin; } else { pend2 = pushed; lin2 = lin; }
                ++it;
                if (BACKWARD) rb.free_end = (grp + 1u) * gs;    // the rays of this pixel have all been read: its slots may hold records now
                grp += grp_step;
                // ... and the BSDF sample of this lane's stratum is shaded in place
                has = !(dead & 2u);
                rd = rdB;
                occluded = (occ >> 1) & 1u;
                if (has) texel = p.texel[rB];
            } else {
                // a full batch while pixels keep coming; whatever waits when the ring is full or the pixels have run out
                const unsigned cnt = q_count >= 64u ? 64u : (do_home ? 0u : q_count);
                if (cnt == 0u) continue;
                __builtin_amdgcn_wave_barrier();    // (entries pushed by other lanes in the pass before)
                has = (unsigned)lane < cnt;
                const unsigned qi = (q_head + (unsigned)lane) & (NVDR_SQ_QCAP - 1u);
                unsigned meta = 0;
                if (has) {
                    rd = q_rd[qi];
                    meta = q_meta[qi];
                    texel = q_tex[qi];
                }
                stratum = meta & 63u;
                ent = (meta >> 6) & 3u;
                occluded = (meta >> 8) & 1u;
                q_head += cnt;
                q_count -= cnt;
                pend0 -= (unsigned)__popcll(__ballot(has && ent == 0u));
                pend1 -= (unsigned)__popcll(__ballot(has && ent == 1u));
                pend2 -= (unsigned)__popcll(__ballot(has && ent == 2u));
                const float *su = setup[ent];
                pos = f3(su[0], su[1], su[2]); nrm = f3(su[3], su[4], su[5]); view_pos = f3(su[6], su[7], su[8]);
                kd = f3(su[9], su[10], su[11]); ks = f3(su[12], su[13], su[14]);
                if (BACKWARD) { dgrad = f3(su[15], su[16], su[17]); sgrad = f3(su[18], su[19], su[20]); }
            }
            float out[NF];
#pragma unroll
            for (int c = 0; c < NF; ++c) out[c] = 0.0f;
            float4 lg_rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            bool lg_has = false;
            if (has) {
                // (a sample's gradient terms are accumulated from zero before they are added to its cell)
                const PixelSetup px = {pos, nrm, view_pos, kd, ks, dgrad, sgrad};
                F3 d = f3(0.0f), sp = f3(0.0f), g_kd = f3(0.0f), g_ks = f3(0.0f), g_pos = f3(0.0f), g_nrm = f3(0.0f), lg = f3(0.0f);
                shade_sample<BACKWARD>(p, px, rd, texel, occluded, sample_frac, d, sp, g_kd, g_ks, g_pos, g_nrm, lg);
                if constexpr (BACKWARD) {
                    lg_has = !(dbg & 2u) && (lg.x != 0.0f || lg.y != 0.0f || lg.z != 0.0f);
                    lg_rec = make_float4(lg.x, lg.y, lg.z, __int_as_float(texel));
                    out[0] = g_kd.x; out[1] = g_kd.y; out[2] = g_kd.z; out[3] = g_ks.x; out[4] = g_ks.y; out[5] = g_ks.z;
                    out[6] = g_pos.x; out[7] = g_pos.y; out[8] = g_pos.z; out[9] = g_nrm.x; out[10] = g_nrm.y; out[11] = g_nrm.z;
                } else {
                    out[0] = d.x; out[1] = d.y; out[2] = d.z; out[3] = sp.x; out[4] = sp.y; out[5] = sp.z;
                }
            }
            // into the result row: the in-place pass initialises every cell of its pixel, the queue pass adds to the cell of its stratum
            if (k == 0) {
#pragma unroll
                for (int c = 0; c < NF; ++c) res[ent][c][lane] = out[c];
                lg_rec0 = lg_rec;
                lg_has0 = lg_has;
            } else {
                __builtin_amdgcn_wave_barrier();    // (cells initialised by other lanes in an in-place pass)
                if (has) {
#pragma unroll
                    for (int c = 0; c < NF; ++c) res[ent][c][stratum] = out[c] + res[ent][c][stratum];
                }
                lg_rec1 = lg_rec;
                lg_has1 = lg_has;
            }
        }
        if (BACKWARD && p.lg_records) rb.emit(p, lane, lg_has1, lg_rec1, lg_has0, lg_rec0, lg_cnt);     // light-sampled records first, as above
        // pixels whose samples are all shaded, in order: the row is summed over the lanes and written
        while (fin < it) {
            const unsigned e = fin % RING;
            const unsigned pend = e == 0u ? pend0 : (e == 1u ? pend1 : pend2);
            if (pend != 0u) break;
            const int lin = e == 0u ? lin0 : (e == 1u ? lin1 : lin2);
            __builtin_amdgcn_wave_barrier();
            if constexpr (BACKWARD && NVDR_SQ_ROW_SUMS) {
                // The twelve row sums by 48 lanes (round 6): lane 4 c + q adds sixteen cells of row c -- four 16-byte LDS reads, fifteen adds in a fixed
                // order; quarter (q + c) % 4, so that the rows spread over the LDS banks -- and the four quarters of a row meet through two quad
                // permutes: ~35 instructions where twelve 64-lane butterflies (72 ds_bpermute + their adds) were ~200, a tenth of the kernel.
                // Another fixed order than the butterfly: gradients agree with the plain kernel's to rounding.  (The FORWARD instantiation keeps the
                // butterfly: with six rows the same change made it 19 % SLOWER -- session 34 --, and its images equal the plain kernel's bit for bit.)
                const int c = lane >> 2, q = lane & 3;
                float sum = 0.0f;
                if (lane < 4 * NF) {
                    const float4 *row = (const float4 *)&res[e][c][16 * ((q + NVDR_SQ_ROW_ROT * c) & 3)];
                    const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
                    sum = (((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w))) + (((r2.x + r2.y) + (r2.z + r2.w)) + ((r3.x + r3.y) + (r3.z + r3.w)));
                }
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0xB1, 0xf, 0xf, true));       // quad_perm [1, 0, 3, 2]
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x4E, 0xf, 0xf, true));       // quad_perm [2, 3, 0, 1]
                if (lane < 4 * NF && q == 0) {
                    float *o = c < 3 ? p.g_kd : (c < 6 ? p.g_ks : (c < 9 ? p.g_pos : p.g_nrm));
                    o[(int64_t)lin * 3 + (c - 3 * (c / 3))] = sum;
                }
            } else {
                float v[NF];
#pragma unroll
                for (int c = 0; c < NF; ++c) v[c] = group_sum(res[e][c][lane], 64);
                if (lane == 0) {
                    if constexpr (BACKWARD) {
                        float *o = p.g_kd + (int64_t)lin * 3;
                        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
                        o = p.g_ks + (int64_t)lin * 3;
                        o[0] = v[3]; o[1] = v[4]; o[2] = v[5];
                        o = p.g_pos + (int64_t)lin * 3;
                        o[0] = v[6]; o[1] = v[7]; o[2] = v[8];
                        o = p.g_nrm + (int64_t)lin * 3;
                        o[0] = v[9]; o[1] = v[10]; o[2] = v[11];
                    } else {
                        float *o = p.diff + (int64_t)lin * 3;
                        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
                        o = p.spec + (int64_t)lin * 3;
                        o[0] = v[3]; o[1] = v[4]; o[2] = v[5];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();        // (the entry is free for the next pixel once its cells have been read)
            ++fin;
        }
    }
    if (BACKWARD && p.lg_records) rb.finish(p, lane);
}

// ---------------------------------------------------------------------------------------------
// stage 3 for S > 64 (several rounds of 64 strata per pixel; n_samples_x = 16: four): a PIXEL-LOCAL queue for the light samples.
//
// env_shade_kernel shades a round in two passes -- the light-sampled rays with the 55-75 % of the lanes whose sample is above the
// horizon, the BSDF-sampled ones with nearly all.  Here the BSDF samples are shaded in place and the live light samples of the pixel's
// rounds go through a queue in LDS (ray, texel, visibility: 20 bytes) that is shaded 64 at a time: ceil(live / 64) passes instead of one
// per round, 7 instead of 8 for the typical pixel at 256 spp.  Every lane keeps its sums in registers across all passes of the pixel,
// exactly as env_shade_kernel does (round 3 measured a variant that routed every sample's result to a row per round in LDS: slower
// than the plain kernel); the queue never crosses a pixel (a pixel's last round drains it), so what a pixel's lanes add up -- and in
// which order -- depends on that pixel alone, not on its neighbours, the chunking or the rank.  It is NOT the order of
// env_shade_kernel: a queued sample is shaded by the lane of its queue position, not of its stratum, so the per-lane partial sums
// differ and the images agree to rounding (1e-6 relative: the tolerance both kernels have against the oracle), not bit for bit.
// Light-gradient records: one placement pass per (queue batch, in-place pass) pair, as in the kernels above.
#define NVDR_SL_QCAP 128u

template <bool BACKWARD, bool DBG>
__global__ void __launch_bounds__(256, BACKWARD ? NVDR_SHADE_OCC : 1) env_shade_local_kernel(ShadeParams p)
{
    const unsigned dbg = DBG ? p.debug : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    reset_trace_queues(p);
    const unsigned P = chunk_pixels(p);         // L == 64: a group is one pixel
    const unsigned waves_total = gridDim.x * (blockDim.x >> 6);
    const unsigned S = p.S;
    const float sample_frac = 1.0f / (float)(p.n * p.n);
    const bool use_bits = BACKWARD && p.vis_cache != nullptr;
    const bool save_bits = !BACKWARD && p.vis_cache != nullptr;

    __shared__ float4 q_rd_all[4][NVDR_SL_QCAP];
    __shared__ int q_tex_all[4][NVDR_SL_QCAP];          // texel | occluded << 31
    float4 *q_rd = q_rd_all[wave];
    int *q_tex = q_tex_all[wave];

    // work split and light-gradient records: see env_shade_kernel
    const unsigned wave_id = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + wave));
    unsigned grp_first = wave_id, grp_last = P, grp_step = waves_total;
    if (BACKWARD) {
        const unsigned per_wave = (P + waves_total - 1) / waves_total;
        grp_first = min(wave_id * per_wave, P);
        grp_last = min(grp_first + per_wave, P);
        grp_step = 1;
    }
    const unsigned gs = 2u * S;
    // per-band record counters of this wavefront (RecordBlocks::emit): zero between calls
    __shared__ unsigned lg_cnt_all[4][16];
    unsigned *const lg_cnt = lg_cnt_all[wave];
    if (BACKWARD && lane < 16) lg_cnt[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    RecordBlocks rb;
    rb.init(grp_first * gs, p.lg_spare_base + wave_id * p.lg_spw);

    for (unsigned grp = grp_first; grp < grp_last; grp += grp_step) {
        const unsigned pi = grp;
        const int lin = p.pix_list[p.pix_begin + pi];
        F3 pos, nrm, view_pos, kd, ks;
        load_setup(p, pi, pos, nrm, view_pos, kd, ks);
        F3 dgrad = f3(0.0f), sgrad = f3(0.0f);
        if (BACKWARD) load_grads(p, pi, dgrad, sgrad);
        const PixelSetup px = {pos, nrm, view_pos, kd, ks, dgrad, sgrad};
        F3 diffAccum = f3(0.0f), specAccum = f3(0.0f);
        F3 g_pos = f3(0.0f), g_nrm = f3(0.0f), g_kd = f3(0.0f), g_ks = f3(0.0f);
        unsigned q_head = 0u, q_count = 0u;             // wave-uniform; the queue is empty between pixels
        const int64_t rbase = (int64_t)pi * 2 * S;
        for (unsigned base = 0; base < S; base += 64u) {
            const unsigned i = base + (unsigned)lane;   // stratum
            const bool active = i < S;
            const unsigned ii = active ? i : 0u;
            const int64_t rA = rbase + ii, rB = rbase + S + ii;
            const float4 rdA = p.rays[rA], rdB = p.rays[rB];
            const unsigned dead = active ? ((__float_as_uint(rdA.w) >> 31) | ((__float_as_uint(rdB.w) >> 31) << 1)) : 3u;
            unsigned occ = 0;
            if (use_bits) {
                const uint32_t *vc = p.vis_cache + (int64_t)lin * 2 * p.vis_words;
                if (active) {
                    occ |= ((vc[i >> 5] >> (i & 31u)) & 1u);
                    occ |= ((vc[p.vis_words + (i >> 5)] >> (i & 31u)) & 1u) << 1;
                }
            } else {
                if (!(dead & 1u)) occ |= p.vis[rA] ? 0u : 1u;
                if (!(dead & 2u)) occ |= p.vis[rB] ? 0u : 2u;
            }
            if (save_bits) {
                const unsigned long long ba = __ballot(occ & 1u), bb = __ballot((occ >> 1) & 1u);
                if (lane == 0) {
                    uint32_t *vc = p.vis_cache + (int64_t)lin * 2 * p.vis_words;
                    const int w0 = 2 * (int)(base >> 6);
                    vc[w0] = (uint32_t)ba;
                    vc[p.vis_words + w0] = (uint32_t)bb;
                    if (w0 + 1 < p.vis_words) {
                        vc[w0 + 1] = (uint32_t)(ba >> 32);
                        vc[p.vis_words + w0 + 1] = (uint32_t)(bb >> 32);
                    }
                }
            }
            // the live light samples of this round wait in the queue
            const bool liveA = !(dead & 1u);
            const unsigned long long mq = __ballot(liveA);
            if (liveA) {
                const unsigned at = (q_head + q_count + __builtin_amdgcn_mbcnt_hi((unsigned)(mq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mq, 0u))) & (NVDR_SL_QCAP - 1u);
                q_rd[at] = rdA;
                q_tex[at] = p.texel[rA] | (int)((occ & 1u) << 31);
            }
            q_count += (unsigned)__popcll(mq);
            const bool last = base + 64u >= S;
            float4 lg_rec0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            bool lg_has0 = false;
            // pass 0: the BSDF samples of this round in place; passes 1..: full batches of the queue (whatever is left in the last round)
#pragma unroll 1
            for (int k = 0; ; ++k) {
                bool has, occluded;
                float4 rd = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                int texel = 0;
                if (k == 0) {
                    has = !(dead & 2u);
                    rd = rdB;
                    occluded = (occ >> 1) & 1u;
                    if (has) texel = p.texel[rB];
                } else {
                    const unsigned cnt = q_count >= 64u ? 64u : (last ? q_count : 0u);
                    if (cnt == 0u) break;
                    __builtin_amdgcn_wave_barrier();    // (entries pushed by other lanes)
                    has = (unsigned)lane < cnt;
                    const unsigned qi = (q_head + (unsigned)lane) & (NVDR_SL_QCAP - 1u);
                    int tq = 0;
                    if (has) {
                        rd = q_rd[qi];
                        tq = q_tex[qi];
                    }
                    texel = tq & 0x7fffffff;
                    occluded = tq < 0;
                    q_head += cnt;
                    q_count -= cnt;
                }
                float4 lg_rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                bool lg_has = false;
                if (has) {
                    F3 lg = f3(0.0f);
                    shade_sample<BACKWARD>(p, px, rd, texel, occluded, sample_frac, diffAccum, specAccum, g_kd, g_ks, g_pos, g_nrm, lg);
                    if (BACKWARD) {
                        lg_has = !(dbg & 2u) && (lg.x != 0.0f || lg.y != 0.0f || lg.z != 0.0f);
                        lg_rec = make_float4(lg.x, lg.y, lg.z, __int_as_float(texel));
                    }
                }
                if (BACKWARD) {
                    if (k == 0) {
                        lg_rec0 = lg_rec;
                        lg_has0 = lg_has;
                    } else {
                        rb.emit(p, lane, lg_has, lg_rec, lg_has0, lg_rec0, lg_cnt);     // light-sampled records first, as above
                        lg_has0 = false;
                    }
                }
            }
            if (BACKWARD) rb.emit(p, lane, false, lg_rec0, lg_has0, lg_rec0, lg_cnt);   // (nothing left to place if a queue batch took them along)
        }
        if (BACKWARD) rb.free_end = (grp + 1u) * gs;    // the rays of this pixel have all been read: its slots may hold records now

        if (!BACKWARD) {
            diffAccum = group_sum3(diffAccum, 64);
            specAccum = group_sum3(specAccum, 64);
            if (lane == 0) {
                float *o = p.diff + (int64_t)lin * 3;
                o[0] = diffAccum.x; o[1] = diffAccum.y; o[2] = diffAccum.z;
                o = p.spec + (int64_t)lin * 3;
                o[0] = specAccum.x; o[1] = specAccum.y; o[2] = specAccum.z;
            }
        } else {
            g_pos = group_sum3(g_pos, 64);
            g_nrm = group_sum3(g_nrm, 64);
            g_kd = group_sum3(g_kd, 64);
            g_ks = group_sum3(g_ks, 64);
            if (lane == 0) {
                float *o = p.g_pos + (int64_t)lin * 3;
                o[0] = g_pos.x; o[1] = g_pos.y; o[2] = g_pos.z;
                o = p.g_nrm + (int64_t)lin * 3;
                o[0] = g_nrm.x; o[1] = g_nrm.y; o[2] = g_nrm.z;
                o = p.g_kd + (int64_t)lin * 3;
                o[0] = g_kd.x; o[1] = g_kd.y; o[2] = g_kd.z;
                o = p.g_ks + (int64_t)lin * 3;
                o[0] = g_ks.x; o[1] = g_ks.y; o[2] = g_ks.z;
            }
        }
    }
    if (BACKWARD) rb.finish(p, lane);
}

// ---------------------------------------------------------------------------------------------
// light gradient (eval_light_bwd, kernel.cu:203-211) without global atomics: BAND-SORTED RECORD BLOCKS + LDS GATHER.
//
// The reference adds every sample's addend to light_grad[texel] with three atomicAdds.  On MI355X fp32 atomics are
// executed at the memory side (rocprofv3, round 1: one 64-byte fabric write per atomic, 63 M of them per 8-view launch,
// WRITE_SIZE 3.75 GB against 0.5 GB of algorithmic addends) -- a third of the backward shading kernel's time.  Here:
//   * the probe is cut into `n_bands` bands of consecutive texels whose fp32 accumulators (band_texels * 12 B) fit the LDS
//     of one workgroup (96 KB -> 8 bands at 256x256);
//   * the backward shading kernel writes a 16-byte RECORD (rgb, texel) per non-zero addend, and it writes the records SORTED BY
//     BAND: every wavefront keeps one open block of 128 stream slots per band (carved out of the part of the ray stream it has
//     already consumed) and tags a block with (band, fill) when it is complete -- RecordBlocks::emit;
//   * workgroup (g, band) of this kernel walks the tag array (2 B per 128 slots), reads the blocks of its band as plain
//     coalesced 2 KB loads -- every lane a record, every fetched byte used -- and adds them into LDS with ds_add_f32 (hot sun
//     texels serialise inside the LDS atomic unit, not on the fabric); it leaves the tags it consumed reset to 0xFFFF;
//   * it then writes its band as ONE plain partial row; light_grad_reduce_kernel sums the partial rows.
// History (8-view launch, 58 M slots, 21 M records): round 2 kept the records in place, one per slot, and every band scanned a 4-byte
// key per slot: 1.31 ms.  Round 3 first shrank the keys to one byte and batched the fetches: still 1.26 ms -- 0.70 ms of it were
// record fetches at cache-line granularity (a line held records of ~2.5 bands and was fetched by each of them: 2.3 GB for 0.34 GB
// of records) and 0.55 ms LDS atomics issued from loops with 40 % of the lanes active (profiles/r03_gather_elimination.md).
// Sorting at the source removes both: 0.34 GB of block reads, all lanes busy.

#define NVDR_LG_THREADS 1024
#define NVDR_LG_PER_BAND_MAX_SLOTS (320ll << 20)    // launches of up to this many stream slots (ten views of 512^2 x 64 spp) deal the CUs to the bands.  Round 4, fp32 accumulators: -13 % / -6 % / -2 % / +-0 of the backward shading + gather time at 1 / 2 / 4 / 8 views of 512^2 (then: 192 Mi slots).  Round 6, with the fp64 build these launches get: -2.3 % at eight views of 512^2 (256 Mi slots), +1.3 % at eight of 800^2 (625 Mi) -- session 27
#ifndef NVDR_LG_SKIP
#define NVDR_LG_SKIP 0                   // timing-only A/B (wrong results): bit 0 no LDS adds, bit 1 no record loads, bit 2 no partial row written, bit 3 no zeroing of the accumulators
#endif
// (Round 6, measured and dropped, profiles/r06_ab_gather_batched_adds.md: all compare-and-swaps of a step issued as one batch, verdicts afterwards -- the
// gather 72 -> 119 us at one view, 275 -> 466 at eight: a swap whose word was read a batch earlier fails far more often on the sampler's hot texels.)
#ifndef NVDR_LG_NB
#define NVDR_LG_NB 4                     // blocks of 128 records a wavefront of the gather fetches together (1: A/B)
#endif
#ifndef NVDR_LG_NATIVE_ATOMICS
#define NVDR_LG_NATIVE_ATOMICS 0        // 1: ds_add_f32 (A/B only)
#endif

// F64 (round 6; the build of SMALL launches -- those whose workgroups are dealt to the bands; NVDR_LG_F64 = 0 / 1: never / wherever the bands allow): the
// accumulators are DOUBLES and a record is three ds_add_f64 -- no read, no loop, no retry.  gfx950 runs ds_add_f64 at nine times
// the rate of ds_add_f32 and, on colliding addresses (the sampler's hot texels), at three times the rate of the compare-and-swap loop
// (tools/ubench/lds_atomic.hip, profiles/r06_lds_atomic_ubench.txt); a band then holds half as many texels, so the launcher takes this build only
// where the probe still fits 16 bands (the benchmark's 256 x 256 does), the fp32 build otherwise.  Sums of fp32 terms in fp64 are rounded once, when
// the partial row is written: closer to the exact sum than the fp32 accumulation, and as independent of the order of the adds as one gets.
// Measured (profiles/r06_ab_gather_f64.md): the gather 67 -> 45 us at one view, nothing at eight (sixteen passes per workgroup instead of eight).  While the
// backward shading kernel still placed its records in one round per band (session 24) twice the bands cost it more than that; with the fixed-step placement
// (RecordBlocks::emit) they cost nothing: backward shading + gather -6.2 % at one view of bob, -2.8 % on 684 k triangles, +1.1 % at eight views (session 27).
template <bool F64>
__global__ void __launch_bounds__(NVDR_LG_THREADS) light_grad_block_kernel(uint16_t *__restrict__ tags, const float4 *__restrict__ recs,
                                                                           const unsigned *__restrict__ pix_count, unsigned pix_begin, unsigned pix_cap,
                                                                           unsigned pixels_per_group, unsigned group_slots, unsigned spare_base,
                                                                           unsigned spare_blocks, int band_texels, int n_bands, int n_texels,
                                                                           float *__restrict__ partials)
{
    extern __shared__ __attribute__((aligned(16))) float lg_acc[];
    double *const lg_acc64 = (double *)lg_acc;
    const unsigned Ptot = *pix_count;
    if (Ptot <= pix_begin) return;                          // empty chunk (light_grad_reduce_kernel makes the same test)
    const unsigned P = chunk_span(Ptot, pix_begin, pix_cap);
    // Work split: workgroup g owns the g-th SLICE of the block list and walks it once per band (the tags of a slice are a few KB).
    // Every band costs every workgroup the same, however unevenly the records are spread over the bands -- with one set of
    // workgroups per band (the first version) the equatorial bands of a lat-long probe kept their 32 CUs busy four times longer than
    // the polar ones kept theirs.
    // (Small launches -- gridDim.y = number of bands: workgroup (g, band) walks the g-th of gridDim.x slices for ITS band only; the
    // eight passes of a workgroup, each with its own zeroing, tag scan, record fetch and 96 KB row, are ~10 us apiece whatever the
    // number of records: 90 us of a one-view iteration.)
    const int g = blockIdx.x, G = gridDim.x;
    const int band_first = gridDim.y > 1 ? (int)blockIdx.y : 0, band_last = gridDim.y > 1 ? (int)blockIdx.y + 1 : n_bands;
    // blocks that may hold records: the chunk's own slots (whole groups of pixels) and the wavefronts' spare blocks behind them
    const unsigned n_groups = (P + pixels_per_group - 1u) / pixels_per_group;
    const unsigned n_own = (n_groups * group_slots + 127u) >> 7;
    const unsigned n_all = n_own + spare_blocks;
    const unsigned per = (n_all + (unsigned)G - 1u) / (unsigned)G;
    const unsigned v_lo = min((unsigned)g * per, n_all), v_hi = min(v_lo + per, n_all);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const float4 none = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    for (int band = band_first; band < band_last; ++band) {
        const int t_lo = band * band_texels, t_hi = min(t_lo + band_texels, n_texels);
        const int n_acc = (t_hi - t_lo) * 3;
        if (!(NVDR_LG_SKIP & 8)) for (int i = threadIdx.x; i < n_acc; i += NVDR_LG_THREADS) { if (F64) lg_acc64[i] = 0.0; else lg_acc[i] = 0.0f; }
        __syncthreads();
        // fp32 add to LDS.  NOT ds_add_f32: gfx950 executes that at 0.8 lane-operations per CU and nanosecond, twenty times slower
        // than its integer LDS atomics (tools/ubench/lds_atomic.hip, profiles/r03_lds_atomic_ubench.txt) -- it was 0.55 ms of the
        // 1.26 ms of this gather.  A compare-and-swap loop on the bit pattern runs at the integer rate; a lane repeats only when
        // another lane hit the same word in between.
        auto add = [&](const float4 &r) {
            const int t = __float_as_int(r.w);
            if (F64) {
                if (t >= t_lo && t < t_hi) {
                    double *a = lg_acc64 + (t - t_lo) * 3;
                    __hip_atomic_fetch_add(a + 0, (double)r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(a + 1, (double)r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(a + 2, (double)r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else if (t >= t_lo && t < t_hi) {
#if NVDR_LG_NATIVE_ATOMICS
                float *a = lg_acc + (t - t_lo) * 3;
                __hip_atomic_fetch_add(a + 0, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(a + 1, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(a + 2, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
                unsigned *a = (unsigned *)lg_acc + (t - t_lo) * 3;
                unsigned o0 = __hip_atomic_load(a + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unsigned o1 = __hip_atomic_load(a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unsigned o2 = __hip_atomic_load(a + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                bool d0 = false, d1 = false, d2 = false;
                do {
                    if (!d0) d0 = __hip_atomic_compare_exchange_strong(a + 0, &o0, __float_as_uint(__uint_as_float(o0) + r.x), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!d1) d1 = __hip_atomic_compare_exchange_strong(a + 1, &o1, __float_as_uint(__uint_as_float(o1) + r.y), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!d2) d2 = __hip_atomic_compare_exchange_strong(a + 2, &o2, __float_as_uint(__uint_as_float(o2) + r.z), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } while (!(d0 && d1 && d2));
#endif
            }
        };
        for (unsigned v0 = v_lo + wave * 64u; v0 < v_hi; v0 += NVDR_LG_THREADS) {
            const unsigned v = v0 + lane;
            unsigned blk = 0u, tag = 0xFFFFu;
            if (v < v_hi) {
                blk = v < n_own ? v : spare_base + (v - n_own);
                tag = tags[blk];
            }
            const bool mine = tag != 0xFFFFu && (tag & 0xFFu) == (unsigned)band;
            unsigned long long m = __ballot(mine);
            if (mine) tags[blk] = 0xFFFFu;                  // consumed: the array reads "no records" again for the next launch
            // NVDR_LG_NB blocks per step: their 2 NB loads are in flight together, then their records are added.  (Rounds 3-5 took one block
            // per step with the next one's two loads in flight: a wavefront's 5-10 blocks were a chain of as many memory round trips, which
            // is what a small launch's gather consisted of -- 76-88 us for the 2.6 M records of one view, 0.26 ms for the 21 M of eight.)
            while (m != 0ull) {
                unsigned bb[NVDR_LG_NB], ff[NVDR_LG_NB];
#pragma unroll
                for (int q = 0; q < NVDR_LG_NB; ++q) {
                    bb[q] = 0u; ff[q] = 0u;
                    if (m != 0ull) {
                        const int j = __builtin_ctzll(m);
                        bb[q] = (unsigned)__builtin_amdgcn_readlane((int)blk, j);
                        ff[q] = (unsigned)__builtin_amdgcn_readlane((int)tag, j) >> 8;
                        m &= m - 1ull;
                    }
                }
                float4 cc[2 * NVDR_LG_NB];
#pragma unroll
                for (int q = 0; q < NVDR_LG_NB; ++q) {
                    cc[2 * q] = none; cc[2 * q + 1] = none;
                    if (!(NVDR_LG_SKIP & 2)) {
                        if (lane < ff[q]) cc[2 * q] = recs[(bb[q] << 7) + lane];
                        if (lane + 64u < ff[q]) cc[2 * q + 1] = recs[(bb[q] << 7) + 64u + lane];
                    } else if (lane < ff[q]) cc[2 * q] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(t_lo + (int)((bb[q] * 64u + lane) % (unsigned)(t_hi - t_lo))));
                }
#pragma unroll
                for (int q = 0; q < 2 * NVDR_LG_NB; ++q) {
                    if (!(NVDR_LG_SKIP & 1)) add(cc[q]);
                    else asm volatile("" :: "v"(cc[q].x), "v"(cc[q].w));
                }
            }
        }
        __syncthreads();
        float *out = partials + (int64_t)g * n_texels * 3 + (int64_t)t_lo * 3;
        if (!(NVDR_LG_SKIP & 4)) for (int i = threadIdx.x; i < n_acc; i += NVDR_LG_THREADS) out[i] = F64 ? (float)lg_acc64[i] : lg_acc[i];
        __syncthreads();
    }
}

// light_grad (+)= sum over the `rows` partial rows (band gather: one row per slice of the key array; atomics mode: the 8
// per-XCD copies).  With pix_count != NULL the chunk may be empty: then the gather wrote nothing and the sum is zero.
__global__ void __launch_bounds__(256) light_grad_reduce_kernel(const float *__restrict__ parts, int n, int rows, float *__restrict__ out,
                                                               int accumulate, const unsigned *__restrict__ pix_count, unsigned pix_begin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
    if (pix_count == nullptr || *pix_count > pix_begin) {
        int k = 0;
        for (; k + 8 <= rows; k += 8) {                 // eight loads in flight (the gather leaves one row per CU)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = parts[(int64_t)(k + j) * n + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; k < rows; ++k) acc += parts[(int64_t)k * n + i];
    }
    if (accumulate) out[i] += acc;
    else out[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// canonical traversal work of the live rays (counting launches only): the BINARY any-hit walk of bvh.h over the same
// rays, i.e. node visits and triangle tests of the reference accounting layout (32-B BVH2 node, 36-B triangle),
// independent of how speculative the production kernel's wide walk is.  tests/test_gpu_bvh.py checks the same counter
// against a CPU walk of the exported tree (oracle_bvh2_walk).
__global__ void __launch_bounds__(NVDR_QUERY_BLOCK) bvh2_count_kernel(BvhView bvh, const float4 *__restrict__ rays,
                                                                       const float4 *__restrict__ pix_origin,
                                                                       const uint32_t *__restrict__ live,
                                                                       const unsigned *__restrict__ ray_count, unsigned rays_per_pixel,
                                                                       int *spill, unsigned long long *out)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const TravStack stack = make_stack(smem, spill, bvh.stack_max, bvh.overflow);
    unsigned nb = 0, nt = 0, nr = 0;
    for (unsigned seg = 0; seg < NVDR_LIVE_SEGS; ++seg) {           // the list's segments, one after the other
        const unsigned total = ray_count[32u * seg], base = seg * ray_count[32u * NVDR_LIVE_SEGS];
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const unsigned slot = live[base + i];
            const float4 rd = rays[slot], ro = pix_origin[slot / rays_per_pixel];
            (void)bvh_any_hit<true>(bvh, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, stack, nb, nt);
            nr++;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        nb += __shfl_xor(nb, o);
        nt += __shfl_xor(nt, o);
        nr += __shfl_xor(nr, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], (unsigned long long)(nb >> 1));   // the binary walk counts two box tests per node visit
        atomicAdd(&out[1], (unsigned long long)nt);
        atomicAdd(&out[2], (unsigned long long)nr);
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers

static int make_tab(Tab &t, const nvdr_tensor &v, int ndim, const char *name)
{
    NVDR_REQUIRE(v.data != nullptr, "env_shade: %s is NULL", name);
    t.p = (const float *)v.data;
    t.n0 = (int)v.size[0]; t.n1 = ndim > 1 ? (int)v.size[1] : 1; t.n2 = ndim > 2 ? (int)v.size[2] : 1;
    t.s0 = (int)v.stride[0]; t.s1 = ndim > 1 ? (int)v.stride[1] : 0; t.s2 = ndim > 2 ? (int)v.stride[2] : 0;
    return 0;
}

static int check_gb(const nvdr_tensor &t, int64_t N, int64_t H, int64_t W, const char *name)
{
    NVDR_REQUIRE(t.data != nullptr, "env_shade: %s is NULL", name);
    NVDR_REQUIRE((t.size[0] == N || t.size[0] == 1) && (t.size[1] == H || t.size[1] == 1) &&
                     (t.size[2] == W || t.size[2] == 1) && (t.size[3] == 3 || t.size[3] == 1),
                 "env_shade: %s has shape [%lld,%lld,%lld,%lld], expected [%lld,%lld,%lld,3] or a broadcastable shape",
                 name, (long long)t.size[0], (long long)t.size[1], (long long)t.size[2], (long long)t.size[3],
                 (long long)N, (long long)H, (long long)W);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// the production traversal launch

// explicit reset of the 64 chunk counters of the traversal kernel (NVDR_DEBUG bit 64 only; see launch_trace)
__global__ void zero_queues_kernel(unsigned *queues) { queues[threadIdx.x * 32u] = 0u; }

static TraceLaunch make_trace_launch(const nvdr_ctx *c, const unsigned *ray_count, unsigned rays_per_pixel, unsigned long long *counters)
{
    TraceLaunch L;
    L.bvh = bvh_view(c);
    L.rays = c->rays; L.pix_origin = c->pix_origin; L.live = c->live;
    L.ray_count = ray_count; L.rays_per_pixel = rays_per_pixel;
    L.vis = c->vis; L.spill = c->spill; L.counters = counters; L.queues = c->queues;
    L.top_nodes = (unsigned)c->trace_top;
    return L;
}

// resident workgroups per CU of the persistent traversal grid: what the compiled occupancy allows and 160 KB of LDS hold (a treetop table
// beyond 64 entries costs resident wavefronts; the grid must not hold workgroups that wait for a slot)
static int trace_blocks_per_cu(size_t lds_bytes, int by_registers)
{
    const int by_lds = (int)((size_t)(160 * 1024) / (lds_bytes ? lds_bytes : 1));
    const int v = by_lds < by_registers ? by_lds : by_registers;
    return v < 1 ? 1 : v;
}

static void launch_trace(nvdr_ctx *c, unsigned blocks, size_t lds, hipStream_t stream, const unsigned *ray_count, unsigned rays_per_pixel,
                         unsigned long long *counters)
{
    // The 64 chunk counters read zero here: the stage-1 / stage-3 kernel (or pack_rays_kernel) that ran before on this stream
    // reset them -- every traversal launch follows one of those (a backward pass that re-traces the forward's stream follows the
    // forward's stage 3; any other launch in between invalidates that stream).  NVDR_DEBUG bit 64 adds an explicit reset kernel.
    if (c->debug & 64u) zero_queues_kernel<<<1, NVDR_TRACE_QUEUES, 0, stream>>>(c->queues);
    if (counters) {
        env_trace_kernel<true><<<blocks, NVDR_QUERY_BLOCK, lds, stream>>>(make_trace_launch(c, ray_count, rays_per_pixel, counters));
        // the same rays twice more through the phase-clock builds (the dealing counters are used up: reset in between; the visibility
        // bytes are rewritten with the same values)
        zero_queues_kernel<<<1, NVDR_TRACE_QUEUES, 0, stream>>>(c->queues);
        env_trace_phase_kernel<1><<<blocks, NVDR_QUERY_BLOCK, lds, stream>>>(make_trace_launch(c, ray_count, rays_per_pixel, counters));
        zero_queues_kernel<<<1, NVDR_TRACE_QUEUES, 0, stream>>>(c->queues);
        env_trace_phase_kernel<2><<<blocks, NVDR_QUERY_BLOCK, lds, stream>>>(make_trace_launch(c, ray_count, rays_per_pixel, counters));
    } else {
        // Which build: the one with split walks in the drain pays for launches of few rays per wavefront and costs the others 3-6 % (trace_kernel.h).
        // The host never learns a launch's live-ray count in time -- but every launch leaves it in the context's host-mapped word, and a context
        // renders the same kind of launch again and again (a HIP-graph capture freezes the choice of its warm-up iterations): the LAST launch's count
        // decides.  NVDR_TRACE_SPLIT_MODE (tuning, read when the context is created): 0 never, 1 always, 2 by the hint (default).
        if (c->trace_top > 0) {
            env_trace_top_kernel<<<blocks, NVDR_QUERY_BLOCK, lds, stream>>>(make_trace_launch(c, ray_count, rays_per_pixel, nullptr));
            return;
        }
        const int mode = c->trace_split_mode;
        const unsigned hint = c->ovf_host ? (unsigned)((volatile int *)c->ovf_host)[1] : 0u;
        const bool split = mode == 1 || (mode == 2 && hint != 0u && hint / (blocks * (NVDR_QUERY_BLOCK / 64u)) < NVDR_TRACE_SPLIT_BELOW);
        if (split) env_trace_split_kernel<<<blocks, NVDR_QUERY_BLOCK, lds, stream>>>(make_trace_launch(c, ray_count, rays_per_pixel, nullptr));
        else env_trace_kernel<false><<<blocks, NVDR_QUERY_BLOCK, lds, stream>>>(make_trace_launch(c, ray_count, rays_per_pixel, nullptr));
    }
}

// Scratch for the ray stream.  Round 1 sized it for the worst case -- every pixel of the launch covered -- which was
// 6.7 GB for the 8-view benchmark (23 % coverage) and rejected launches beyond 2^31 rays.  Now the stream holds ONE CHUNK
// of `cap` covered pixels (x 2S rays x 25 B), cap from the context's byte budget (nvdr_ctx_set_stream_budget; default
// 8 GiB or NVDR_STREAM_BUDGET_MB, read once when the context is created), and a launch
// walks the compacted pixel list chunk by chunk: gen -> trace -> shade per chunk.  The host never learns the covered
// count (no synchronisation): it issues ceil(N*H*W / cap) chunks and the ones behind the device-side count are empty
// launches (~4 us each).  Slot numbers are chunk-local, so the 31-bit limit applies to cap * 2S only.
static int64_t stream_chunk_pixels(const nvdr_ctx *c, int64_t npix, unsigned S, int64_t spare_bytes)
{
    const int64_t per_pixel = (int64_t)2 * S * (16 + 4 + 1 + 4) + 16;
    // the spare blocks of the light-gradient records come out of the same budget (unless that would halve it)
    const int64_t budget = c->stream_budget > 2 * spare_bytes ? c->stream_budget - spare_bytes : c->stream_budget;
    int64_t cap = budget / per_pixel;
    const int64_t lim31 = ((1ll << 31) - 64) / (2ll * S);          // chunk-local slot numbers stay below 2^31
    if (cap > lim31) cap = lim31;
    const int64_t min_cap = (npix + NVDR_MAX_CHUNKS - 1) / NVDR_MAX_CHUNKS;
    if (cap < min_cap) cap = min_cap;
    if (cap < 64) cap = 64;
    if (cap > npix) cap = npix;
    return cap;
}

// `own_slots`: the chunk's stream slots (whole groups of pixels, rounded up to blocks of 128); `spare_slots`: the spare blocks of the
// backward shading kernel's wavefronts behind them (light-gradient records, see env_shade_kernel<true>) -- only the 16-byte `rays`
// array has them
static int reserve_stream(nvdr_ctx *c, int64_t npix, int64_t cap, size_t own_slots, size_t spare_slots, size_t live_slots, hipStream_t stream)
{
    const size_t rays = own_slots;
    const size_t n_tags = (own_slots + spare_slots) >> 7;
    if (c->lg_tags_cap < n_tags) {
        NVDR_HIP_TRY(hipStreamSynchronize(stream));
        ctx_free(c, c->lg_tags);
        c->lg_tags_cap = 0;
        NVDR_HIP_TRY(ctx_malloc(c, &c->lg_tags, sizeof(uint16_t) * n_tags, stream));
        c->lg_tags_cap = n_tags;
        c->lg_tags_dirty = true;
    }
    if (c->stream_cap_rays >= rays && c->stream_cap_total >= rays + spare_slots && c->pix_cap >= npix && c->stream_cap_pixels >= cap) return 0;
    NVDR_HIP_TRY(hipStreamSynchronize(stream));
    if (c->pix_cap < npix) {
        ctx_free(c, c->pix_list);
        c->pix_cap = 0;
        NVDR_HIP_TRY(ctx_malloc(c, &c->pix_list, sizeof(int) * npix, stream));
        c->pix_cap = npix;
    }
    if (c->stream_cap_pixels < cap) {
        ctx_free(c, c->pix_origin);
        ctx_free(c, c->pix_setup);
        ctx_free(c, c->pix_grad);
        c->stream_cap_pixels = 0;
        NVDR_HIP_TRY(ctx_malloc(c, &c->pix_origin, sizeof(float4) * cap, stream));
        NVDR_HIP_TRY(ctx_malloc(c, &c->pix_setup, sizeof(float4) * 4 * cap, stream));
        NVDR_HIP_TRY(ctx_malloc(c, &c->pix_grad, sizeof(float4) * 2 * cap, stream));
        c->stream_cap_pixels = cap;
    }
    if (c->stream_cap_rays < rays) {
        ctx_free(c, c->texel);
        ctx_free(c, c->vis);
        c->stream_cap_rays = 0;
        NVDR_HIP_TRY(ctx_malloc(c, &c->texel, sizeof(int) * rays, stream));
        NVDR_HIP_TRY(ctx_malloc(c, &c->vis, rays, stream));
        c->stream_cap_rays = rays;
    }
    if (c->stream_cap_live < live_slots) {
        ctx_free(c, c->live);
        c->stream_cap_live = 0;
        NVDR_HIP_TRY(ctx_malloc(c, &c->live, sizeof(uint32_t) * live_slots, stream));
        c->stream_cap_live = live_slots;
    }
    if (c->stream_cap_total < rays + spare_slots) {
        ctx_free(c, c->rays);
        c->stream_cap_total = 0;
        NVDR_HIP_TRY(ctx_malloc(c, &c->rays, sizeof(float4) * (rays + spare_slots), stream));
        c->stream_cap_total = rays + spare_slots;
    }
    c->stream_id = 0;
    return 0;
}

// LDS the band gather may use per workgroup (bytes); > 64 KB needs the attribute below
static size_t lg_lds_budget()
{
    static size_t budget = 0;
    if (budget) return budget;
    size_t kb = 96;
    if (const char *e = nvdr_tuning_env("NVDR_LG_LDS_KB")) kb = (size_t)atoll(e);
    if (kb < 8) kb = 8;
    if (kb > 160) kb = 160;
    size_t want = kb * 1024;
    if (want > 64 * 1024 &&
        (hipFuncSetAttribute((const void *)light_grad_block_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess ||
         hipFuncSetAttribute((const void *)light_grad_block_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess)) {
        (void)hipGetLastError();
        want = 64 * 1024;
    }
    budget = want;
    return budget;
}

static int make_view4s(View4s &v, const nvdr_tensor &t, const char *name)
{
    for (int d = 0; d < 4; ++d)
        NVDR_REQUIRE(t.size[d] == 1 || llabs(t.stride[d]) < (1ll << 31), "env_shade: %s has a stride beyond 2^31 elements", name);
    v.p = (const float *)t.data;
    v.s0 = t.size[0] == 1 ? 0 : (int)t.stride[0];
    v.s1 = t.size[1] == 1 ? 0 : (int)t.stride[1];
    v.s2 = t.size[2] == 1 ? 0 : (int)t.stride[2];
    v.s3 = t.size[3] == 1 ? 0 : (int)t.stride[3];
    return 0;
}

static int env_shade_launch(nvdr_ctx *c, const nvdr_env_shade_args *a, bool backward, hipStream_t stream)
{
    NVDR_REQUIRE(c && a, "env_shade: NULL argument");
    NVDR_REQUIRE(c->n_tris > 0, "env_shade: no BVH built on this context (call optix_build_bvh first)");
    if (int r0 = ctx_check_overflow(c, "env_shade")) return r0;
    NVDR_REQUIRE(a->bsdf <= 2, "env_shade: BSDF index %u out of range", a->bsdf);
    NVDR_REQUIRE(!a->rnd_seed_snapshot || a->rnd_seed_offset, "env_shade: rnd_seed_snapshot without rnd_seed_offset");
    NVDR_REQUIRE(a->n_samples_x >= 1 && a->n_samples_x <= 256, "env_shade: n_samples_x %u out of range", a->n_samples_x);
    const int64_t N = a->ro.size[0], H = a->ro.size[1], W = a->ro.size[2];
    NVDR_REQUIRE(N > 0 && H > 0 && W > 0 && N * H * W < (1ll << 31), "env_shade: bad launch extent");
    int r;
    if ((r = check_gb(a->ro, N, H, W, "ro"))) return r;
    if ((r = check_gb(a->gb_pos, N, H, W, "gb_pos"))) return r;
    if ((r = check_gb(a->gb_normal, N, H, W, "gb_normal"))) return r;
    if ((r = check_gb(a->gb_view_pos, N, H, W, "gb_view_pos"))) return r;
    if ((r = check_gb(a->gb_kd, N, H, W, "gb_kd"))) return r;
    if ((r = check_gb(a->gb_ks, N, H, W, "gb_ks"))) return r;
    NVDR_REQUIRE(a->mask.data && a->mask.size[0] == N && a->mask.size[1] == H && a->mask.size[2] == W,
                 "env_shade: mask must be [N,H,W]");
    const unsigned S = a->n_samples_x * a->n_samples_x;
    NVDR_REQUIRE(a->perms.data && a->perms.size[1] == (int64_t)S && a->perms.size[0] > 0,
                 "env_shade: perms must be int32 [NP, %u]", S);
    NVDR_REQUIRE(a->light.size[2] == 3 || a->light.size[2] == 1, "env_shade: light must be [Hl,Wl,3]");
    NVDR_REQUIRE(a->pdf.size[0] == a->cols.size[0] && a->pdf.size[1] == a->cols.size[1] && a->rows.size[0] == a->pdf.size[0],
                 "env_shade: pdf/rows/cols shapes disagree");
    const int64_t npix = N * H * W;
    NVDR_HIP_TRY(hipSetDevice(c->device));
    int L = 1, lg = 0;
    while (L < (int)S && L < 64) { L <<= 1; lg++; }
    const int G = 64 / L;                                       // pixels per wavefront round (a "group")
    const int64_t group_slots = (int64_t)G * 2 * S;
    // (small launches -- up to four views of 512^2 -- start the generation kernel with 4 workgroups per CU, exactly what is resident:
    // its wavefronts hold BSDF tasks back until 64 of one lobe wait and drain the rest when they run out of pixels, so few pixels per
    // wavefront mean many half-empty batches: 0.301 vs 0.315 ms for one view; 8 per CU is 3 % better for eight views)
    int per_cu_launch[3] = {c->per_cu[0], c->per_cu[1], c->per_cu[2]};
    if (!c->per_cu_user && npix <= (1ll << 20)) {
        // (one view: 3 -- the generation kernel's workgroups take a quarter of a CU's registers and LDS each, and with all four resident
        // the BVH build on the side stream cannot place a workgroup until they exit: generation 0.262 -> 0.287 ms, the traversal's wait
        // for the tree 0.458 -> 0.423 ms from launch to end, iteration 2.062 -> 2.042 ms; 2: 0.334 / 0.389 / 2.076)
        per_cu_launch[0] = npix <= (1ll << 18) ? 3 : 4;
        per_cu_launch[2] = 8;        // (backward shading of one view: 0.465 vs 0.483 ms with 8 instead of 6 workgroups per CU; 8 views: +1.5 %)
    }
    const int *per_cu = per_cu_launch;   // blocks per CU of the three per-pixel kernels (generation, forward shading, backward shading): {10, 6, 6},
                                     // measured within 3 % of the best for each kernel; NVDR_PBLOCKS="g,f,b" is read once per context
    const int waves_per_block = 4;

    // Light gradient: records + LDS gather when the probe's accumulators fit <= 16 LDS bands (and a group of pixels <= 16 blocks of
    // 128 slots: n_samples_x <= 32), memory-side atomics otherwise.  Decided the same way by the forward and the backward launch: the
    // forward already reserves the spare blocks the backward shading kernel's wavefronts will need (a reallocation in between
    // would lose the forward's stream).
    const int n_texels = (int)(a->light.size[0] * a->light.size[1]);
    int n_bands = 0, band_texels = 0, lg_rows = 8, lg_shift = 0, lg_records = 0, lg_grid_y = 1;
    size_t lg_lds = 0;
    bool lg_f64 = false;
    {
        const size_t lds_budget = lg_lds_budget();
        // a band = the largest power-of-two number of texels whose accumulators fit the LDS budget (band = texel >> shift): fp64 accumulators
        // (24 bytes per texel, the ds_add_f64 build of the gather) when the probe then still fits 16 bands, fp32 ones otherwise
        while ((size_t)(2 << lg_shift) * 12 <= lds_budget) ++lg_shift;
        // one gather workgroup per CU (its accumulators take most of the CU's LDS): each walks all bands (large launches), or
        // the CUs are dealt to the bands (small launches: fewer passes and partial rows, at the price of uneven bands)
        const bool per_band = c->lg_mode >= 0 ? c->lg_mode == 1 : npix * 2 * (int64_t)S <= NVDR_LG_PER_BAND_MAX_SLOTS;
        // fp64 for the small launches only (NVDR_LG_F64: -1 this rule, 0 never, 1 wherever the bands allow): a workgroup that walks ALL bands pays
        // for twice as many passes what ds_add_f64 saves (session 27: 8 views +1.1 %, one view -6.2 % of backward shading + gather)
        if ((c->lg_f64 > 0 || (c->lg_f64 < 0 && per_band)) && lg_shift > 0 && (n_texels + (1 << (lg_shift - 1)) - 1) / (1 << (lg_shift - 1)) <= 16) {
            lg_f64 = true;
            --lg_shift;
        }
        band_texels = 1 << lg_shift;
        n_bands = (n_texels + band_texels - 1) / band_texels;
        lg_records = (n_bands <= 16 && group_slots <= 16 * 128 && !(c->debug & 16u)) ? 1 : 0;
        if (lg_records) {
            if (band_texels > n_texels) band_texels = n_texels;
            lg_lds = (size_t)band_texels * (lg_f64 ? 24 : 12);
            // (per band, one view, round 4: 4 / 8 / 16 workgroups per band instead of n_cus / n_bands = 32: +61 / +27 / +9 % of the backward
            // shading + gather time -- the records, not the partial rows, are the work; 64: -0.5 %)
            lg_rows = per_band ? (c->n_cus / n_bands < 1 ? 1 : c->n_cus / n_bands) : c->n_cus;
            lg_grid_y = per_band ? n_bands : 1;
        }
    }
    // S == 64 (one pixel per wavefront round, lane = stratum): the shading kernels that queue the live light samples across pixels
    // (env_shade_queue_kernel; NVDR_SHADE_QUEUE bit 0 backward, bit 1 forward).  The atomics fallback of the light gradient and the
    // contention experiment (NVDR_DEBUG bit 4) stay with the plain kernels.  Decided the same way by the forward and the backward
    // launch (the grid of the backward kernel sizes the spare blocks).  Their workgroups hold 31 / 49 KB of LDS -- 5 / 3 resident per
    // CU -- and a finer split evens out what the pixels cost: forward 15 per CU (0.823 ms per 8-view launch of bob at 64 spp; 5: 0.866,
    // 10: 0.839, 30: 0.825), backward 12 (2.52 ms; 6: 2.60, 9: 2.58, 18: 2.51); one view: 10 and 3 (0.120 / 0.405 ms; 15 and 12: 0.120 / 0.432).
    // Only S == 64: a version of the kernels with one result row per ROUND of a pixel (S = 256: four) was measured and dropped -- the
    // plain kernels keep a lane's sums in registers across the rounds and won there (256 spp, 4 views of spot: forward 1.35 against
    // 1.94 ms, backward 5.28 against 5.55), and the generality cost the one-round case 8 % (32 more VGPRs).
    const bool queue_ok = L == 64 && S == 64;
    const bool queue_fwd = (c->shade_queue & 2) && queue_ok;
    const bool queue_bwd = (c->shade_queue & 1) && queue_ok && lg_records && !(c->debug & 4u);
    // S > 64 (several rounds per pixel): the kernels with a pixel-local queue (env_shade_local_kernel).  Backward by default (bit 0): 256 spp,
    // 4 views of spot 5.45 -> 4.58 ms, 8 views of the 171 k-triangle mesh 9.94 -> 8.08 ms per launch.  The forward kernel gains nothing from
    // the fuller passes (1.26 -> 1.27 ms, 2.39 -> 2.51: it waits for its loads, not for its lanes) and stays plain unless bit 2 is set.
    const bool local_ok = L == 64 && S > 64;
    const bool local_fwd = (c->shade_queue & 4) && local_ok;
    const bool local_bwd = (c->shade_queue & 1) && local_ok && lg_records && !(c->debug & 4u);
    if (!c->per_cu_user) {
        if (queue_fwd) per_cu_launch[1] = npix <= (1ll << 18) ? 10 : 15;
        if (queue_bwd) per_cu_launch[2] = npix <= (1ll << 18) ? 3 : (npix <= (1ll << 20) ? 9 : 12);     // (2 / 4 views: 9: 0.78 / 1.33 ms, 3: 0.78 / 1.39, 12: 0.79 / 1.37)
        if (local_bwd && npix > (1ll << 18)) per_cu_launch[2] = 12;     // (4 views of spot at 256 spp: 3: 4.87 ms, 6: 4.72, 9: 4.65, 12: 4.61)
    }
    // spare blocks per wavefront of the backward shading kernel: one open block per band + the records of the first group (nothing
    // consumed yet) + the alignment of its range to 128 slots + the round in flight (env_shade_kernel<true>)
    const unsigned lg_spw = lg_records ? (unsigned)(n_bands + (group_slots + 127) / 128 + 2) : 0u;
    const int64_t bwd_waves_max = (int64_t)c->n_cus * (per_cu[2] < 1 ? 1 : per_cu[2]) * waves_per_block;
    const int64_t cap = stream_chunk_pixels(c, npix, S, bwd_waves_max * lg_spw * 128 * 16);
    // chunk-local slot numbers are 31-bit: stream_chunk_pixels may have RAISED cap to npix / NVDR_MAX_CHUNKS behind its own clamp
    // (a launch so large that it cannot be cut into NVDR_MAX_CHUNKS chunks below the limit) -- an error, not a silent wrap-around
    NVDR_REQUIRE(cap * 2ll * (int64_t)S < (1ll << 31), "env_shade: %lld pixels at %u strata need chunks of %lld pixels, beyond the 2^31 "
                 "ray slots of one chunk of the ray stream (at most %d chunks per launch): render fewer views per launch",
                 (long long)npix, S, (long long)cap, NVDR_MAX_CHUNKS);
    const int n_chunks = (int)((npix + cap - 1) / cap);
    // a->phase (forward): the launch issued in two calls, cut between the sample generation and the traversal (nvdr_hip.h)
    NVDR_REQUIRE(a->phase <= 2u && !(backward && a->phase), "env_shade: phase %u (0 whole launch, 1 up to the sample generation, 2 the rest; forward only)", a->phase);
    const bool cut = a->phase != 0u && n_chunks == 1;
    if (a->phase == 1u && !cut) return 0;                       // a launch of several chunks cannot be cut: phase 2 runs all of it
    const bool front = !(cut && a->phase == 2u), back = !(cut && a->phase == 1u);
    // the chunk is raised to npix / NVDR_MAX_CHUNKS when the byte budget asks for more chunks than that; chunk-local slot numbers
    // must still fit 31 bits (they are stored as unsigned / int in the live list, the light-gradient keys and the band gather)
    NVDR_REQUIRE(cap * 2 * (int64_t)S <= (1ll << 31) - 64,
                 "env_shade: %lld pixels x %u rays per pixel cannot be cut into %d chunks of < 2^31 rays; raise the stream budget "
                 "(nvdr_ctx_set_stream_budget) or launch fewer views at once", (long long)npix, 2 * S, NVDR_MAX_CHUNKS);

    ShadeParams p;
    memset(&p, 0, sizeof(p));
    if ((r = make_view4s(p.ro, a->ro, "ro")) || (r = make_view4s(p.pos, a->gb_pos, "gb_pos")) || (r = make_view4s(p.nrm, a->gb_normal, "gb_normal")) ||
        (r = make_view4s(p.view_pos, a->gb_view_pos, "gb_view_pos")) || (r = make_view4s(p.kd, a->gb_kd, "gb_kd")) || (r = make_view4s(p.ks, a->gb_ks, "gb_ks")))
        return r;
    p.mask = (const float *)a->mask.data;
    NVDR_REQUIRE(llabs(a->mask.stride[0]) < (1ll << 31) && llabs(a->mask.stride[1]) < (1ll << 31) && llabs(a->mask.stride[2]) < (1ll << 31),
                 "env_shade: mask strides beyond 2^31 elements");
    p.ms0 = (int)a->mask.stride[0]; p.ms1 = (int)a->mask.stride[1]; p.ms2 = (int)a->mask.stride[2];
    if ((r = make_tab(p.light, a->light, 3, "light"))) return r;
    if ((r = make_tab(p.pdf, a->pdf, 2, "pdf"))) return r;
    if ((r = make_tab(p.rows, a->rows, 1, "rows"))) return r;
    if ((r = make_tab(p.cols, a->cols, 2, "cols"))) return r;
    p.perms = (const int *)a->perms.data;
    p.perm_s0 = (int)a->perms.stride[0]; p.perm_s1 = (int)a->perms.stride[1];
    p.n_perms = (unsigned)a->perms.size[0];
    p.N = (int)N; p.H = (int)H; p.W = (int)W;
    p.bsdf = a->bsdf; p.n = a->n_samples_x; p.S = S; p.seed = a->rnd_seed; p.pix_offset = a->pixel_index_offset;
    p.shadow_scale = a->shadow_scale;
    p.seed_dev = (!backward && a->rnd_seed_snapshot) ? a->rnd_seed_snapshot : a->rnd_seed_offset;
    p.L = L; p.log2L = lg;
    p.vis_cache = a->vis_cache;
    p.vis_words = (int)((S + 31) / 32);
    p.debug = c->debug;
    p.pix_cap = (unsigned)cap;

    p.lg_records = backward ? lg_records : 0;
    const bool shade_queue = backward ? queue_bwd : queue_fwd;
    const bool shade_local = backward ? local_bwd : local_fwd;
    p.lg_shift = lg_shift;

    if (!backward) {
        NVDR_REQUIRE(a->diff && a->spec, "env_shade_fwd: NULL output");
        p.diff = a->diff; p.spec = a->spec;       // torch::zeros (torch_bindings.cpp:148-149): the uncovered pixels are zeroed by compact_pixels_kernel below
    } else {
        NVDR_REQUIRE(a->gb_pos_grad && a->gb_normal_grad && a->gb_kd_grad && a->gb_ks_grad && a->light_grad,
                     "env_shade_bwd: NULL output");
        if ((r = check_gb(a->diff_grad, N, H, W, "diff_grad"))) return r;
        if ((r = check_gb(a->spec_grad, N, H, W, "spec_grad"))) return r;
        if ((r = make_view4s(p.dgrad, a->diff_grad, "diff_grad")) || (r = make_view4s(p.sgrad, a->spec_grad, "spec_grad"))) return r;
        p.g_pos = a->gb_pos_grad; p.g_nrm = a->gb_normal_grad; p.g_kd = a->gb_kd_grad; p.g_ks = a->gb_ks_grad;
        p.g_light = a->light_grad;
        // (a kernel, never hipMemsetAsync: this launch is replayed from HIP graphs, and a memset node inside a captured side branch was what
        // stalled the refit graph of round 6 -- no memset or memcpy node on any path the harness captures)
        zero_outputs_kernel<<<dim3(min(div_up(3 * npix, 1024), 2048u), 4), 256, 0, stream>>>(p.g_pos, p.g_nrm, p.g_kd, p.g_ks, 3 * npix);
        p.light_elems = 3 * n_texels;
        const size_t need = (size_t)p.light_elems * (size_t)(lg_rows > 8 ? lg_rows : 8);
        if (c->lg_cap < need) {
            NVDR_HIP_TRY(hipStreamSynchronize(stream));
            ctx_free(c, c->lg_part);
            c->lg_cap = 0;
            NVDR_HIP_TRY(ctx_malloc(c, &c->lg_part, sizeof(float) * need, stream));
            c->lg_cap = need;
        }
        p.g_light_xcd = c->lg_part;
        if (!p.lg_records || (c->debug & 2u)) {
            {
                const int64_t q = 2 * (int64_t)p.light_elems;       // (the eight per-XCD copies as four quarters)
                zero_outputs_kernel<<<dim3(min(div_up(q, 1024), 2048u), 4), 256, 0, stream>>>(c->lg_part, c->lg_part + q, c->lg_part + 2 * q, c->lg_part + 3 * q, q);
            }
            if (c->debug & 2u) NVDR_HIP_TRY(hipMemsetAsync(p.g_light, 0, sizeof(float) * p.light_elems, stream));
        }
    }
    // persistent grids (the covered-pixel count lives on the device)
    const int64_t max_groups = (cap + G - 1) / G;
    int64_t pb[3];
    for (int k = 0; k < 3; ++k) {
        pb[k] = (int64_t)c->n_cus * (per_cu[k] < 1 ? 1 : per_cu[k]);
        if (pb[k] * waves_per_block > max_groups) pb[k] = (max_groups + waves_per_block - 1) / waves_per_block;
        if (pb[k] < 1) pb[k] = 1;
    }
    const size_t own_slots = ((size_t)max_groups * group_slots + 127) & ~(size_t)127;
    const size_t spare_blocks = (size_t)pb[2] * waves_per_block * lg_spw;
    // the live list's segments: generation wavefront w appends to segment w % NVDR_LIVE_SEGS, so a segment takes at most the slots of the
    // groups of ceil(waves / SEGS) wavefronts
    const int64_t gen_waves = pb[0] * waves_per_block;
    const int64_t seg_cap64 = (int64_t)live_segment_capacity((unsigned long long)gen_waves, (unsigned long long)max_groups, (unsigned long long)group_slots);
    NVDR_REQUIRE(seg_cap64 * NVDR_LIVE_SEGS < (1ll << 32), "nvdr_env_shade: the live-ray list of a chunk of %lld pixels does not fit 32-bit positions", (long long)cap);
    const unsigned seg_cap = (unsigned)seg_cap64;
    if ((r = reserve_stream(c, npix, cap, own_slots, spare_blocks * 128, (size_t)seg_cap64 * NVDR_LIVE_SEGS, stream))) return r;
    p.lg_tags = c->lg_tags;
    p.lg_spare_base = (unsigned)(own_slots >> 7);
    p.lg_spw = lg_spw;
    if (backward && lg_records) {
        if (c->lg_tags_dirty) NVDR_HIP_TRY(hipMemsetAsync(c->lg_tags, 0xFF, sizeof(uint16_t) * c->lg_tags_cap, stream));
        c->lg_tags_dirty = true;        // until the gather of the last chunk has been enqueued
    }
    p.pix_list = c->pix_list;
    p.pix_count = &c->dinfo->pix_count;
    p.rays = c->rays; p.texel = c->texel; p.pix_origin = c->pix_origin; p.pix_setup = c->pix_setup; p.pix_grad = c->pix_grad; p.vis = c->vis;
    p.live = c->live;
    p.queues = c->queues;

    // backward pass of a forward launch whose work list (and, if it fitted one chunk, ray stream) is still in the context
    const bool reuse = backward && a->reuse_stream_id != 0 && a->reuse_stream_id == c->stream_id;
    p.reuse = reuse ? 1 : 0;
#ifndef NVDR_TRACE_BLOCKS_PER_CU
#define NVDR_TRACE_BLOCKS_PER_CU 8
#endif
    const size_t trace_lds = NVDR_TRACE_LDS_BYTES(NVDR_QUERY_BLOCK, c->trace_top);
    int64_t tblocks = (int64_t)c->n_cus * trace_blocks_per_cu(trace_lds, NVDR_TRACE_BLOCKS_PER_CU);
    if (tblocks > NVDR_QUERY_MAX_BLOCKS) tblocks = NVDR_QUERY_MAX_BLOCKS;
    {
        const int64_t need = (cap * 2 * S + NVDR_QUERY_BLOCK - 1) / NVDR_QUERY_BLOCK;
        if (tblocks > need) tblocks = need < NVDR_TRACE_QUEUES / 4 ? NVDR_TRACE_QUEUES / 4 : need;     // (at least 64 wavefronts: one per dealing counter, trace_kernel.h)
    }
    const size_t count_lds = NVDR_STACK_LDS_BYTES(NVDR_QUERY_BLOCK);
    const bool replay = backward && p.vis_cache != nullptr;   // forward bits handed back by the caller: no traversal

    // guide tables of the CDF inversion (rebuilt per launch: the light is a trained parameter); tables up to 2^15 entries per row
    p.cdf_guide = nullptr;
    if (!(reuse && n_chunks == 1) && a->rows.size[0] <= 32768 && a->cols.size[1] <= 32768 && NVDR_CDF_GUIDE) {
        int lr = 0, lc = 0;
        while ((1 << lr) < (int)a->rows.size[0]) ++lr;
        while ((1 << lc) < (int)a->cols.size[1]) ++lc;
        const size_t pitch = (size_t)1 << (lr > lc ? lr : lc), need = pitch * ((size_t)a->cols.size[0] + 1);
        if (c->guide_cap < need) {
            NVDR_HIP_TRY(hipStreamSynchronize(stream));
            ctx_free(c, c->cdf_guide);
            c->guide_cap = 0;
            NVDR_HIP_TRY(ctx_malloc(c, &c->cdf_guide, sizeof(uint16_t) * need, stream));
            c->guide_cap = need;
        }
        if (front) cdf_guide_kernel<<<div_up((int64_t)need, 256), 256, 0, stream>>>(p.rows, p.cols, lr, lc, c->cdf_guide);
        p.cdf_guide = c->cdf_guide;
        p.guide_log_rows = lr;
        p.guide_log_cols = lc;
    }
    c->stream_id = 0; // invalid while being rewritten
    if (front)
    begin_launch_kernel<<<1, 256, 0, stream>>>(&c->dinfo->pix_count, c->chunk_counts, n_chunks, p.reuse, p.pix_cap, seg_cap,
                                               const_cast<unsigned *>(a->rnd_seed_offset), backward ? nullptr : a->rnd_seed_snapshot, a->rnd_seed_advance);
    if (!reuse && front)
        compact_pixels_kernel<<<div_up(npix, 256 * NVDR_COMPACT_ROUNDS), 256, 0, stream>>>(p.mask, p.ms0, p.ms1, p.ms2, p.N, p.H, p.W, c->pix_list,
                                                                      &c->dinfo->pix_count, backward ? nullptr : p.diff, backward ? nullptr : p.spec);
    for (int k = 0; k < n_chunks; ++k) {
        p.pix_begin = (unsigned)((int64_t)k * cap);
        p.ray_count = c->chunk_counts + (size_t)k * NVDR_LIVE_WORDS;
        hipEvent_t *pe = nullptr;
        if (c->profiling && n_chunks <= NVDR_PROF_RING && !cut) {      // (a launch of more chunks than records would overwrite its own first chunks; a cut launch is not timed)
            const int slot = (int)(c->prof_n % NVDR_PROF_RING);
            pe = c->prof_ev[slot];
            c->prof_kind[slot] = (backward ? 1 : 0) | (k == 0 ? 2 : 0);
            c->prof_n++;
            NVDR_HIP_TRY(hipEventRecord(pe[0], stream));
        }
        // stage 1 (skipped on the host when the forward's stream is known to be whole: one chunk covers the launch;
        // otherwise the kernel itself decides from the device-side pixel count)
        if (!(reuse && n_chunks == 1) && front) {
            NvdrRange r("nvdr:gen");
            if (c->debug) env_gen_kernel<true><<<(unsigned)pb[0], 256, 0, stream>>>(p);
            else env_gen_kernel<false><<<(unsigned)pb[0], 256, 0, stream>>>(p);
        }
        if (pe) NVDR_HIP_TRY(hipEventRecord(pe[1], stream));
        if (!back) break;                                       // phase 1 of a cut launch ends here (one chunk)
        // stage 2 (the first launch of this call that needs the tree: a build may still be running on the context's side stream)
        if (!replay) {
            NvdrRange r("nvdr:trace");
            if (int rw = ctx_wait_built(c, stream)) return rw;
            if (c->debug & 1u) {
                NVDR_HIP_TRY(hipMemsetAsync(c->vis, 1, (size_t)cap * 2 * S, stream));
            } else {
                launch_trace(c, (unsigned)tblocks, trace_lds, stream, p.ray_count, 2 * S, a->counters);
                if (a->counters)
                    bvh2_count_kernel<<<(unsigned)tblocks, NVDR_QUERY_BLOCK, count_lds, stream>>>(bvh_view(c), c->rays, c->pix_origin, c->live, p.ray_count,
                                                                                                2 * S, c->spill, a->counters + NVDR_COUNTERS_BVH2);
            }
        }
        if (pe) NVDR_HIP_TRY(hipEventRecord(pe[2], stream));
        // stage 3
        NvdrRange r3(backward ? "nvdr:shade_bwd+light_grad" : "nvdr:shade_fwd");
        if (backward) {
            pack_grads_kernel<<<min(div_up(cap, 256), 2048u), 256, 0, stream>>>(p);
            if (shade_queue) { if (c->debug) env_shade_queue_kernel<true, true><<<(unsigned)pb[2], 256, 0, stream>>>(p); else env_shade_queue_kernel<true, false><<<(unsigned)pb[2], 256, 0, stream>>>(p); }
            else if (shade_local) { if (c->debug) env_shade_local_kernel<true, true><<<(unsigned)pb[2], 256, 0, stream>>>(p); else env_shade_local_kernel<true, false><<<(unsigned)pb[2], 256, 0, stream>>>(p); }
            else { if (c->debug) env_shade_kernel<true, true><<<(unsigned)pb[2], 256, 0, stream>>>(p); else env_shade_kernel<true, false><<<(unsigned)pb[2], 256, 0, stream>>>(p); }
            if (p.lg_records && !(c->debug & 2u)) {
                if (lg_f64)
                    light_grad_block_kernel<true><<<dim3((unsigned)lg_rows, (unsigned)lg_grid_y), NVDR_LG_THREADS, lg_lds, stream>>>(
                        c->lg_tags, c->rays, p.pix_count, p.pix_begin, p.pix_cap, (unsigned)G, (unsigned)group_slots, p.lg_spare_base,
                        (unsigned)spare_blocks, 1 << p.lg_shift, n_bands, n_texels, c->lg_part);
                else
                    light_grad_block_kernel<false><<<dim3((unsigned)lg_rows, (unsigned)lg_grid_y), NVDR_LG_THREADS, lg_lds, stream>>>(
                        c->lg_tags, c->rays, p.pix_count, p.pix_begin, p.pix_cap, (unsigned)G, (unsigned)group_slots, p.lg_spare_base,
                        (unsigned)spare_blocks, 1 << p.lg_shift, n_bands, n_texels, c->lg_part);
                light_grad_reduce_kernel<<<div_up(p.light_elems, 256), 256, 0, stream>>>(c->lg_part, p.light_elems, lg_rows, p.g_light,
                                                                                         k > 0 ? 1 : 0, p.pix_count, p.pix_begin);
            }
        } else {
            if (shade_queue) { if (c->debug) env_shade_queue_kernel<false, true><<<(unsigned)pb[1], 256, 0, stream>>>(p); else env_shade_queue_kernel<false, false><<<(unsigned)pb[1], 256, 0, stream>>>(p); }
            else if (shade_local) { if (c->debug) env_shade_local_kernel<false, true><<<(unsigned)pb[1], 256, 0, stream>>>(p); else env_shade_local_kernel<false, false><<<(unsigned)pb[1], 256, 0, stream>>>(p); }
            else { if (c->debug) env_shade_kernel<false, true><<<(unsigned)pb[1], 256, 0, stream>>>(p); else env_shade_kernel<false, false><<<(unsigned)pb[1], 256, 0, stream>>>(p); }
        }
        if (pe) NVDR_HIP_TRY(hipEventRecord(pe[3], stream));
    }
    if (!back) { NVDR_LAUNCH_CHECK(); return 0; }              // (the stream becomes valid when phase 2 has been enqueued)
    if (backward && p.lg_records) c->lg_tags_dirty = false;    // every chunk's gather consumed (and reset) the tags its shading kernel wrote
    if (backward && !p.lg_records && !(c->debug & 2u))
        light_grad_reduce_kernel<<<div_up(p.light_elems, 256), 256, 0, stream>>>(c->lg_part, p.light_elems, 8, p.g_light, 0, nullptr, 0);
    NVDR_LAUNCH_CHECK();
    // The stream a later backward pass may reuse is the one a FORWARD launch wrote; the record-writing backward pass
    // consumes it (ray directions are overwritten by light-gradient records), so it is invalid afterwards.
    c->stream_id = (backward && p.lg_records) ? 0 : (reuse ? a->reuse_stream_id : ++c->stream_seq);
    return 0;
}

// Test hook: any-hit visibility of ARBITRARY rays through the PRODUCTION shadow-ray kernel (env_trace_kernel: persistent
// wavefronts, wide nodes, lane refill) -- the rays are packed into a one-ray-per-pixel stream.  nvdr_trace_visibility
// (bvh.hip) answers the same question with the binary walk; tests require the two to agree bit for bit.
__global__ void pack_rays_kernel(const float *__restrict__ ro, const float *__restrict__ rd, unsigned n, float4 *__restrict__ rays,
                                 float4 *__restrict__ origin, uint32_t *__restrict__ live, unsigned *count, unsigned *queues)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= NVDR_LIVE_SEGS) count[32u * i] = (i == 0u || i == NVDR_LIVE_SEGS) ? n : 0u;       // one segment holds them all
    if (i < NVDR_TRACE_QUEUES) queues[i * 32u] = 0u;
    if (i >= n) return;
    rays[i] = make_float4(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2], 1.0f);
    origin[i] = make_float4(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2], 0.0f);
    live[i] = i;
}

static int trace_visibility_wide(nvdr_ctx *c, const float *ro, const float *rd, int64_t n_rays, uint8_t *out_vis, unsigned long long *counters,
                                 hipStream_t stream, const char *who)
{
    NVDR_REQUIRE(c && c->n_tris > 0, "%s: no BVH built", who);
    NVDR_REQUIRE(n_rays < (1ll << 30), "%s: too many rays", who);
    if (int r0 = ctx_check_overflow(c, who)) return r0;
    if (n_rays <= 0) return 0;
    NVDR_HIP_TRY(hipSetDevice(c->device));
    int r = reserve_stream(c, n_rays, n_rays, ((size_t)n_rays * 2 + 127) & ~(size_t)127, 0, (size_t)n_rays, stream);
    if (r) return r;
    c->stream_id = 0;
    if (int rw = ctx_wait_built(c, stream)) return rw;
    pack_rays_kernel<<<div_up(n_rays, 256), 256, 0, stream>>>(ro, rd, (unsigned)n_rays, c->rays, c->pix_origin, c->live, c->chunk_counts, c->queues);
    const size_t trace_lds = NVDR_TRACE_LDS_BYTES(NVDR_QUERY_BLOCK, c->trace_top);
    int64_t tblocks = (int64_t)c->n_cus * trace_blocks_per_cu(trace_lds, 8);
    if (tblocks > NVDR_QUERY_MAX_BLOCKS) tblocks = NVDR_QUERY_MAX_BLOCKS;
    const int64_t need = (n_rays + NVDR_QUERY_BLOCK - 1) / NVDR_QUERY_BLOCK;
    if (tblocks > need) tblocks = need < NVDR_TRACE_QUEUES / 4 ? NVDR_TRACE_QUEUES / 4 : need;
    launch_trace(c, (unsigned)tblocks, trace_lds, stream, c->chunk_counts, 1u, counters);
    NVDR_HIP_TRY(hipMemcpyAsync(out_vis, c->vis, (size_t)n_rays, hipMemcpyDeviceToDevice, stream));
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_trace_visibility_wide(nvdr_ctx *c, const float *ro, const float *rd, int64_t n_rays, uint8_t *out_vis, void *stream_)
{
    return trace_visibility_wide(c, ro, rd, n_rays, out_vis, nullptr, (hipStream_t)stream_, "nvdr_trace_visibility_wide");
}

// the same through the COUNTING build of the production kernel: counters as nvdr_env_shade_args.counters
extern "C" int nvdr_trace_visibility_wide_counted(nvdr_ctx *c, const float *ro, const float *rd, int64_t n_rays, uint8_t *out_vis,
                                                  unsigned long long *counters, void *stream_)
{
    NVDR_REQUIRE(counters != nullptr, "nvdr_trace_visibility_wide_counted: counters is NULL");
    return trace_visibility_wide(c, ro, rd, n_rays, out_vis, counters, (hipStream_t)stream_, "nvdr_trace_visibility_wide_counted");
}

extern "C" int nvdr_ctx_set_stream_budget(nvdr_ctx *c, int64_t bytes)
{
    NVDR_REQUIRE(c, "nvdr_ctx_set_stream_budget: NULL ctx");
    NVDR_REQUIRE(bytes >= (1 << 20), "nvdr_ctx_set_stream_budget: %lld bytes is below the 1 MiB minimum", (long long)bytes);
    c->stream_budget = bytes;
    c->stream_id = 0;     // a stream written with another chunk size must not be reused
    return 0;
}

extern "C" int nvdr_ctx_set_profiling(nvdr_ctx *c, int enable)
{
    NVDR_REQUIRE(c, "nvdr_ctx_set_profiling: NULL ctx");
    NVDR_HIP_TRY(hipSetDevice(c->device));
    if (enable && !c->prof_ev[0][0]) {
        for (int i = 0; i < NVDR_PROF_RING; ++i)
            for (int k = 0; k < 4; ++k) NVDR_HIP_TRY(hipEventCreate(&c->prof_ev[i][k]));
    }
    c->profiling = enable != 0;
    c->prof_n = 0;
    return 0;
}

extern "C" int nvdr_env_shade_stage_times(nvdr_ctx *c, int backward, double *ms, int64_t *count)
{
    NVDR_REQUIRE(c && ms && count, "nvdr_env_shade_stage_times: NULL argument");
    ms[0] = ms[1] = ms[2] = 0.0;
    *count = 0;
    const int64_t n = c->prof_n < NVDR_PROF_RING ? c->prof_n : NVDR_PROF_RING;
    if (n == 0) return 0;
    NVDR_HIP_TRY(hipEventSynchronize(c->prof_ev[(c->prof_n - 1) % NVDR_PROF_RING][3]));
    // the ring holds one record per (launch, chunk); a launch whose first chunk has been overwritten is left out
    const int64_t first = c->prof_n - n;
    bool counting = false;
    for (int64_t j = first; j < c->prof_n; ++j) {
        const int i = (int)(j % NVDR_PROF_RING);
        if (c->prof_kind[i] & 2) counting = (c->prof_kind[i] & 1) == (backward ? 1 : 0);
        else if ((c->prof_kind[i] & 1) != (backward ? 1 : 0)) counting = false;
        if (!counting) continue;
        for (int k = 0; k < 3; ++k) {
            float t = 0.0f;
            NVDR_HIP_TRY(hipEventElapsedTime(&t, c->prof_ev[i][k], c->prof_ev[i][k + 1]));
            ms[k] += t;
        }
        if (c->prof_kind[i] & 2) (*count)++;
    }
    return 0;
}

extern "C" int nvdr_env_shade_fwd(nvdr_ctx *c, const nvdr_env_shade_args *a, void *stream)
{
    NvdrRange r("nvdr_env_shade_fwd");
    return env_shade_launch(c, a, false, (hipStream_t)stream);
}
extern "C" int nvdr_env_shade_bwd(nvdr_ctx *c, const nvdr_env_shade_args *a, void *stream)
{
    NvdrRange r("nvdr_env_shade_bwd");
    return env_shade_launch(c, a, true, (hipStream_t)stream);
}
extern "C" int nvdr_env_shade_stream_id(nvdr_ctx *c, uint64_t *out)
{
    NVDR_REQUIRE(c && out, "nvdr_env_shade_stream_id: NULL argument");
    *out = c->stream_id;
    return 0;
}
extern "C" int nvdr_env_shade_last_pixel_count(nvdr_ctx *c, int64_t *out, void *stream_)
{
    NVDR_REQUIRE(c && out, "nvdr_env_shade_last_pixel_count: NULL argument");
    unsigned v = 0;
    hipStream_t stream = (hipStream_t)stream_;
    NVDR_HIP_TRY(hipMemcpyAsync(&v, &c->dinfo->pix_count, sizeof(v), hipMemcpyDeviceToHost, stream));
    NVDR_HIP_TRY(hipStreamSynchronize(stream));
    *out = (int64_t)v;
    return 0;
}
