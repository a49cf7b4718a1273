// optim.hip -- the parameter update of one training iteration as ONE launch.
//
// Replaces, for the three tensors the direct-lighting iteration trains (kd texture, ks, light probe), the sequence
//     lgt.base.grad *= 64                      train.py:439-440
//     optimizer.step()                         train.py:452-461  (torch.optim.Adam, no weight decay, no amsgrad)
//     clamps of the parameters                 train.py:470-476 / material clamps
// which on ROCm is a multi-tensor Adam launch that puts ~1 M elements on 16 workgroups (45 us on a 256-CU part) plus five
// small elementwise kernels.  One thread per element over all tensors, the step counter in device memory (the launch can be
// captured in a HIP graph and replayed), same arithmetic and the same order of operations as torch's single-tensor Adam:
//     m = m + (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g
//     p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps);  p = max(min(p, hi), lo)
#include "common.h"
#include "nvdr_hip.h"

#define ADAM_SPARSE_TILES 4      // tiles of 64 texels per wavefront of the sparse-texture path
struct AdamTable {
    nvdr_adam_tensor t[NVDR_ADAM_MAX_TENSORS];
    int first_block[NVDR_ADAM_MAX_TENSORS + 1];     // workgroups [first_block[k], first_block[k + 1]) work on tensor k
    int per_thread;                                 // elements per thread (a workgroup covers 256 * per_thread of them)
    int n;
};

__global__ void __launch_bounds__(256) adam_step_kernel(AdamTable tab, double lr, double beta1d, double beta2d, float eps, const int *state)
{
    // state[0] = steps taken so far, state[1] unused, then two doubles: beta1^step, beta2^step (kept as running products: pow() in
    // double costs a workgroup ~10 us of latency); read-only here, advanced by adam_advance_kernel behind this launch
    __shared__ float corr[2];
    // the hyper-parameters arrive as doubles, like the Python floats torch's Adam computes 1 - beta from: (float)(1 - 0.999) is
    // 0.001f, 1 - 0.999f is 1.3e-5 off (visible in the parameters after ten steps)
    const float beta2 = (float)beta2d, omb1 = (float)(1.0 - beta1d), omb2 = (float)(1.0 - beta2d);
    const int step = state[0] + 1;
    if (threadIdx.x == 0) {
        const double *pw = (const double *)(state + 2);
        const double b1p = (step == 1 ? 1.0 : pw[0]) * beta1d, b2p = (step == 1 ? 1.0 : pw[1]) * beta2d;
        corr[0] = (float)(lr / (1.0 - b1p));        // step_size = lr / bias_correction1
        corr[1] = (float)sqrt(1.0 - b2p);                   // sqrt(bias_correction2)
    }
    __syncthreads();
    const float step_size0 = corr[0], bc2_sqrt = corr[1];
    // the tensor of this workgroup (uniform: the table entry is read with scalar loads)
    int k = 0;
    for (int q = 1; q < tab.n; ++q)
        if ((int)blockIdx.x >= tab.first_block[q]) k = q;
    const nvdr_adam_tensor &T = tab.t[k];
    if (T.frozen) {
        // A frozen tensor (lr_scale 0 on the host) takes no update, but its gradient is still CONSUMED: with zero_grad the producer
        // scatter-adds into a persistent buffer that nobody else clears -- and with several ranks that buffer is the exchange bucket,
        // all-reduced in place every step: left alone its content would grow by the world size per iteration until it overflows.
        if (T.active && T.zero_grad) {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int64_t texels = T.n / 3, tiles = (texels + 63) / 64;
            const int64_t t0 = ((int64_t)((int)blockIdx.x - tab.first_block[k]) * 4 + wave) * ADAM_SPARSE_TILES;
            float *gw = const_cast<float *>(T.grad);
            for (int j0 = 0; j0 < ADAM_SPARSE_TILES; ++j0) {
                const int64_t tile = t0 + j0, t = tile * 64 + lane;
                if (tile >= tiles) break;
                const bool ok = t < texels;
                const float g0 = ok ? T.grad[3 * t] : 0.0f, g1 = ok ? T.grad[3 * t + 1] : 0.0f, g2 = ok ? T.grad[3 * t + 2] : 0.0f;
                // (a NaN compares unequal to zero: a poisoned tile is cleared as well)
                if (__ballot(g0 != 0.0f || g1 != 0.0f || g2 != 0.0f) != 0ull && ok) { gw[3 * t] = 0.0f; gw[3 * t + 1] = 0.0f; gw[3 * t + 2] = 0.0f; }
            }
        }
        return;
    }
    const float step_size = step_size0 * (T.lr_scale == 0.0f ? 1.0f : T.lr_scale);   // (a zero-initialised block: the plain learning rate)        // this tensor's learning rate (train.py:336-338: position / material / light)
    const int64_t e0 = (int64_t)((int)blockIdx.x - tab.first_block[k]) * 256 * tab.per_thread;
    if (T.active) {
        // SPARSE TEXTURE path (round 4).  A trained texture of 1024^2 texels receives gradient only at the texels some covered pixel
        // looked up -- ~5 % of them with the fixed benchmark views -- and a texel that has never received any still has zero moments:
        // its update is p - step * 0 / (0 + eps) = p, a clamp of a value the first step already clamped.  Tiles of 64 texels (one per
        // lane) with an all-zero gradient and no history are therefore skipped after reading the gradient alone: 12 instead of 84
        // bytes per texel, EXACTLY the same parameters (step 1 visits every tile, so that the clamps / the normalisation have been
        // applied to whatever the initial values were).  `active[tile]` remembers history.  zero_grad: the gradient of a visited tile
        // is zeroed behind the update, so that a scatter-add backward (nvdr_texture_lookup_bwd into a persistent buffer) needs no
        // memset of 3 x 12.6 MB per iteration.
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int64_t texels = T.n / 3, tiles = (texels + 63) / 64;
        const int64_t t0 = ((int64_t)((int)blockIdx.x - tab.first_block[k]) * 4 + wave) * ADAM_SPARSE_TILES;
        float *gw = const_cast<float *>(T.grad);
        float g3[ADAM_SPARSE_TILES][3];
        bool act[ADAM_SPARSE_TILES];
#pragma unroll
        for (int j0 = 0; j0 < ADAM_SPARSE_TILES; ++j0) {               // the gradients (and histories) of all tiles of this wavefront first: one round trip
            const int64_t tile = t0 + j0, t = tile * 64 + lane;
            const bool ok = tile < tiles && t < texels;
            g3[j0][0] = ok ? T.grad[3 * t] : 0.0f; g3[j0][1] = ok ? T.grad[3 * t + 1] : 0.0f; g3[j0][2] = ok ? T.grad[3 * t + 2] : 0.0f;
            act[j0] = tile < tiles ? T.active[tile] != 0 : false;
        }
#pragma unroll
        for (int j0 = 0; j0 < ADAM_SPARSE_TILES; ++j0) {
            const int64_t tile = t0 + j0;
            if (tile >= tiles) break;
            const int64_t t = tile * 64 + lane;
            const bool ok = t < texels;
            const bool any = __ballot(g3[j0][0] != 0.0f || g3[j0][1] != 0.0f || g3[j0][2] != 0.0f) != 0ull;
            if (!any && !act[j0] && step > 1) continue;
            if (ok) {
                float q[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int64_t e = 3 * t + c;
                    const float gg = g3[j0][c] * T.grad_scale, m0 = T.exp_avg[e], v0 = T.exp_avg_sq[e];
                    const float mm = m0 + (gg - m0) * omb1;
                    const float vv = v0 * beta2 + (omb2 * gg) * gg;
                    const float denom = sqrtf(vv) / bc2_sqrt + eps;
                    const float p = T.param[e] - step_size * (mm / denom);
                    float lo = T.lo, hi = T.hi;
                    if (T.lo_vec) lo = fmaxf(lo, T.lo_vec[e % T.lo_vec_n]);
                    if (T.hi_vec) hi = fminf(hi, T.hi_vec[e % T.hi_vec_n]);
                    q[c] = p < lo ? lo : (p > hi ? hi : p);
                    T.exp_avg[e] = mm;
                    T.exp_avg_sq[e] = vv;
                    if (T.zero_grad && any) gw[e] = 0.0f;
                }
                if (T.normalize3) {
                    const float len = sqrtf(fmaxf((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2], 1e-20f));
                    q[0] /= len; q[1] /= len; q[2] /= len;
                }
                T.param[3 * t] = q[0]; T.param[3 * t + 1] = q[1]; T.param[3 * t + 2] = q[2];
            }
            if (any && !act[j0] && lane == 0) T.active[tile] = 1;
        }
    } else if (T.normalize3) {
        // a unit is a texel of three channels: updated, clamped per channel and then renormalised by one thread
        // (Texture2D.clamp_ + normalize_ of the normal map, train.py:470-474, render/texture.py:86-96)
        const int64_t units = T.n / 3;
        for (int j0 = 0; j0 < tab.per_thread; ++j0) {
            const int64_t t = e0 + (int64_t)j0 * 256 + threadIdx.x;
            if (t >= units) break;
            float q[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int64_t e = 3 * t + c;
                const float gg = T.grad[e] * T.grad_scale, m0 = T.exp_avg[e], v0 = T.exp_avg_sq[e];
                const float mm = m0 + (gg - m0) * omb1;
                const float vv = v0 * beta2 + (omb2 * gg) * gg;
                const float denom = sqrtf(vv) / bc2_sqrt + eps;
                float p = T.param[e] - step_size * (mm / denom);
                float lo = T.lo, hi = T.hi;
                if (T.lo_vec) lo = fmaxf(lo, T.lo_vec[e % T.lo_vec_n]);
                if (T.hi_vec) hi = fminf(hi, T.hi_vec[e % T.hi_vec_n]);
                q[c] = p < lo ? lo : (p > hi ? hi : p);             // keeps a NaN (fmaxf / fminf would turn it into a bound)
                T.exp_avg[e] = mm;
                T.exp_avg_sq[e] = vv;
            }
            const float len = sqrtf(fmaxf((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2], 1e-20f));     // util.safe_normalize
            T.param[3 * t] = q[0] / len; T.param[3 * t + 1] = q[1] / len; T.param[3 * t + 2] = q[2] / len;
        }
    } else
    for (int j0 = 0; j0 < tab.per_thread; j0 += 4) {        // four elements per round: their twelve loads are in flight together
        float g[4], m[4], v[4], pp[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t e = e0 + (int64_t)(j0 + u) * 256 + threadIdx.x;
            ok[u] = j0 + u < tab.per_thread && e < T.n;
            const int64_t ee = ok[u] ? e : 0;
            g[u] = T.grad[ee]; m[u] = T.exp_avg[ee]; v[u] = T.exp_avg_sq[ee]; pp[u] = T.param[ee];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            const int64_t e = e0 + (int64_t)(j0 + u) * 256 + threadIdx.x;
            const float gg = g[u] * T.grad_scale;
            const float mm = m[u] + (gg - m[u]) * omb1;                 // lerp_(grad, 1 - beta1)
            const float vv = v[u] * beta2 + (omb2 * gg) * gg;           // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
            const float denom = sqrtf(vv) / bc2_sqrt + eps;
            float p = pp[u] - step_size * (mm / denom);                 // addcdiv_(exp_avg, denom, value = -step_size)
            float lo = T.lo, hi = T.hi;
            if (T.lo_vec) lo = fmaxf(lo, T.lo_vec[e % T.lo_vec_n]);
            if (T.hi_vec) hi = fminf(hi, T.hi_vec[e % T.hi_vec_n]);
            p = p < lo ? lo : (p > hi ? hi : p);                        // torch.clamp semantics: a NaN parameter stays NaN
            T.exp_avg[e] = mm;
            T.exp_avg_sq[e] = vv;
            T.param[e] = p;
        }
    }
}

// The step counter and the running powers move in a launch of their own (one thread, ~4 us) behind the update: every workgroup of the
// update has read them by then (stream order).  Rounds 2-3 let the workgroups of the update draw tickets on ONE word and the last
// one publish the state: up to 2 048 same-address atomics, which retire one every ~12 ns -- 25 us of a 33 us kernel.
__global__ void adam_advance_kernel(int *state, double beta1d, double beta2d)
{
    const int step = state[0] + 1;
    double *pw = (double *)(state + 2);
    pw[0] = (step == 1 ? 1.0 : pw[0]) * beta1d;
    pw[1] = (step == 1 ? 1.0 : pw[1]) * beta2d;
    state[0] = step;
}

extern "C" int nvdr_adam_step(const nvdr_adam_tensor *tensors, int n_tensors, double lr, double beta1, double beta2, double eps,
                              int *state, void *stream_)
{
    return nvdr_adam_step_partial(tensors, n_tensors, lr, beta1, beta2, eps, state, 1, stream_);
}

extern "C" int nvdr_adam_step_partial(const nvdr_adam_tensor *tensors, int n_tensors, double lr, double beta1, double beta2, double eps,
                                      int *state, int advance, void *stream_)
{
    NvdrRange range("nvdr_adam_step");
    NVDR_REQUIRE(tensors && state, "adam_step: NULL argument");
    NVDR_REQUIRE(n_tensors >= 1 && n_tensors <= NVDR_ADAM_MAX_TENSORS, "adam_step: %d tensors (1..%d supported)", n_tensors, NVDR_ADAM_MAX_TENSORS);
    NVDR_REQUIRE(lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "adam_step: bad hyper-parameters");
    AdamTable tab;
    memset(&tab, 0, sizeof(tab));
    tab.n = n_tensors;
    int64_t total = 0;
    for (int k = 0; k < n_tensors; ++k) {
        const nvdr_adam_tensor &t = tensors[k];
        NVDR_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq && t.n >= 0, "adam_step: tensor %d has a NULL buffer", k);
        NVDR_REQUIRE(!t.lo_vec || t.lo_vec_n > 0, "adam_step: tensor %d: lo_vec without length", k);
        NVDR_REQUIRE(!t.hi_vec || t.hi_vec_n > 0, "adam_step: tensor %d: hi_vec without length", k);
        NVDR_REQUIRE(!t.normalize3 || t.n % 3 == 0, "adam_step: tensor %d: normalize3 needs a multiple of three elements", k);
        NVDR_REQUIRE(t.lr_scale >= 0.0f, "adam_step: tensor %d: negative lr_scale (0 = the plain rate; `frozen` freezes)", k);
        NVDR_REQUIRE(!t.active || t.n % 3 == 0, "adam_step: tensor %d: the sparse path works on texels of three channels", k);
        NVDR_REQUIRE(!t.zero_grad || t.active, "adam_step: tensor %d: zero_grad needs the sparse path (active)", k);
        tab.t[k] = t;
        total += t.n;
    }
    NVDR_REQUIRE(total < (1ll << 40), "adam_step: too many elements");
    const int64_t per_thread = 4;        // four elements (texels of a normalised tensor) per thread
    tab.per_thread = (int)per_thread;
    int64_t blocks = 0;
    for (int k = 0; k < n_tensors; ++k) {
        tab.first_block[k] = (int)blocks;
        blocks += tensors[k].active ? div_up((tensors[k].n / 3 + 63) / 64, 4 * ADAM_SPARSE_TILES)
                                    : div_up(tensors[k].normalize3 ? tensors[k].n / 3 : tensors[k].n, 256 * per_thread);
    }
    for (int k = n_tensors; k <= NVDR_ADAM_MAX_TENSORS; ++k) tab.first_block[k] = (int)blocks;
    if (blocks < 1) blocks = 1;
    adam_step_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(tab, lr, beta1, beta2, (float)eps, state);
    if (advance) adam_advance_kernel<<<1, 1, 0, (hipStream_t)stream_>>>(state, beta1, beta2);
    NVDR_LAUNCH_CHECK();
    return 0;
}
