// exchange.hip -- device side of the tile-sparse gradient exchange (nvdiffrecmc_amd/parallel.py GradientExchange).
//
// One view per GPU: the gradient of a trained texture that is looked up at the nearest texel is zero outside the texels this rank's
// covered pixels touched.  Instead of all-reducing the dense 37.7 MB of texture gradients, the ranks
//     1. flag the tiles (64 texels x 3 channels = 192 floats = 768 contiguous bytes) of their bucket that hold a non-zero value,
//     2. all-reduce (MAX) the flag bytes -- 49 KB for three 1024^2 textures -- so every rank knows the UNION of the touched tiles,
//     3. list the union in ascending tile order (the same list on every rank, a function of the flags alone),
//     4. gather those tiles into a compact buffer, all-reduce (SUM) it, and scatter the sums back into the dense bucket,
// where the optimizer reads them exactly as it reads a densely reduced bucket (an untouched tile is zero on every rank, and so is its
// sum).  The sums are the dense exchange's sums bit for bit: the same addends meet in the same collective, only zeros stay home.
// The reference has no distributed code; this is plumbing around an RCCL collective, HBM-bound byte moving.
#include "common.h"
#include "nvdr_hip.h"

// flags[t] = 1 if tile t of `grad` holds a value != 0 (a NaN counts: it must travel), else 0.  One wavefront per NVDR_FLAG_TILES tiles,
// float4 loads: 37.7 MB are read once (the bucket is mostly zeros, but only reading it tells).
#define NVDR_FLAG_TILES 4
__global__ void __launch_bounds__(256) tile_flags_kernel(const float4 *__restrict__ grad, int64_t n_tiles, int tile_vec4, uint8_t *__restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t t0 = wave * NVDR_FLAG_TILES;
#pragma unroll
    for (int j = 0; j < NVDR_FLAG_TILES; ++j) {
        const int64_t t = t0 + j;
        if (t >= n_tiles) break;
        bool nz = false;
        for (int e = lane; e < tile_vec4; e += 64) {
            const float4 v = grad[t * tile_vec4 + e];
            nz |= (v.x != 0.0f) | (v.y != 0.0f) | (v.z != 0.0f) | (v.w != 0.0f);
        }
        const bool any = __ballot(nz) != 0ull;
        if (lane == 0) flags[t] = any ? 1 : 0;
    }
}

// list[0 .. count) = the indices of the flagged tiles in ascending order; ONE workgroup (the list must be the same on every rank and
// is at most a few hundred thousand entries: a thread takes a contiguous run of flags, the runs' counts are scanned through LDS).
__global__ void __launch_bounds__(1024) tile_plan_kernel(const uint8_t *__restrict__ flags, int64_t n_tiles, int32_t *__restrict__ list, int32_t *__restrict__ count)
{
    __shared__ int wave_sum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (n_tiles + 1023) / 1024;
    const int64_t b = (int64_t)tid * per, e = min(b + per, n_tiles);
    int mine = 0;
    for (int64_t t = b; t < e; ++t) mine += flags[t] != 0;
    int incl = mine;                                    // inclusive scan inside the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) base += wave_sum[w];
        total += wave_sum[w];
    }
    int pos = base + incl - mine;
    for (int64_t t = b; t < e; ++t)
        if (flags[t] != 0) list[pos++] = (int32_t)t;
    if (tid == 0) *count = total;
}

// compact[i] = dense[list[i]] (GATHER) or dense[list[i]] = compact[i] (scatter), tiles of tile_vec4 float4; the grid is sized for
// every tile, the count is read on the device (the host learns it only to size the collective).
template <bool GATHER>
__global__ void __launch_bounds__(256) tile_move_kernel(float4 *__restrict__ dense, float4 *__restrict__ compact, const int32_t *__restrict__ list,
                                                        const int32_t *__restrict__ count, int tile_vec4)
{
    const int64_t total = (int64_t)*count * tile_vec4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = i / tile_vec4;
        const int within = (int)(i - slot * tile_vec4);
        const int64_t d = (int64_t)list[slot] * tile_vec4 + within;
        if (GATHER) compact[i] = dense[d];
        else dense[d] = compact[i];
    }
}

static int check_tiles(const void *p, int64_t n_tiles, int tile_floats, const char *who)
{
    NVDR_REQUIRE(p, "%s: NULL buffer", who);
    NVDR_REQUIRE(n_tiles >= 0 && n_tiles < (1ll << 31), "%s: %lld tiles out of range", who, (long long)n_tiles);
    NVDR_REQUIRE(tile_floats >= 4 && tile_floats % 4 == 0, "%s: tile_floats %d must be a positive multiple of 4", who, tile_floats);
    NVDR_REQUIRE(((uintptr_t)p & 15u) == 0, "%s: buffer must be 16-byte aligned", who);
    return 0;
}

extern "C" int nvdr_tile_flags(const float *grad, int64_t n_tiles, int tile_floats, uint8_t *flags, void *stream)
{
    if (int r = check_tiles(grad, n_tiles, tile_floats, "nvdr_tile_flags")) return r;
    NVDR_REQUIRE(flags, "nvdr_tile_flags: NULL flags");
    if (n_tiles == 0) return 0;
    tile_flags_kernel<<<div_up(n_tiles, 4 * NVDR_FLAG_TILES), 256, 0, (hipStream_t)stream>>>((const float4 *)grad, n_tiles, tile_floats / 4, flags);
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_tile_plan(const uint8_t *flags, int64_t n_tiles, int32_t *list, int32_t *count, void *stream)
{
    NVDR_REQUIRE(flags && list && count, "nvdr_tile_plan: NULL argument");
    NVDR_REQUIRE(n_tiles >= 0 && n_tiles < (1ll << 31), "nvdr_tile_plan: %lld tiles out of range", (long long)n_tiles);
    tile_plan_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(flags, n_tiles, list, count);
    NVDR_LAUNCH_CHECK();
    return 0;
}

static unsigned move_grid(int64_t n_tiles, int tile_floats)
{
    const int64_t blocks = ((int64_t)n_tiles * (tile_floats / 4) + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

extern "C" int nvdr_tile_gather(const float *dense, const int32_t *list, const int32_t *count, int64_t n_tiles, int tile_floats, float *compact, void *stream)
{
    if (int r = check_tiles(dense, n_tiles, tile_floats, "nvdr_tile_gather")) return r;
    if (int r = check_tiles(compact, n_tiles, tile_floats, "nvdr_tile_gather")) return r;
    NVDR_REQUIRE(list && count, "nvdr_tile_gather: NULL list");
    if (n_tiles == 0) return 0;
    tile_move_kernel<true><<<move_grid(n_tiles, tile_floats), 256, 0, (hipStream_t)stream>>>((float4 *)const_cast<float *>(dense), (float4 *)compact, list, count, tile_floats / 4);
    NVDR_LAUNCH_CHECK();
    return 0;
}

extern "C" int nvdr_tile_scatter(const float *compact, const int32_t *list, const int32_t *count, int64_t n_tiles, int tile_floats, float *dense, void *stream)
{
    if (int r = check_tiles(dense, n_tiles, tile_floats, "nvdr_tile_scatter")) return r;
    if (int r = check_tiles(compact, n_tiles, tile_floats, "nvdr_tile_scatter")) return r;
    NVDR_REQUIRE(list && count, "nvdr_tile_scatter: NULL list");
    if (n_tiles == 0) return 0;
    tile_move_kernel<false><<<move_grid(n_tiles, tile_floats), 256, 0, (hipStream_t)stream>>>((float4 *)dense, (float4 *)const_cast<float *>(compact), list, count, tile_floats / 4);
    NVDR_LAUNCH_CHECK();
    return 0;
}
