// light.hip -- EnvironmentLight.update_pdf fused into three small launches.
//
// Replaces the ~10 small torch kernels of render/light.py:46-59 that the reference runs every
// training iteration (train.py:422): pdf = max_c(base) * sin(pi*(y+0.5)/H), normalised to sum 1;
// cols = per-row inclusive prefix sum normalised by the row total; rows = prefix sum of the row
// totals normalised by the grand total.  One workgroup per probe row does the row scan in LDS;
// every row block then rescales with the grand total; a single workgroup finally scans the H row totals.
#include "common.h"

#define LT_THREADS 256

__device__ __forceinline__ float block_scan_inclusive(float v, float *wsum, float &total)
{
    // wave-level inclusive scan, then scan of the wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    float off = 0.0f, tot = 0.0f;
    for (int k = 0; k < LT_THREADS / 64; ++k) {
        if (k < wave) off += wsum[k];
        tot += wsum[k];
    }
    __syncthreads();
    total = tot;
    return v + off;
}

// one block per row: unnormalised pdf row, its inclusive scan (cols, unnormalised), row total
__global__ void __launch_bounds__(LT_THREADS) light_rows_kernel(const float *__restrict__ base, int H, int W,
                                                                 float *__restrict__ pdf, float *__restrict__ cols,
                                                                 float *__restrict__ row_tot)
{
    __shared__ float wsum[LT_THREADS / 64];
    const int y = blockIdx.x;
    const float Y = ((float)y + 0.5f) / (float)H;
    const float s = sinf(Y * 3.14159265358979323846f);
    float carry = 0.0f;
    for (int x0 = 0; x0 < W; x0 += LT_THREADS) {
        const int x = x0 + threadIdx.x;
        float v = 0.0f;
        if (x < W) {
            const float *b = base + ((int64_t)y * W + x) * 3;
            v = fmaxf(b[0], fmaxf(b[1], b[2])) * s;
            pdf[(int64_t)y * W + x] = v;
        }
        float tot;
        const float inc = block_scan_inclusive(v, wsum, tot);
        if (x < W) cols[(int64_t)y * W + x] = carry + inc;
        carry += tot;
    }
    if (threadIdx.x == 0) row_tot[y] = carry;
}

// per-row blocks: every block re-derives the grand total from the H row totals (read-only), then
// normalises its own pdf / cols row.  No block touches another block's row, so no ordering is needed.
__global__ void __launch_bounds__(LT_THREADS) light_normalise_kernel(int H, int W, const float *__restrict__ row_tot,
                                                                      float *__restrict__ pdf, float *__restrict__ cols)
{
    __shared__ float wsum[LT_THREADS / 64];
    float acc = 0.0f;
    for (int y = threadIdx.x; y < H; y += LT_THREADS) acc += row_tot[y];
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    float total = 0.0f;
    for (int k = 0; k < LT_THREADS / 64; ++k) total += wsum[k];
    const int y = blockIdx.x;
    const float rt = row_tot[y] / total; // the row total of the NORMALISED pdf, as light.py:55-58 sees it
    const float den = rt > 0.0f ? rt : 1.0f;
    for (int x = threadIdx.x; x < W; x += LT_THREADS) {
        const int64_t i = (int64_t)y * W + x;
        pdf[i] = pdf[i] / total;
        cols[i] = (cols[i] / total) / den;
    }
}

// single block, last: turn the row totals (held in `rows`) into the normalised row CDF in place
__global__ void __launch_bounds__(LT_THREADS) light_rows_cdf_kernel(int H, float *rows)
{
    __shared__ float wsum[LT_THREADS / 64];
    float carry = 0.0f;
    for (int y0 = 0; y0 < H; y0 += LT_THREADS) {
        const int y = y0 + threadIdx.x;
        const float v = y < H ? rows[y] : 0.0f;
        float tot;
        const float inc = block_scan_inclusive(v, wsum, tot);
        if (y < H) rows[y] = carry + inc;
        carry += tot;
    }
    __syncthreads();
    const float den = carry > 0.0f ? carry : 1.0f;
    for (int y = threadIdx.x; y < H; y += LT_THREADS) rows[y] = rows[y] / den;
}

extern "C" int nvdr_light_update_pdf(const float *base, int64_t hl, int64_t wl, float *pdf, float *cols, float *rows,
                                     void *stream_)
{
    NvdrRange range("nvdr_light_update_pdf");
    NVDR_REQUIRE(base && pdf && cols && rows, "light_update_pdf: NULL argument");
    NVDR_REQUIRE(hl > 0 && wl > 0 && hl < (1 << 20) && wl < (1 << 20), "light_update_pdf: bad probe size");
    hipStream_t stream = (hipStream_t)stream_;
    light_rows_kernel<<<(unsigned)hl, LT_THREADS, 0, stream>>>(base, (int)hl, (int)wl, pdf, cols, rows);
    light_normalise_kernel<<<(unsigned)hl, LT_THREADS, 0, stream>>>((int)hl, (int)wl, rows, pdf, cols);
    light_rows_cdf_kernel<<<1, LT_THREADS, 0, stream>>>((int)hl, rows);
    NVDR_LAUNCH_CHECK();
    return 0;
}
