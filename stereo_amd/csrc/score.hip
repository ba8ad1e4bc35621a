// Shi-Tomasi scoring, non-maximum suppression and score ordering for gfx950: the part of
// ScoreAndClusterFeatures (stereo/Features.cpp:672-786) that is the reference's own
// arithmetic -- the scoring loop (:704-746), the O(n^2) suppression (:752-778, "the most
// time consuming part of the feature extraction", :45-46) and the final sort (:783).  The
// CV_8U gradient images are inputs: the reference makes them with OpenCV's GaussianBlur and
// Sobel (:684-690), which stay with the caller.
//
// Every floating-point statement of the scoring loop is kept as its own rounded fp32
// operation in the reference's order (this file is compiled with -ffp-contract=off; sqrtf is
// the correctly rounded v_sqrt_f32 sequence), so scores are bit-identical to the oracle's.
//
// Pipeline (all on the context's stream, no host synchronisation; counts stay on the device):
//   score_kernel      one lane per feature: score, keep = score > threshold, kept per tile
//   compact_kernel    order-preserving compaction of the kept features (tile prefix by a
//                     short loop over the tile counts, in-tile prefix by ballots)
//   nms_kernel        kept x kept pair space in 256 x 256 tiles over a 2-D grid; a lane that
//                     finds a strictly better feature inside the window sets suppressed[a]
//   nms_count_kernel  flags + kept-per-tile counts
//   compact_kernel    survivors, still in input order
//   rank_kernel       same tiling: partial rank = number of survivors with a larger score
//                     (ties: earlier input position first), summed with integer atomics
//   rank_scatter      output[rank] = survivor -- a deterministic sort
#include "stx_internal.h"

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_SLICES = 64;  // b-tile slices of the O(n^2) kernels (grid.y)

struct Kernel3x3 {
    float w[9];
};

// Features.cpp:704-740 for one feature; NaN if the 3x3 window leaves the image
__device__ __forceinline__ float shi_tomasi(const uint8_t *__restrict__ gx, const uint8_t *__restrict__ gy,
                                            int w, int h, size_t stride, float px, float py,
                                            const Kernel3x3 &k)
{
    const int cy = (int)py, cx = (int)px;  // :718-719
    if (cx < 1 || cy < 1 || cx > w - 2 || cy > h - 2) return __builtin_nanf("");
    float m00 = 0.f, m01 = 0.f, m10 = 0.f, m11 = 0.f;
#pragma unroll
    for (int n = -1; n <= 1; ++n) {
#pragma unroll
        for (int m = -1; m <= 1; ++m) {
            const size_t at = (size_t)(cy + n) * stride + (size_t)(cx + m);
            const float wgt = k.w[(n + 1) * 3 + (m + 1)];  // :720
            const int vx = gx[at], vy = gy[at];
            m00 = m00 + wgt * (float)(vx * vx);  // :721
            m01 = m01 + wgt * (float)(vx * vy);  // :722
            m10 = m10 + wgt * (float)(vx * vy);  // :723
            m11 = m11 + wgt * (float)(vy * vy);  // :724
        }
    }
    const float a = 1.f;
    const float b = -1.f * (m00 + m11);              // :733
    const float c = m00 * m11 - m10 * m01;           // :734
    const float root = sqrtf(b * b - 4.f * a * c);
    const float eigen1 = (-b + root) / 2.f * a;      // :735
    const float eigen2 = (-b - root) / 2.f * a;      // :736
    return eigen2 < eigen1 ? eigen2 : eigen1;        // std::min, :738
}

// kept-per-tile counts: one int per block of SC_THREADS items
__device__ __forceinline__ void tile_count(bool keep, int *__restrict__ tile_counts)
{
    __shared__ int wave_counts[SC_THREADS / 64];
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    if ((threadIdx.x & 63) == 0) wave_counts[threadIdx.x >> 6] = __builtin_popcountll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < SC_THREADS / 64; ++i) s += wave_counts[i];
        tile_counts[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(SC_THREADS) void score_kernel(
    const uint8_t *__restrict__ gx, const uint8_t *__restrict__ gy, int w, int h, size_t stride,
    const float2 *__restrict__ xy, int n, Kernel3x3 k, float threshold, float *__restrict__ score,
    uint8_t *__restrict__ keep, int *__restrict__ tile_counts)
{
    const int i = blockIdx.x * SC_THREADS + threadIdx.x;
    bool kp = false;
    if (i < n) {
        const float2 p = xy[i];
        const float s = shi_tomasi(gx, gy, w, h, stride, p.x, p.y, k);
        score[i] = s;
        kp = s > threshold;  // :742 (false for NaN)
        keep[i] = kp;
    }
    tile_count(kp, tile_counts);
}

// Order-preserving compaction of items [0, *n_dev or n) with keep[i] set: out_index[pos] = the
// item's index in the ORIGINAL feature list (src_index[i], or i when src_index is null).
__global__ __launch_bounds__(SC_THREADS) void compact_kernel(
    const uint8_t *__restrict__ keep, const int *__restrict__ tile_counts, const int *__restrict__ src_index,
    int n, const int *__restrict__ n_dev, int *__restrict__ out_index, int *__restrict__ out_count)
{
    __shared__ int wave_base[SC_THREADS / 64 + 1];
    __shared__ int tile_base;
    const int nn = n_dev ? *n_dev : n;
    const int tiles = (nn + SC_THREADS - 1) / SC_THREADS;
    if (tiles == 0 && blockIdx.x == 0 && threadIdx.x == 0) *out_count = 0;
    if ((int)blockIdx.x >= tiles) return;
    const int i = blockIdx.x * SC_THREADS + threadIdx.x;
    const bool kp = i < nn && keep[i];
    // tiles before this one
    int part = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += SC_THREADS) part += tile_counts[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(kp);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) tile_base = 0;
    __syncthreads();
    if (lane == 0) {
        atomicAdd(&tile_base, part);  // integer sum: order does not matter
        wave_base[wave] = __builtin_popcountll(bal);
    }
    __syncthreads();
    int before = tile_base;
    for (int v = 0; v < wave; ++v) before += wave_base[v];
    if (kp) {
        const int pos = before + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        out_index[pos] = src_index ? src_index[i] : i;
    }
    if ((int)blockIdx.x == tiles - 1 && threadIdx.x == SC_THREADS - 1) {
        int total = tile_base;
        for (int v = 0; v < SC_THREADS / 64; ++v) total += wave_base[v];
        *out_count = total;
    }
}

// Features.cpp:752-778 over the kept features good[0 .. *ngood): feature a survives unless a
// kept feature b != a lies within `dist` on both axes and has a strictly larger score.
// The O(n^2) pair space is cut into tiles of SC_THREADS x SC_THREADS: block (x, y) holds the
// a-tile x and walks the b-tiles y, y + gridDim.y, ...; a lane that finds a better neighbour
// sets suppressed[a] (every writer stores the same 1: no ordering issue).
__global__ __launch_bounds__(SC_THREADS) void nms_kernel(
    const float2 *__restrict__ xy, const float *__restrict__ score, const int *__restrict__ good, int n,
    const int *__restrict__ ngood, float dist, uint8_t *__restrict__ suppressed)
{
    // good == null: the kept list is the whole input list (ClusterFeatures); ngood == null: n
    __shared__ float sx[SC_THREADS], sy[SC_THREADS], ss[SC_THREADS];
    const int ng = ngood ? *ngood : n;
    const int tiles = (ng + SC_THREADS - 1) / SC_THREADS;
    if ((int)blockIdx.x >= tiles || (int)blockIdx.y >= tiles) return;
    const int a = blockIdx.x * SC_THREADS + threadIdx.x;
    float ax = 0.f, ay = 0.f, as = 0.f;
    if (a < ng) {
        const int f = good ? good[a] : a;
        const float2 p = xy[f];
        ax = p.x; ay = p.y; as = score[f];
    }
    bool is_max = true;
    for (int tb = blockIdx.y; tb < tiles; tb += gridDim.y) {
        const int t0 = tb * SC_THREADS;
        const int b = t0 + threadIdx.x;
        if (b < ng) {
            const int f2 = good ? good[b] : b;
            const float2 p = xy[f2];
            sx[threadIdx.x] = p.x; sy[threadIdx.x] = p.y; ss[threadIdx.x] = score[f2];
        }
        __syncthreads();
        const int cnt = min(SC_THREADS, ng - t0);
        for (int j = 0; j < cnt; ++j) {
            // (b == a is harmless: a score is never < itself)
            const int xmargin = (int)fabsf(ax - sx[j]);  // :762
            const int ymargin = (int)fabsf(ay - sy[j]);  // :763
            if ((float)xmargin <= dist && (float)ymargin <= dist && as < ss[j]) is_max = false;  // :764-766
        }
        __syncthreads();
    }
    if (a < ng && !is_max) suppressed[a] = 1;
}

// suppressed -> keep flags (in place) and kept-per-tile counts
__global__ __launch_bounds__(SC_THREADS) void nms_count_kernel(uint8_t *__restrict__ flags, int n,
                                                               const int *__restrict__ ngood,
                                                               int *__restrict__ tile_counts)
{
    const int ng = ngood ? *ngood : n;
    const int tiles = (ng + SC_THREADS - 1) / SC_THREADS;
    if ((int)blockIdx.x >= tiles) return;
    const int a = blockIdx.x * SC_THREADS + threadIdx.x;
    const bool kp = a < ng && !flags[a];
    if (a < ng) flags[a] = kp;
    tile_count(kp, tile_counts);
}

// Features.cpp:783: sort by score descending; equal scores keep their input order.  Partial
// ranks over the same tiling of the pair space, summed with integer atomics.
__global__ __launch_bounds__(SC_THREADS) void rank_kernel(
    const float *__restrict__ score, const int *__restrict__ kept, const int *__restrict__ nkept,
    int *__restrict__ rank_out)
{
    __shared__ float ss[SC_THREADS];
    const int nk = *nkept;
    const int tiles = (nk + SC_THREADS - 1) / SC_THREADS;
    if ((int)blockIdx.x >= tiles || (int)blockIdx.y >= tiles) return;
    const int a = blockIdx.x * SC_THREADS + threadIdx.x;
    const float as = a < nk ? score[kept[a]] : 0.f;
    int rank = 0;
    for (int tb = blockIdx.y; tb < tiles; tb += gridDim.y) {
        const int t0 = tb * SC_THREADS;
        const int b = t0 + threadIdx.x;
        if (b < nk) ss[threadIdx.x] = score[kept[b]];
        __syncthreads();
        const int cnt = min(SC_THREADS, nk - t0);
        for (int j = 0; j < cnt; ++j) rank += (ss[j] > as) || (ss[j] == as && t0 + j < a);
        __syncthreads();
    }
    if (a < nk && rank) atomicAdd(rank_out + a, rank);
}

__global__ __launch_bounds__(SC_THREADS) void rank_scatter_kernel(
    const float2 *__restrict__ xy, const float *__restrict__ score, const int *__restrict__ kept,
    const int *__restrict__ nkept, const int *__restrict__ rank, float2 *__restrict__ xy_out,
    float *__restrict__ score_out, int32_t *__restrict__ index_out)
{
    const int a = blockIdx.x * SC_THREADS + threadIdx.x;
    if (a >= *nkept) return;
    const int f = kept[a], r = rank[a];
    xy_out[r] = xy[f];
    score_out[r] = score[f];
    index_out[r] = f;
}

}  // namespace

int stx_launch_score_cluster(stx_ctx *ctx, const uint8_t *d_gx, const uint8_t *d_gy, int w, int h,
                             size_t stride, const float *d_xy, int n, const float kernel3x3[9],
                             float score_threshold, float cluster_dist, float *d_xy_out,
                             float *d_score_out, int32_t *d_index_out, int *d_count)
{
    hipStream_t st = ctx->stream;
    if (n <= 0) return (int)hipMemsetAsync(d_count, 0, sizeof(int), st);
    const int tiles = (n + SC_THREADS - 1) / SC_THREADS;
    int err;
    // [score n][good n][kept n][rank n][tile_counts tiles][counts 4][flags n bytes]
    const size_t words = (size_t)4 * n + tiles + 4;
    int *ws = (int *)stx_workspace(ctx, STX_SLOT_SCORE, sizeof(int) * words + (size_t)n, &err);
    if (!ws) return err;
    float *score = reinterpret_cast<float *>(ws);
    int *good = ws + n;
    int *kept = good + n;
    int *rank = kept + n;
    int *tile_counts = rank + n;
    int *counts = tile_counts + tiles;  // [0] = ngood
    uint8_t *flags = reinterpret_cast<uint8_t *>(counts + 4);
    Kernel3x3 k;
    for (int i = 0; i < 9; ++i) k.w[i] = kernel3x3[i];
    const float2 *xy = reinterpret_cast<const float2 *>(d_xy);
    const dim3 pairs(tiles, tiles < SC_SLICES ? tiles : SC_SLICES);
    hipLaunchKernelGGL(score_kernel, dim3(tiles), dim3(SC_THREADS), 0, st, d_gx, d_gy, w, h, stride, xy, n, k,
                       score_threshold, score, flags, tile_counts);
    hipLaunchKernelGGL(compact_kernel, dim3(tiles), dim3(SC_THREADS), 0, st, flags, tile_counts,
                       (const int *)nullptr, n, (const int *)nullptr, good, counts);
    STX_HIP(hipMemsetAsync(flags, 0, (size_t)n, st));
    STX_HIP(hipMemsetAsync(rank, 0, sizeof(int) * (size_t)n, st));
    hipLaunchKernelGGL(nms_kernel, pairs, dim3(SC_THREADS), 0, st, xy, score, good, n, counts, cluster_dist, flags);
    hipLaunchKernelGGL(nms_count_kernel, dim3(tiles), dim3(SC_THREADS), 0, st, flags, n, counts, tile_counts);
    hipLaunchKernelGGL(compact_kernel, dim3(tiles), dim3(SC_THREADS), 0, st, flags, tile_counts, good, n, counts,
                       kept, d_count);
    hipLaunchKernelGGL(rank_kernel, pairs, dim3(SC_THREADS), 0, st, score, kept, d_count, rank);
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(tiles), dim3(SC_THREADS), 0, st, xy, score, kept, d_count, rank,
                       reinterpret_cast<float2 *>(d_xy_out), d_score_out, d_index_out);
    return (int)hipGetLastError();
}

// ClusterFeatures (stereo/Features.cpp:48-80): the same suppression over a list that already
// carries scores; survivors keep their input order.
int stx_launch_cluster(stx_ctx *ctx, const float *d_xy, const float *d_score, int n, float window,
                       int32_t *d_index_out, int *d_count)
{
    hipStream_t st = ctx->stream;
    if (n <= 0) return (int)hipMemsetAsync(d_count, 0, sizeof(int), st);
    const int tiles = (n + SC_THREADS - 1) / SC_THREADS;
    int err;
    int *tile_counts = (int *)stx_workspace(ctx, STX_SLOT_SCORE, sizeof(int) * (size_t)tiles + (size_t)n, &err);
    if (!tile_counts) return err;
    uint8_t *keep = reinterpret_cast<uint8_t *>(tile_counts + tiles);
    STX_HIP(hipMemsetAsync(keep, 0, (size_t)n, st));
    hipLaunchKernelGGL(nms_kernel, dim3(tiles, tiles < SC_SLICES ? tiles : SC_SLICES), dim3(SC_THREADS), 0, st,
                       reinterpret_cast<const float2 *>(d_xy), d_score, (const int *)nullptr, n,
                       (const int *)nullptr, window, keep);
    hipLaunchKernelGGL(nms_count_kernel, dim3(tiles), dim3(SC_THREADS), 0, st, keep, n, (const int *)nullptr,
                       tile_counts);
    hipLaunchKernelGGL(compact_kernel, dim3(tiles), dim3(SC_THREADS), 0, st, keep, tile_counts,
                       (const int *)nullptr, n, (const int *)nullptr, d_index_out, d_count);
    return (int)hipGetLastError();
}
