// FAST-12 corner detection for gfx950: FindFASTFeatures (stereo/Features.cpp:407-471) with
// CheckForSequential12 (:534-602) and ThreeOfFourValuesBrighterOrDarker (:477-517).
//
// The reference's scan accepts a pixel iff its 16-point ring holds a circular run of at
// least 12 points that are all > p + t or all < p - t (the three-of-four pretest never
// rejects such a ring; both facts are checked exhaustively over all 3^16 patterns by
// oracle_fast_selfcheck).  The kernel evaluates exactly that with two 16-bit masks and
// rotate-and steps.  The ring order is the reference's (:429-453).
//
// Output order is the reference's raster order (rows outer, columns inner):
//   fast_flags_kernel    one lane per pixel, wave ballot -> one 64-bit word per 64 pixels
//   fast_scan_kernel     exclusive prefix sum of the words' popcounts (one workgroup)
//   fast_scatter_kernel  one lane per word writes its corners at prefix + rank
#include "stx_internal.h"

namespace {

__device__ __forceinline__ unsigned rot16(unsigned m, int k) { return ((m << k) | (m >> (16 - k))) & 0xFFFFu; }
__device__ __forceinline__ bool has_run12(unsigned m)
{
    const unsigned x2 = m & rot16(m, 1);
    const unsigned x4 = x2 & rot16(x2, 2);
    const unsigned x8 = x4 & rot16(x4, 4);
    return (x8 & rot16(x8, 4)) != 0;
}

__global__ __launch_bounds__(256) void fast_flags_kernel(const uint8_t *__restrict__ img, int w, int h,
                                                         size_t stride, int threshold, int words_per_row,
                                                         unsigned long long *__restrict__ words)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    bool corner = false;
    if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {  // FAST_SPACING border, :414-416
        const uint8_t *c = img + (size_t)y * stride + x;
        const int p = c[0];
        const int pb = p + threshold, p_b = p - threshold;  // :421-423
        // ring offsets in the reference's order i1..i16 (:429-452)
        const int dy[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
        const int dx[16] = {0, 1, 2, 3, 3, 3, 2, -1, 0, 1, -2, -3, -3, -3, -2, -1};
        unsigned bright = 0, dark = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int v = c[(ptrdiff_t)dy[k] * (ptrdiff_t)stride + dx[k]];
            bright |= (unsigned)(v > pb) << k;
            dark |= (unsigned)(v < p_b) << k;
        }
        corner = has_run12(bright) || has_run12(dark);
    }
    const unsigned long long bal = __ballot(corner);
    if ((threadIdx.x & 63) == 0) {
        const int wi = x >> 6;
        if (wi < words_per_row) words[(size_t)y * words_per_row + wi] = bal;
    }
}

constexpr int SCAN_THREADS = 1024;

__global__ __launch_bounds__(SCAN_THREADS) void fast_scan_kernel(const unsigned long long *__restrict__ words,
                                                                 long n_words, int *__restrict__ prefix,
                                                                 int *__restrict__ total)
{
    __shared__ int part[SCAN_THREADS];
    const int t = threadIdx.x;
    const long chunk = (n_words + SCAN_THREADS - 1) / SCAN_THREADS;
    const long lo = min((long)t * chunk, n_words), hi = min(lo + chunk, n_words);
    int sum = 0;
    for (long i = lo; i < hi; ++i) sum += __popcll(words[i]);
    part[t] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partial sums
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;  // exclusive
    for (long i = lo; i < hi; ++i) {
        prefix[i] = run;
        run += __popcll(words[i]);
    }
    if (t == SCAN_THREADS - 1) *total = part[t];
}

__global__ __launch_bounds__(256) void fast_scatter_kernel(const unsigned long long *__restrict__ words,
                                                           const int *__restrict__ prefix, long n_words,
                                                           int words_per_row, float *__restrict__ xy,
                                                           int max_out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_words) return;
    unsigned long long m = words[i];
    if (!m) return;
    const int y = (int)(i / words_per_row);
    const int x0 = (int)(i % words_per_row) * 64;
    int pos = prefix[i];
    while (m && pos < max_out) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        xy[2 * (size_t)pos] = (float)(x0 + b);      // feature.p.x = w, :463
        xy[2 * (size_t)pos + 1] = (float)y;         // feature.p.y = h, :464
        ++pos;
    }
}

}  // namespace

int stx_launch_fast(stx_ctx *ctx, const uint8_t *d_img, int w, int h, size_t stride, int threshold,
                    float *d_xy, int max_out, int *d_count)
{
    if (h > 65535) return STX_ERR_TOO_LARGE;
    const int wpr = (w + 63) / 64;
    const long n_words = (long)wpr * h;
    int err;
    // [words u64 n][prefix i32 n]
    unsigned long long *words =
        (unsigned long long *)stx_workspace(ctx, STX_SLOT_FAST, (size_t)n_words * 12 + 64, &err);
    if (!words) return err;
    int *prefix = (int *)(words + n_words);
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(fast_flags_kernel, dim3((w + 255) / 256, h), dim3(256), 0, st, d_img, w, h, stride,
                       threshold, wpr, words);
    hipLaunchKernelGGL(fast_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, words, n_words, prefix, d_count);
    hipLaunchKernelGGL(fast_scatter_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, words,
                       prefix, n_words, wpr, d_xy, max_out);
    return (int)hipGetLastError();
}
