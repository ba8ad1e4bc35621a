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
//   fast_flags_kernel    workgroup = 256 columns x 8 rows staged through LDS (with the 3-pixel
//                        ring halo); one lane per column, wave ballot -> one 64-bit word per
//                        64 pixels, and the tile's corner count
//   fast_scatter_kernel  workgroup = one block of 8 rows: sums the counts of all tiles above
//                        it (a few thousand ints), scans its own words and writes its corners
//                        at prefix + rank.  No global scan kernel, no atomics.
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

constexpr int FT_ROWS = 8;            // rows per tile / per scatter block
constexpr int FT_COLS = 256;          // columns per tile = threads per workgroup
constexpr int FT_PITCH = FT_COLS + 8; // LDS row: columns x0-3 .. x0+258, padded

__global__ __launch_bounds__(FT_COLS) void fast_flags_kernel(const uint8_t *__restrict__ img, int w, int h,
                                                             size_t stride, int threshold, int words_per_row,
                                                             unsigned long long *__restrict__ words,
                                                             int *__restrict__ tilecount)
{
    __shared__ uint8_t tile[FT_ROWS + 6][FT_PITCH];
    __shared__ int wsum[FT_COLS / 64];
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * FT_COLS, y0 = blockIdx.y * FT_ROWS;
    // stage rows y0-3 .. y0+FT_ROWS+2, columns x0-3 .. x0+258.  Coordinates are clamped into
    // the image instead of tested: a clamped value is only ever a ring point of a pixel inside
    // the 3-pixel border, which is never tested.  All loads are issued before the first write.
    {
        const int cx = min(max(x0 - 3 + t, 0), w - 1);
        const int cx2 = min(x0 - 3 + FT_COLS + t, w - 1);
        uint8_t v[FT_ROWS + 6], e[FT_ROWS + 6];
#pragma unroll
        for (int r = 0; r < FT_ROWS + 6; ++r) {
            const uint8_t *row = img + (size_t)min(max(y0 - 3 + r, 0), h - 1) * stride;
            v[r] = row[cx];
            if (t < 6) e[r] = row[cx2];
        }
#pragma unroll
        for (int r = 0; r < FT_ROWS + 6; ++r) {
            tile[r][t] = v[r];
            if (t < 6) tile[r][FT_COLS + t] = e[r];
        }
    }
    __syncthreads();
    const int x = x0 + t;
    const bool col_ok = x >= 3 && x < w - 3;  // FAST_SPACING border, :414-416
    int cnt = 0;
    for (int r = 0; r < FT_ROWS; ++r) {
        const int y = y0 + r;
        if (y >= h) break;
        bool corner = false;
        if (col_ok && y >= 3 && y < h - 3) {
            const uint8_t *c = &tile[r + 3][t + 3];
            const int p = c[0];
            const int pb = p + threshold, p_b = p - threshold;  // :421-423
            // ring offsets in the reference's order i1..i16 (:429-452)
            const int dy[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
            const int dx[16] = {0, 1, 2, 3, 3, 3, 2, -1, 0, 1, -2, -3, -3, -3, -2, -1};
            unsigned bright = 0, dark = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int v = c[dy[k] * FT_PITCH + dx[k]];
                // sign bit of (pb - v) <=> v > pb; of (v - p_b) <=> v < p_b; shifted in at the bottom,
                // which reverses the ring order -- a circular run is a run either way round
                bright = __builtin_amdgcn_alignbit(bright, (unsigned)(pb - v), 31);
                dark = __builtin_amdgcn_alignbit(dark, (unsigned)(v - p_b), 31);
            }
            corner = has_run12(bright & 0xFFFFu) || has_run12(dark & 0xFFFFu);
        }
        const unsigned long long bal = __ballot(corner);
        cnt += __popcll(bal);
        if ((t & 63) == 0) {
            const int wi = x >> 6;
            if (wi < words_per_row) words[(size_t)y * words_per_row + wi] = bal;
        }
    }
    if ((t & 63) == 0) wsum[t >> 6] = cnt;
    __syncthreads();
    if (t == 0) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < FT_COLS / 64; ++k) s += wsum[k];
        tilecount[blockIdx.y * gridDim.x + blockIdx.x] = s;
    }
}

// Workgroup-wide inclusive scan of one int per thread (256 threads); returns the inclusive
// value, *total receives the sum over the workgroup.
__device__ __forceinline__ int block_scan_256(int v, int *lds4, int *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(inc, off, 64);
        if (lane >= off) inc += n;
    }
    __syncthreads();  // lds4 may still be read from the previous call
    if (lane == 63) lds4[wv] = inc;
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = lds4[k];
        if (k < wv) base += s;
        sum += s;
    }
    *total = sum;
    return inc + base;
}

__global__ __launch_bounds__(256) void fast_scatter_kernel(const unsigned long long *__restrict__ words,
                                                           const int *__restrict__ tilecount, int tiles_x,
                                                           int h, int words_per_row, float *__restrict__ xy,
                                                           int max_out, int *__restrict__ total_out)
{
    __shared__ int lds4[4];
    const int t = threadIdx.x;
    const int rb = blockIdx.x;
    // corners in all row blocks above this one
    int part = 0;
    for (int i = t; i < rb * tiles_x; i += 256) part += tilecount[i];
    int above;
    block_scan_256(part, lds4, &above);
    const int y0 = rb * FT_ROWS;
    const int rows = min(FT_ROWS, h - y0);
    const long first = (long)y0 * words_per_row;
    const int n = rows * words_per_row;
    int run = above;
    for (int base = 0; base < n; base += 256) {  // uniform trip count
        const int i = base + t;
        unsigned long long m = i < n ? words[first + i] : 0ull;
        int chunk;
        const int inc = block_scan_256(__popcll(m), lds4, &chunk);
        int pos = run + inc - __popcll(m);
        const int y = y0 + i / words_per_row;
        const int xw = (i % words_per_row) * 64;
        while (m && pos < max_out) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            xy[2 * (size_t)pos] = (float)(xw + b);  // feature.p.x = w, :463
            xy[2 * (size_t)pos + 1] = (float)y;     // feature.p.y = h, :464
            ++pos;
        }
        run += chunk;
    }
    if (rb == (int)gridDim.x - 1 && t == 0) *total_out = run;
}

}  // namespace

int stx_launch_fast(stx_ctx *ctx, const uint8_t *d_img, int w, int h, size_t stride, int threshold,
                    float *d_xy, int max_out, int *d_count)
{
    const int wpr = (w + 63) / 64;
    const int tiles_x = (w + FT_COLS - 1) / FT_COLS;
    const int row_blocks = (h + FT_ROWS - 1) / FT_ROWS;
    if (row_blocks > 65535) return STX_ERR_TOO_LARGE;
    const long n_words = (long)wpr * h;
    int err;
    // [words u64 n_words][tilecount i32 tiles_x * row_blocks]
    unsigned long long *words = (unsigned long long *)stx_workspace(
        ctx, STX_SLOT_FAST, (size_t)n_words * 8 + (size_t)tiles_x * row_blocks * 4 + 64, &err);
    if (!words) return err;
    int *tilecount = (int *)(words + n_words);
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(fast_flags_kernel, dim3(tiles_x, row_blocks), dim3(FT_COLS), 0, st, d_img, w, h, stride,
                       threshold, wpr, words, tilecount);
    hipLaunchKernelGGL(fast_scatter_kernel, dim3(row_blocks), dim3(256), 0, st, words, tilecount, tiles_x, h,
                       wpr, d_xy, max_out, d_count);
    return (int)hipGetLastError();
}
