// Rectification warp for gfx950: the loop body of RectifyImage
// (stereo/Stereography.cpp:612-630) with BilinearInterpolatePixel (:577-594) inlined.
//
// The inverse homography is computed once on the host (the reference recomputes
// H.inverse() per pixel at :617; it is loop invariant).  Every floating-point
// statement of the reference is kept as its own rounded fp32 operation in the
// same order (this file is compiled with -ffp-contract=off; '/' is the correctly
// rounded IEEE divide), so the output is bit-identical to oracle_rectify.
//
// Layout: one lane owns RECT_PX consecutive destination pixels of one row, so a
// wave covers 256 consecutive bytes of the destination row and stores them as
// dwords; the four source taps of neighbouring lanes are neighbouring bytes of
// two source rows (near-identity homographies), served from L1/L2.
#include "stx_internal.h"

namespace {

constexpr int RECT_THREADS = 256;
constexpr int RECT_PX = 4;

// (uchar) of a float known to be finite and in [0, 256): truncation toward zero
__device__ __forceinline__ unsigned trunc_u8(float v) { return (unsigned)(int)v & 0xFFu; }

__device__ __forceinline__ bool rect_pixel(const uint8_t *__restrict__ src, size_t src_stride,
                                           const float *hv, float px, float py, float xmax,
                                           float ymax, unsigned &val)
{
    // Stereography.cpp:617: 3x3 * 3x1, each row reduced as a0 + (a1 + a2), a2 = h2 * 1
    const float q0 = hv[0] * px + (hv[1] * py + hv[2]);
    const float q1 = hv[3] * px + (hv[4] * py + hv[5]);
    const float q2 = hv[6] * px + (hv[7] * py + hv[8]);
    // :618 divide by the saved third coordinate
    const float tx = q0 / q2;
    const float ty = q1 / q2;
    // :621, :623 strict bounds; NaN fails both
    if (!(0.0f < tx && tx < xmax && 0.0f < ty && ty < ymax)) return false;
    // :584-587
    const float x1 = floorf(tx), x2 = ceilf(tx);
    const float y1 = floorf(ty), y2 = ceilf(ty);
    // An integer coordinate makes the reference's weights 0/0 = NaN, which its
    // float -> uchar conversion turns into 0 (:589-592).  Otherwise x2 - x1 == 1
    // exactly and the divisions by it are identities.
    if (x1 == x2 || y1 == y2) {
        val = 0;
        return true;
    }
    const int ix = (int)x1, iy = (int)y1;
    const uint8_t *p = src + (size_t)iy * src_stride + ix;
    const float i11 = (float)p[0], i12 = (float)p[1];
    const float i21 = (float)p[src_stride], i22 = (float)p[src_stride + 1];
    const float wx1 = x2 - tx, wx2 = tx - x1;
    const unsigned y1v = trunc_u8(wx1 * i11 + wx2 * i12);  // :589
    const unsigned y2v = trunc_u8(wx1 * i21 + wx2 * i22);  // :590
    const float wy1 = y2 - ty, wy2 = ty - y1;
    val = trunc_u8(wy1 * (float)y1v + wy2 * (float)y2v);   // :592
    return true;
}

__global__ __launch_bounds__(RECT_THREADS) void rectify_kernel(
    const uint8_t *__restrict__ src, int src_w, int src_h, size_t src_stride,
    size_t src_image_stride, uint8_t *__restrict__ dst, int dst_w, int dst_h, size_t dst_stride,
    size_t dst_image_stride, const float *__restrict__ hinv, int zero_fill, int dst_y0)
{
    const int img = blockIdx.z;
    const int y = blockIdx.y;  // row of the band held at dst; image row dst_y0 + y
    const int x0 = (blockIdx.x * RECT_THREADS + threadIdx.x) * RECT_PX;
    if (x0 >= dst_w) return;
    float hv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) hv[i] = hinv[img * 9 + i];
    const uint8_t *s = src + (size_t)img * src_image_stride;
    uint8_t *drow = dst + (size_t)img * dst_image_stride + (size_t)y * dst_stride;
    const float xmax = (float)(src_w - 1), ymax = (float)(src_h - 1);
    const float py = (float)(dst_y0 + y);

    unsigned word = 0, mask = 0;
#pragma unroll
    for (int k = 0; k < RECT_PX; ++k) {
        const int x = x0 + k;
        unsigned v = 0;
        if (x < dst_w && rect_pixel(s, src_stride, hv, (float)x, py, xmax, ymax, v)) {
            word |= v << (8 * k);
            mask |= 0xFFu << (8 * k);
        }
    }
    uint8_t *d = drow + x0;
    if (x0 + RECT_PX <= dst_w && ((uintptr_t)d & 3) == 0) {
        unsigned *dw = reinterpret_cast<unsigned *>(d);
        if (zero_fill || mask == 0xFFFFFFFFu) {
            *dw = word;
        } else if (mask != 0) {
            *dw = (*dw & ~mask) | word;  // misses keep their prior value
        }
    } else {
#pragma unroll
        for (int k = 0; k < RECT_PX; ++k) {
            if (x0 + k >= dst_w) break;
            if ((mask >> (8 * k)) & 1u)
                d[k] = (uint8_t)(word >> (8 * k));
            else if (zero_fill)
                d[k] = 0;
        }
    }
}

}  // namespace

int stx_launch_rectify(stx_ctx *ctx, const uint8_t *d_src, int src_w, int src_h,
                       size_t src_stride, size_t src_image_stride, uint8_t *d_dst, int dst_w,
                       int dst_h, size_t dst_stride, size_t dst_image_stride,
                       const float *d_hinv, int n_images, int flags, int dst_y0)
{
    if (dst_w <= 0 || dst_h <= 0 || n_images <= 0) return STX_OK;
    if (dst_h > 65535 || n_images > 65535) return STX_ERR_TOO_LARGE;
    dim3 grid((dst_w + RECT_THREADS * RECT_PX - 1) / (RECT_THREADS * RECT_PX), dst_h, n_images);
    hipLaunchKernelGGL(rectify_kernel, grid, dim3(RECT_THREADS), 0, ctx->stream, d_src, src_w,
                       src_h, src_stride, src_image_stride, d_dst, dst_w, dst_h, dst_stride,
                       dst_image_stride, d_hinv, flags == STX_RECT_DST_IS_ZERO ? 1 : 0, dst_y0);
    return (int)hipGetLastError();
}
