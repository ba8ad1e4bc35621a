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

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// Stereography.cpp:618, both quotients n0/d and n1/d, correctly rounded.
//
// The compiler's expansion of an IEEE fp32 divide on gfx950 is v_div_scale x2, v_rcp, a
// Newton step on the reciprocal, two residual corrections of the quotient, v_div_fmas and
// v_div_fixup.  When |n| and |d| lie in [2^-40, 2^40] neither v_div_scale rescales
// anything (exponent difference < 96, no denormal operand, reciprocal or quotient, numerator
// exponent > 23), v_div_fmas is a plain fma and v_div_fixup returns its input, so the same
// fma chain without those three instructions produces the same bits; the reciprocal and its
// Newton step are shared by the two quotients and the chains run as packed fp32.  Lanes
// outside that range (a coordinate crossing zero, a degenerate homography) take the
// compiler's full divide.
__device__ __forceinline__ f32x2 divide_pair(f32x2 n, float d)
{
    const float lo = fminf(fminf(fabsf(n.x), fabsf(n.y)), fabsf(d));
    const float hi = fmaxf(fmaxf(fabsf(n.x), fabsf(n.y)), fabsf(d));
    f32x2 t;
    if (lo >= 0x1p-40f && hi <= 0x1p40f) {
        float r = __builtin_amdgcn_rcpf(d);
        const float e = __builtin_fmaf(-d, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        const f32x2 nd = {-d, -d}, rr = {r, r};
        t = n * rr;
        f32x2 res = fma2(nd, t, n);
        t = fma2(res, rr, t);
        res = fma2(nd, t, n);
        t = fma2(res, rr, t);
    } else {
        t.x = n.x / d;
        t.y = n.y / d;
    }
    return t;
}

// Four consecutive destination pixels per lane, in three branch-free phases so that the
// eight tap loads of a lane are in flight together.
//
// BilinearInterpolatePixel (:577-594) for an in-bounds, hence positive, coordinate t:
//   x1 = floor(t), x2 = ceil(t); t integer  =>  weights 0/0 = NaN  =>  (uchar) gives 0;
//   otherwise x2 - x1 == 1 exactly, the divisions by it are identities,
//   wx2 = t - x1 is exact in fp32 (= fract(t)) and wx1 = x2 - t is the rounding of the
//   real number 1 - wx2, i.e. the fp32 operation 1.0f - wx2.
// (uchar) of the interpolated values (all in [0, 256)) is truncation.
// Up to two inverse homographies travel as kernel arguments (hinv == nullptr): one image, or
// the pair of main.cpp:362-369, needs no upload at all.
struct RectHinv {
    float h[18];
};

template <bool ZERO_FILL>
__global__ __launch_bounds__(RECT_THREADS) void rectify_kernel(
    const uint8_t *__restrict__ src, int src_w, int src_h, unsigned src_stride,
    size_t src_image_stride, uint8_t *__restrict__ dst, int dst_w, int dst_h, size_t dst_stride,
    size_t dst_image_stride, const float *__restrict__ hinv, RectHinv hk, int dst_y0)
{
    const int img = blockIdx.z;
    const int y = blockIdx.y;  // row of the band held at dst; image row dst_y0 + y
    const int x0 = (blockIdx.x * RECT_THREADS + threadIdx.x) * RECT_PX;
    if (x0 >= dst_w) return;
    float hv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) hv[i] = hinv ? hinv[img * 9 + i] : hk.h[(img & 1) * 9 + i];
    const uint8_t *s = src + (size_t)img * src_image_stride;
    // buffer descriptor over this source image: taps are fetched with a 32-bit byte offset in
    // a VGPR and the row pitch as the scalar offset, no 64-bit address arithmetic per tap
    const __amdgpu_buffer_rsrc_t taps = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(s), 0, (int)(src_stride * (unsigned)src_h), 0x00020000);
    uint8_t *drow = dst + (size_t)img * dst_image_stride + (size_t)y * dst_stride;
    const float xmax = (float)(src_w - 1), ymax = (float)(src_h - 1);
    const float py = (float)(dst_y0 + y);
    // Stereography.cpp:617: 3x3 * 3x1, each row reduced as a0 + (a1 + a2), a2 = h2 * 1
    const float c0 = hv[1] * py + hv[2], c1 = hv[4] * py + hv[5], c2 = hv[7] * py + hv[8];

    f32x2 t[RECT_PX];
#pragma unroll
    for (int k = 0; k < RECT_PX; ++k) {
        const float px = (float)(x0 + k);
        const f32x2 q = {hv[0] * px + c0, hv[3] * px + c1};
        const float q2 = hv[6] * px + c2;
        t[k] = divide_pair(q, q2);  // :618
    }
    // :621, :623 strict bounds (NaN fails); a pixel that fails is not written
    bool inb[RECT_PX];
    unsigned off[RECT_PX], row0[RECT_PX], row1[RECT_PX];
    bool any = false;
#pragma unroll
    for (int k = 0; k < RECT_PX; ++k) {
        inb[k] = 0.0f < t[k].x && t[k].x < xmax && 0.0f < t[k].y && t[k].y < ymax && x0 + k < dst_w;
        any |= inb[k];
        // in bounds implies 0 <= iy <= src_h - 2, 0 <= ix <= src_w - 2: both taps of both rows exist
        off[k] = inb[k] ? (unsigned)(int)t[k].y * src_stride + (unsigned)(int)t[k].x : 0u;
        row0[k] = row1[k] = 0;
    }
    if (any) {  // lanes with no pixel in bounds load nothing (src may be smaller than 2x2)
#pragma unroll
        for (int k = 0; k < RECT_PX; ++k) {
            // two unaligned 16-bit loads: taps (x1, x2) of source rows y1 and y2
            row0[k] = __builtin_amdgcn_raw_buffer_load_b16(taps, off[k], 0, 0);
            row1[k] = __builtin_amdgcn_raw_buffer_load_b16(taps, off[k], src_stride, 0);
        }
    }
    unsigned word = 0, mask = 0;
#pragma unroll
    for (int k = 0; k < RECT_PX; ++k) {
        const float wx2 = __builtin_amdgcn_fractf(t[k].x), wy2 = __builtin_amdgcn_fractf(t[k].y);
        const float wx1 = 1.0f - wx2, wy1 = 1.0f - wy2;
        const f32x2 left = {(float)(row0[k] & 0xFFu), (float)(row1[k] & 0xFFu)};
        const f32x2 right = {(float)(row0[k] >> 8), (float)(row1[k] >> 8)};
        const f32x2 rows = left * (f32x2){wx1, wx1} + right * (f32x2){wx2, wx2};  // :589, :590
        const float y1v = __builtin_truncf(rows.x), y2v = __builtin_truncf(rows.y);
        const unsigned v = (unsigned)(int)(wy1 * y1v + wy2 * y2v);                // :592
        // integer coordinate on either axis: the reference's NaN weights give 0
        const bool live = inb[k] && fminf(wx2, wy2) != 0.0f;
        word |= (live ? v : 0u) << (8 * k);
        mask |= (inb[k] ? 0xFFu : 0u) << (8 * k);
    }
    uint8_t *d = drow + x0;
    if (x0 + RECT_PX <= dst_w && ((uintptr_t)d & 3) == 0) {
        unsigned *dw = reinterpret_cast<unsigned *>(d);
        if (ZERO_FILL || mask == 0xFFFFFFFFu) {
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
            else if (ZERO_FILL)
                d[k] = 0;
        }
    }
}

}  // namespace

int stx_launch_rectify(stx_ctx *ctx, const uint8_t *d_src, int src_w, int src_h,
                       size_t src_stride, size_t src_image_stride, uint8_t *d_dst, int dst_w,
                       int dst_h, size_t dst_stride, size_t dst_image_stride,
                       const float *d_hinv, int n_images, int flags, int dst_y0, const float *host_hinv)
{
    if (dst_w <= 0 || dst_h <= 0 || n_images <= 0) return STX_OK;
    if (dst_h > 65535 || n_images > 65535) return STX_ERR_TOO_LARGE;
    // tap offsets are 32-bit: one source image must stay below 4 GiB
    if (src_h > 0 && (unsigned long long)src_stride * (unsigned long long)src_h > 0xFFFFFFFFull)
        return STX_ERR_TOO_LARGE;
    RectHinv hk = {};
    if (host_hinv) {
        if (n_images > 2) return STX_ERR_INVALID_ARG;
        for (int i = 0; i < 9 * n_images; ++i) hk.h[i] = host_hinv[i];
        d_hinv = nullptr;
    } else if (!d_hinv) {
        return STX_ERR_INVALID_ARG;
    }
    dim3 grid((dst_w + RECT_THREADS * RECT_PX - 1) / (RECT_THREADS * RECT_PX), dst_h, n_images);
    if (flags == STX_RECT_DST_IS_ZERO)
        hipLaunchKernelGGL(rectify_kernel<true>, grid, dim3(RECT_THREADS), 0, ctx->stream, d_src, src_w,
                           src_h, (unsigned)src_stride, src_image_stride, d_dst, dst_w, dst_h, dst_stride,
                           dst_image_stride, d_hinv, hk, dst_y0);
    else
        hipLaunchKernelGGL(rectify_kernel<false>, grid, dim3(RECT_THREADS), 0, ctx->stream, d_src, src_w,
                           src_h, (unsigned)src_stride, src_image_stride, d_dst, dst_w, dst_h, dst_stride,
                           dst_image_stride, d_hinv, hk, dst_y0);
    return (int)hipGetLastError();
}
