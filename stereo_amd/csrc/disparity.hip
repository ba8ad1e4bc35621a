// Dense disparity scan for gfx950.
//
//   disp_ref_kernel   reference mode (ComputeDepthImage, stereo/Stereography.cpp:649-691):
//                     1x1 window, every column of the row is a candidate.
//   disp_ext_kernel   D1-ext (this library's generalisation): win x win
//                     SAD/SSD, candidates x1 in [max(0,x-D+1), x].
//
// All arithmetic is integer; results are bit-exact against oracle/stereo_oracle.c.
#include "stx_internal.h"

namespace {

// ---------------------------------------------------------------------------
// Reference mode.
//
// One workgroup = one image row x REF_PX*256 consecutive pixels.  The row of img1
// is staged through LDS in chunks as a table of 32-bit "candidate words"
//     T[x1] = img1[y][x1] != 0 ? img1[y][x1] << 20 : 0xFFF00000
// and each lane keeps, per owned pixel, a packed key (|p1-p0| << 20) + x1 whose
// running unsigned minimum is exactly the reference's scan: smallest distance
// first, and for equal distances the smallest x1 (strict '<' at
// Stereography.cpp:677 keeps the first).  One candidate costs one v_sad_u32
// (|a-b|+c in one instruction, the candidate word is a broadcast LDS read and
// x1 a scalar) and half a v_min3_u32.  The zero-skip at :676 is the 0xFFF00000
// word, which can never beat the initial key 255<<20 (:671-672).
// ---------------------------------------------------------------------------

constexpr int REF_THREADS = 256;
constexpr int REF_PX = 4;        // pixels per lane
constexpr int REF_CHUNK = 2048;  // candidate words per LDS chunk (8 KiB)
constexpr unsigned REF_SHIFT = 20;
constexpr unsigned REF_SKIP = 0xFFF00000u;
constexpr unsigned REF_INIT = 255u << REF_SHIFT;

__device__ __forceinline__ unsigned absdiff_add(unsigned a, unsigned b, unsigned c)
{
    // pattern-matched by hipcc to v_sad_u32
    return (a > b ? a - b : b - a) + c;
}

__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c)
{
    return min(a, min(b, c));
}

__global__ __launch_bounds__(REF_THREADS) void disp_ref_kernel(
    const uint8_t *__restrict__ img0, const uint8_t *__restrict__ img1, int w, size_t stride,
    size_t pair_stride, int row0, uint8_t *__restrict__ out, size_t out_stride,
    size_t out_pair_stride, int out_bytes)
{
    __shared__ uint4 tab4[REF_CHUNK / 4];
    unsigned *tab = reinterpret_cast<unsigned *>(tab4);

    const int y = row0 + blockIdx.y;
    const size_t pair = blockIdx.z;
    const uint8_t *r0 = img0 + pair * pair_stride + (size_t)y * stride;
    const uint8_t *r1 = img1 + pair * pair_stride + (size_t)y * stride;
    const int xbase = blockIdx.x * (REF_THREADS * REF_PX);

    unsigned a[REF_PX], best[REF_PX];
#pragma unroll
    for (int p = 0; p < REF_PX; ++p) {
        const int x = xbase + p * REF_THREADS + threadIdx.x;
        const unsigned p0 = x < w ? r0[x] : 0u;
        a[p] = p0 << REF_SHIFT;
        best[p] = REF_INIT;
    }

    for (int c0 = 0; c0 < w; c0 += REF_CHUNK) {
        const int n = min(REF_CHUNK, w - c0);
        const int n4 = (n + 3) & ~3;
        __syncthreads();
        for (int i = threadIdx.x; i < n4; i += REF_THREADS) {
            const unsigned p1 = i < n ? r1[c0 + i] : 0u;
            tab[i] = p1 ? (p1 << REF_SHIFT) : REF_SKIP;
        }
        __syncthreads();
        for (int i = 0; i < n4; i += 4) {
            const uint4 t = tab4[i >> 2];  // same address in every lane: LDS broadcast
            // the four candidate columns as scalars; the empty asm keeps hipcc from
            // folding the +1..+3 into a separate vector add after the |a-b|
            unsigned x1 = (unsigned)(c0 + i), x1b = x1 + 1, x1c = x1 + 2, x1d = x1 + 3;
            asm volatile("" : "+s"(x1), "+s"(x1b), "+s"(x1c), "+s"(x1d));
#pragma unroll
            for (int p = 0; p < REF_PX; ++p) {
                const unsigned k0 = absdiff_add(a[p], t.x, x1);
                const unsigned k1 = absdiff_add(a[p], t.y, x1b);
                const unsigned k2 = absdiff_add(a[p], t.z, x1c);
                const unsigned k3 = absdiff_add(a[p], t.w, x1d);
                best[p] = umin3(best[p], k0, k1);
                asm volatile("" : "+v"(best[p]));  // keep two v_min3_u32 (else min,min,min3)
                best[p] = umin3(best[p], k2, k3);
            }
        }
    }

    uint8_t *orow = out + pair * out_pair_stride + (size_t)blockIdx.y * out_stride;
#pragma unroll
    for (int p = 0; p < REF_PX; ++p) {
        const int x = xbase + p * REF_THREADS + threadIdx.x;
        if (x < w) {
            const int xbest = (int)(best[p] & ((1u << REF_SHIFT) - 1));
            int v = xbest - x;
            v = v < 0 ? -v : v;
            if (a[p] == 0) v = 0;  // pixel0 == 0: the reference leaves Mat::zeros (:670)
            if (out_bytes == 1)
                orow[x] = (uint8_t)(v & 0xFF);  // int -> uchar at :686
            else
                reinterpret_cast<uint16_t *>(orow)[x] = (uint16_t)(v & 0xFFFF);
        }
    }
}

// ---------------------------------------------------------------------------
// D1-ext, first version: every (pixel, candidate) cost is summed tap by tap
// from LDS tiles.  Correct for every supported parameter combination; used for
// the parameter ranges the sliding-sum kernel does not cover.
// ---------------------------------------------------------------------------

constexpr int EXT_TX = 64;  // tile columns
constexpr int EXT_TY = 16;  // tile rows
constexpr int EXT_THREADS = 256;

template <bool SSD>
__global__ __launch_bounds__(EXT_THREADS) void disp_ext_bruteforce_kernel(
    const uint8_t *__restrict__ img0, const uint8_t *__restrict__ img1, int w, int h,
    size_t stride, size_t pair_stride, int row0, int rows, int r, int max_disp,
    uint8_t *__restrict__ out, size_t out_stride, size_t out_pair_stride, int out_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lw = EXT_TX + 2 * r;                 // L tile width
    const int rw = EXT_TX + 2 * r + max_disp - 1;  // R tile width
    const int th = EXT_TY + 2 * r;
    uint8_t *Lt = smem;
    uint8_t *Rt = smem + (size_t)lw * th;

    const size_t pair = blockIdx.z;
    const uint8_t *i0 = img0 + pair * pair_stride;
    const uint8_t *i1 = img1 + pair * pair_stride;
    const int x0 = blockIdx.x * EXT_TX;
    const int y0 = row0 + blockIdx.y * EXT_TY;

    // stage tiles; out-of-image positions are zero-filled and never used as taps
    for (int i = threadIdx.x; i < lw * th; i += EXT_THREADS) {
        const int ty = i / lw, tx = i - ty * lw;
        const int gy = y0 - r + ty, gx = x0 - r + tx;
        Lt[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? i0[(size_t)gy * stride + gx] : 0;
    }
    for (int i = threadIdx.x; i < rw * th; i += EXT_THREADS) {
        const int ty = i / rw, tx = i - ty * rw;
        const int gy = y0 - r + ty, gx = x0 - r - (max_disp - 1) + tx;
        Rt[i] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? i1[(size_t)gy * stride + gx] : 0;
    }
    __syncthreads();

    const int lx = threadIdx.x & (EXT_TX - 1);
    const int x = x0 + lx;
    if (x >= w) return;
    const int win = 2 * r + 1;
    const unsigned init = (SSD ? 65025u : 255u) * (unsigned)win * (unsigned)win;
    const int dx_hi = min(r, w - 1 - x);

    for (int ly = threadIdx.x / EXT_TX; ly < EXT_TY; ly += EXT_THREADS / EXT_TX) {
        const int y = y0 + ly;
        if (y >= h || y >= row0 + rows) break;
        int v = 0;
        const unsigned p0 = Lt[(ly + r) * lw + lx + r];
        if (p0 != 0) {
            const int dy_lo = max(-r, -y), dy_hi = min(r, h - 1 - y);
            unsigned best = init;
            int xbest = 0;
            const int dmax = min(max_disp - 1, x);
            for (int d = dmax; d >= 0; --d) {  // ascending x1
                // R tile column of image column c: c - (x0 - r - (max_disp-1))
                const int rc = lx + r + (max_disp - 1) - d;  // column of x1 = x - d
                if (Rt[(ly + r) * rw + rc] == 0) continue;
                const int dx_lo = max(-r, d - x);
                unsigned cost = 0;
                for (int dy = dy_lo; dy <= dy_hi; ++dy) {
                    const uint8_t *lrow = Lt + (ly + r + dy) * lw + lx + r;
                    const uint8_t *rrow = Rt + (ly + r + dy) * rw + rc;
                    for (int dx = dx_lo; dx <= dx_hi; ++dx) {
                        const int a = lrow[dx], b = rrow[dx];
                        const unsigned ad = (unsigned)(a > b ? a - b : b - a);
                        cost += SSD ? ad * ad : ad;
                    }
                }
                if (cost < best) {
                    best = cost;
                    xbest = x - d;
                }
            }
            v = x - xbest;  // xbest <= x
        }
        uint8_t *orow = out + pair * out_pair_stride + (size_t)(y - row0) * out_stride;
        if (out_bytes == 1)
            orow[x] = (uint8_t)(v & 0xFF);
        else
            reinterpret_cast<uint16_t *>(orow)[x] = (uint16_t)(v & 0xFFFF);
    }
}

}  // namespace

int stx_launch_disparity(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1,
                         int n_pairs, int w, int h, size_t stride, size_t pair_stride,
                         int row0, int rows, int win, int max_disp, int cost_mode,
                         void *d_out, size_t out_stride, size_t out_pair_stride,
                         int out_bytes)
{
    if (w <= 0 || h <= 0 || rows <= 0 || n_pairs <= 0) return STX_OK;
    if (rows > 65535 * EXT_TY || n_pairs > 65535) return STX_ERR_TOO_LARGE;
    uint8_t *o = static_cast<uint8_t *>(d_out);

    if (max_disp == 0) {
        // reference mode (win == 1 checked by the caller); SSD orders candidates
        // exactly as SAD does for a single tap, so one kernel serves both.
        if (w > (1 << REF_SHIFT)) return STX_ERR_TOO_LARGE;
        if (rows > 65535) return STX_ERR_TOO_LARGE;
        dim3 grid((w + REF_THREADS * REF_PX - 1) / (REF_THREADS * REF_PX), rows, n_pairs);
        hipLaunchKernelGGL(disp_ref_kernel, grid, dim3(REF_THREADS), 0, ctx->stream, d_img0,
                           d_img1, w, stride, pair_stride, row0, o, out_stride, out_pair_stride,
                           out_bytes);
        return (int)hipGetLastError();
    }

    const int r = win / 2;
    const int D = max_disp < w ? max_disp : w;  // candidates beyond column 0 do not exist
    const size_t lds = (size_t)(EXT_TX + 2 * r) * (EXT_TY + 2 * r) +
                       (size_t)(EXT_TX + 2 * r + D - 1) * (EXT_TY + 2 * r);
    if (lds > 64 * 1024) return STX_ERR_TOO_LARGE;
    dim3 grid((w + EXT_TX - 1) / EXT_TX, (rows + EXT_TY - 1) / EXT_TY, n_pairs);
    if (cost_mode == STX_COST_SSD)
        hipLaunchKernelGGL(disp_ext_bruteforce_kernel<true>, grid, dim3(EXT_THREADS), lds,
                           ctx->stream, d_img0, d_img1, w, h, stride, pair_stride, row0, rows, r, D,
                           o, out_stride, out_pair_stride, out_bytes);
    else
        hipLaunchKernelGGL(disp_ext_bruteforce_kernel<false>, grid, dim3(EXT_THREADS), lds,
                           ctx->stream, d_img0, d_img1, w, h, stride, pair_stride, row0, rows, r, D,
                           o, out_stride, out_pair_stride, out_bytes);
    return (int)hipGetLastError();
}
