// SIFT-like descriptors for gfx950: the per-feature body of CreateSIFTDescriptors
// (stereo/Features.cpp:844-934) with ComputeFeatureOrientation (:945-995), on the CV_8U
// gradient images the reference makes with OpenCV's GaussianBlur and Sobel (:830-837; they
// stay with the caller).  Output: the 128-float descriptors that MatchDescriptors consumes.
//
// Bit-exactness.  The only transcendental on the path is atan(gY / gX) of two gradient BYTES
// (:976, :886), i.e. one of 65 536 values: the host tabulates RAD2DEG(atanf(gY / gX)) once per
// context with its own libm -- the function the reference's host code would call -- and the
// kernel looks the angle up, so the result follows the host's libm exactly, whatever it is.
// Everything else (magnitude sqrt, the kernel weights, histogram accumulation in the
// reference's tap order, the two normalisations with their sequential sums of squares and
// true divisions, the 0.2 clamp) is kept as separately rounded fp32 operations
// (-ffp-contract=off).  Kept as written: the angle in DEGREES minus f.angle in RADIANS
// (:890), PI = 3.14159f (Features.h:42), the Gaussian kernels centred at size/2.f.
//
// Sixteen lanes per feature.  What is order dependent is the fp32 accumulation INTO one
// histogram bin and the two sums of squares; everything else is independent work:
//   orientation: the 81 taps' bins and weighted magnitudes are computed five or six per lane;
//     lane b owns bin b of the histogram (gradient bytes are >= 0, so the tabulated angles lie
//     in [0, 90.0001] and only bins 0..9 of the 36 can be reached), walks the taps in the
//     reference's order and adds those of its own bin;
//   cells: lane c owns cell c of the 4x4 grid (its 16 taps, its 8 bins, in the reference's order);
//   normalisation: each lane squares and divides its own 8 values; the two sums of squares run
//     over all 128 values in index order from LDS, in every lane alike.
#include "stx_internal.h"

#include <math.h>
#include <memory>
#include <new>

namespace {

constexpr int SIFT_LANES = 16;                          // lanes per feature
constexpr int SIFT_THREADS = 256;
constexpr int SIFT_FEATS = SIFT_THREADS / SIFT_LANES;   // features per workgroup
constexpr int SIFT_OBINS = 10;  // reachable bins of the 36-bin orientation histogram
constexpr float SIFT_PI = 3.14159f;  // Features.h:42

__global__ __launch_bounds__(SIFT_THREADS) void sift_kernel(
    const uint8_t *__restrict__ gx, const uint8_t *__restrict__ gy, int w, int h, size_t stride,
    const float2 *__restrict__ xy, int n, const float *__restrict__ tables, float *__restrict__ angle,
    float *__restrict__ desc)
{
    __shared__ float chist[8][SIFT_THREADS];             // one column of cell bins per lane
    __shared__ float ohist[SIFT_FEATS][SIFT_LANES];       // orientation bins of a feature
    __shared__ int tap_bin[SIFT_FEATS][81];               // orientation taps: bin (-1: outside the image)
    __shared__ float tap_term[SIFT_FEATS][81];            // ... and weighted magnitude
    __shared__ __align__(16) float sq[SIFT_FEATS][128];   // squares of a feature's 128 values
    const float *lut = tables;           // [gy][gx] angle in degrees, 90 where gx == 0
    const float *k9 = tables + 65536;    // CreateGaussianKernel(9, 1)
    const float *k4 = k9 + 81;           // CreateGaussianKernel(4, 1)
    const int t = threadIdx.x;
    const int g = t / SIFT_LANES, l = t % SIFT_LANES;
    const int fi = blockIdx.x * SIFT_FEATS + g;
    const bool live = fi < n;
    const int f = live ? fi : n - 1;  // spare lanes repeat the last feature and store nothing
    const float2 p = xy[f];
    const int cx = (int)p.x, cy = (int)p.y;

    // ---- ComputeFeatureOrientation, :945-995 --------------------------------------
    // the 81 taps are evaluated five or six per lane; then lane l walks them in the reference's
    // order and accumulates bin l
    for (int k = l; k < 81; k += SIFT_LANES) {
        const int nn = k / 9 - 4, m = k % 9 - 4;
        const int i = nn + cy, j = m + cx;
        int bin = -1;
        float term = 0.0f;
        if (i >= 0 && i < h && j >= 0 && j < w) {  // :964, :966
            const int vx = gx[(size_t)i * stride + j], vy = gy[(size_t)i * stride + j];
            const float gX = (float)vx, gY = (float)vy;
            const float mag = sqrtf(gX * gX + gY * gY);  // :973
            const float ang = lut[vy * 256 + vx];        // :974-978
            bin = (int)(ang / 10.f);
            term = mag * k9[k];
        }
        tap_bin[g][k] = bin;
        tap_term[g][k] = term;
    }
    __syncthreads();
    float acc = 0.0f;
    for (int k = 0; k < 81; ++k)
        if (tap_bin[g][k] == l) acc = acc + tap_term[g][k];  // :979
    ohist[g][l] = acc;
    __syncthreads();
    float fang = angle[f];  // kept when every bin is 0
    float dominant = 0.f;
#pragma unroll
    for (int b = 0; b < SIFT_OBINS; ++b) {
        const float v = ohist[g][b];
        if (v > dominant) {  // :988
            dominant = v;
            fang = (float)b * 10.f * SIFT_PI / 180.f;  // DEG2RAD(i * 10.f), :992
        }
    }
    // every lane of the feature has read angle[f] before lane 0 replaces it
    __syncthreads();
    if (live && l == 0) angle[f] = fang;

    // ---- 4x4 cells x 8 bins, :860-906: lane l is cell (l / 4, l % 4) --------------
    {
        const int j = (l / 4) * 4, k = (l % 4) * 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) chist[b][t] = 0.0f;
        const float wgt = k4[l];  // k4[(j / 4) * 4 + (k / 4)]
        for (int nn = j; nn < j + 4; ++nn) {
            const int imgY = cy - 8 + nn;  // :871
            if (imgY < 0 || imgY >= h) continue;
            for (int m = k; m < k + 4; ++m) {
                const int imgX = cx - 8 + m;  // :870
                if (imgX < 0 || imgX >= w) continue;
                const int vx = gx[(size_t)imgY * stride + imgX], vy = gy[(size_t)imgY * stride + imgX];
                const float gX = (float)vx, gY = (float)vy;
                const float mag = sqrtf(gX * gX + gY * gY);  // :883
                float ang = lut[vy * 256 + vx];              // :884-888
                ang = ang - fang;                            // :890
                if (ang > 360.f) ang = (float)((int)ang % 360);  // :891-892
                if (ang < 0.f) ang = 0.f;                    // :893-894
                int bin = (int)ang / 45;                     // :895
                if (bin > 7) bin = 7;  // ang == 360: the reference writes past hist[7]
                chist[bin][t] = chist[bin][t] + mag * wgt;
            }
        }
    }

    // ---- normalise, clamp at 0.2, renormalise, :909-932 -----------------------
    // this lane's values are vec[8 l .. 8 l + 7] (:900-904 appends the cells in lane order)
    float v[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        v[b] = chist[b][t];
        sq[g][8 * l + b] = v[b] * v[b];
    }
    __syncthreads();
    float norm = 0.f;
    for (int i = 0; i < 128; ++i) norm = norm + sq[g][i];  // L2_norm, :806-815, in index order
    const float s = sqrtf(norm);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        float q = v[b] / s;          // :822
        if (q > 0.2f) q = 0.2f;      // ILLUMINANCE_BOUND, :922-925
        v[b] = q;
        sq[g][8 * l + b] = q * q;
    }
    __syncthreads();
    float norm2 = 0.f;
    for (int i = 0; i < 128; ++i) norm2 = norm2 + sq[g][i];
    const float s2 = sqrtf(norm2);
    if (live) {
        float *vec = desc + (size_t)f * 128 + 8 * l;
#pragma unroll
        for (int b = 0; b < 8; ++b) vec[b] = v[b] / s2;
    }
}

}  // namespace

// [lut 65536][k9 81][k4 16] floats, built once per context with the host's libm
static int sift_tables(stx_ctx *ctx, const float **d_tables)
{
    const size_t count = 65536 + 81 + 16;
    int err;
    const bool built = ctx->slot_bytes[STX_SLOT_SIFT_TABLES] >= sizeof(float) * count;
    float *d = (float *)stx_workspace(ctx, STX_SLOT_SIFT_TABLES, sizeof(float) * count, &err);
    if (!d) return err;
    if (!built) {
        // no exception may cross the C boundary: nothrow allocation
        std::unique_ptr<float[]> host(new (std::nothrow) float[count]);
        if (!host) return STX_ERR_HOST_OOM;
        for (int vy = 0; vy < 256; ++vy)
            for (int vx = 0; vx < 256; ++vx) {
                const float gX = (float)vx, gY = (float)vy;
                // :974-978; atan of a float argument is the float overload in the reference's C++
                host[(size_t)vy * 256 + vx] = vx != 0 ? atanf(gY / gX) * 180.f / SIFT_PI : 90.f;
            }
        int rc = stx_gaussian_kernel(9, 1.f, host.get() + 65536);
        if (!rc) rc = stx_gaussian_kernel(4, 1.f, host.get() + 65536 + 81);
        if (!rc) rc = (int)hipMemcpy(d, host.get(), sizeof(float) * count, hipMemcpyHostToDevice);
        if (rc) {  // a failed build must not be taken for a finished one by the next call
            ctx->slot_bytes[STX_SLOT_SIFT_TABLES] = 0;
            (void)hipFree(d);
            ctx->slot_ptr[STX_SLOT_SIFT_TABLES] = nullptr;
            return rc;
        }
    }
    *d_tables = d;
    return STX_OK;
}

int stx_launch_sift(stx_ctx *ctx, const uint8_t *d_gx, const uint8_t *d_gy, int w, int h, size_t stride,
                    const float *d_xy, int n, float *d_angle, float *d_desc)
{
    if (n <= 0) return STX_OK;
    const float *tables = nullptr;
    int rc = sift_tables(ctx, &tables);
    if (rc) return rc;
    hipLaunchKernelGGL(sift_kernel, dim3((n + SIFT_FEATS - 1) / SIFT_FEATS), dim3(SIFT_THREADS), 0, ctx->stream,
                       d_gx, d_gy, w, h, stride, reinterpret_cast<const float2 *>(d_xy), n, tables, d_angle, d_desc);
    return (int)hipGetLastError();
}
