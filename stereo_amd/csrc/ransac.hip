// RANSAC hypothesis scoring for gfx950: the pair loop of FindFundamentalMatrixWithRANSAC
// (stereo/Stereography.cpp:226-238) for every hypothesis of a batch at once -- per (hypothesis,
// pair) ReprojectionError (:163-182) -> Triangulate (:383-443) -> LindstromOptimisation
// (:354-381), then per hypothesis the inlier count and the in-order sum of the inlier errors.
//
// What stays on the host: drawing the eight pairs, FindFundamentalMatrix and the singular value
// decompositions of DecomposeEssentialMatrix (:301-353) -- 3x3 / 8x9 Eigen BDCSVD calls, a few
// hundred per image pair.  Their outputs come in as one record per hypothesis (stereo_hip.h).
// ReprojectionError takes E by value (:163), so the decomposition inside Triangulate (:398) sees
// the same matrix for every pair of a hypothesis: it is a per-hypothesis input as well.
//
// Arithmetic: fp32, every multiply, add, divide and sqrt its own correctly rounded operation
// (-ffp-contract=off); three-term sums of fixed-size Eigen expressions as a0 + (a1 + a2), Eigen's
// unrolled reduction (the order the rectification kernel already follows).  The error sum of a
// hypothesis is taken in pair order by a serial chain (fp32 addition does not reassociate), one
// workgroup of two waves per hypothesis.
//
// Bound: compute (about 30 IEEE divisions and square roots per evaluation); 16 B of pair data
// in and 4 B of error out per evaluation.
#include "stx_internal.h"

#include <math.h>
#include <memory>
#include <new>

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_REC = 48;  // floats per device record

// device record: E[9] Rinv[9] Rinv*t[3] R'[9] t'[3] ok exclude[8] pad
struct Cams {
    float k1inv[9], k2inv[9], k2[9];
};

// Three-term sums of fixed-size Eigen expressions (Matrix3f * Vector3f, dot(), norm()) follow
// Eigen's unrolled reduction, a0 + (a1 + a2), as in the rectification kernel.
__device__ __forceinline__ float row_dot3(const float *m, int r, float v0, float v1, float v2)
{
    const float a0 = m[r * 3] * v0, a1 = m[r * 3 + 1] * v1, a2 = m[r * 3 + 2] * v2;
    return a0 + (a1 + a2);
}

__device__ __forceinline__ float col_dot3(const float *m, int c, float v0, float v1, float v2)
{
    const float a0 = m[c] * v0, a1 = m[3 + c] * v1, a2 = m[6 + c] * v2;
    return a0 + (a1 + a2);
}

__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2)
{
    const float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    return p0 + (p1 + p2);
}

// The products through the dynamic-size selector S (MatrixXf, :356-363) take Eigen's run-time
// paths: three terms accumulated in index order.
__device__ __forceinline__ float row_dot3_seq(const float *m, int r, float v0, float v1, float v2)
{
    float s = m[r * 3] * v0;
    s = s + m[r * 3 + 1] * v1;
    s = s + m[r * 3 + 2] * v2;
    return s;
}

__device__ __forceinline__ float col_dot3_seq(const float *m, int c, float v0, float v1, float v2)
{
    float s = m[c] * v0;
    s = s + m[3 + c] * v1;
    s = s + m[6 + c] * v2;
    return s;
}

// ---- divisions ----------------------------------------------------------------------------
// Triangulate and ReprojectionError divide whole vectors by one scalar (x / x(2), u / |u|, ...):
// 15 of the kernel's 20 divisions share a denominator with one or two others, and three of them
// are x / x.  The compiler's expansion of an IEEE fp32 divide on gfx950 is v_div_scale x2, v_rcp, a
// Newton step on the reciprocal, two residual corrections of the quotient, v_div_fmas, v_div_fixup.
// When |n| and |d| lie in [2^-40, 2^40] neither v_div_scale rescales anything, v_div_fmas is a plain
// fma and v_div_fixup returns its input, so the same fma chain without those three instructions gives
// the same bits (the argument and the guard of divide_pair in rectify.hip, whose test walks the
// guard's limits); the reciprocal and its Newton step are shared, two chains run as packed fp32.
// Lanes outside the range take the compiler's full divide.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool div_in_range(float lo, float hi) { return lo >= 0x1p-40f && hi <= 0x1p40f; }

__device__ __forceinline__ float shared_rcp(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

__device__ __forceinline__ f32x2 quot2(f32x2 n, float d, float r)
{
    const f32x2 nd = {-d, -d}, rr = {r, r};
    f32x2 t = n * rr;
    f32x2 res = __builtin_elementwise_fma(nd, t, n);
    t = __builtin_elementwise_fma(res, rr, t);
    res = __builtin_elementwise_fma(nd, t, n);
    return __builtin_elementwise_fma(res, rr, t);
}

__device__ __forceinline__ float quot1(float n, float d, float r)
{
    float t = n * r;
    float res = __builtin_fmaf(-d, t, n);
    t = __builtin_fmaf(res, r, t);
    res = __builtin_fmaf(-d, t, n);
    return __builtin_fmaf(res, r, t);
}

// n0 / d and n1 / d, correctly rounded
__device__ __forceinline__ void divide2(float &n0, float &n1, float d)
{
    const float a0 = fabsf(n0), a1 = fabsf(n1), ad = fabsf(d);
    if (div_in_range(fminf(fminf(a0, a1), ad), fmaxf(fmaxf(a0, a1), ad))) {
        const f32x2 q = quot2(f32x2{n0, n1}, d, shared_rcp(d));
        n0 = q.x;
        n1 = q.y;
    } else {
        n0 = n0 / d;
        n1 = n1 / d;
    }
}

// n0 / d, n1 / d and n2 / d, correctly rounded
__device__ __forceinline__ void divide3(float &n0, float &n1, float &n2, float d)
{
    const float a0 = fabsf(n0), a1 = fabsf(n1), a2 = fabsf(n2), ad = fabsf(d);
    if (div_in_range(fminf(fminf(a0, a1), fminf(a2, ad)), fmaxf(fmaxf(a0, a1), fmaxf(a2, ad)))) {
        const float r = shared_rcp(d);
        const f32x2 q = quot2(f32x2{n0, n1}, d, r);
        n0 = q.x;
        n1 = q.y;
        n2 = quot1(n2, d, r);
    } else {
        n0 = n0 / d;
        n1 = n1 / d;
        n2 = n2 / d;
    }
}

// x / x: exactly 1 for every finite non-zero x (denormals included: the quotient of equal numbers is
// representable, so the correctly rounded result is it), NaN for 0, +-inf and NaN
__device__ __forceinline__ float self_quotient(float x)
{
    return (x - x == 0.0f && x != 0.0f) ? 1.0f : __builtin_nanf("");
}

__global__ __launch_bounds__(RS_THREADS) void ransac_error_kernel(
    const float *__restrict__ rec, Cams cams, const float2 *__restrict__ p1, const float2 *__restrict__ p2, int n,
    float *__restrict__ errors, float2 *__restrict__ depths)
{
    __shared__ float r[RS_REC];
    const int h = blockIdx.y;
    if (threadIdx.x < RS_REC) r[threadIdx.x] = rec[(size_t)h * RS_REC + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * RS_THREADS + threadIdx.x;
    if (i >= n) return;
    const float *E = r, *Rinv = r + 9, *Rinvt = r + 18, *Rt = r + 21, *tt = r + 30;
    const int *excl = reinterpret_cast<const int *>(r + 34);
    bool skip = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) skip |= excl[k] == i;
    const size_t o = (size_t)h * n + i;
    const float nanv = __builtin_nanf("");
    if (skip) {
        errors[o] = -1.0f;
        if (depths) depths[o] = make_float2(nanv, nanv);
        return;
    }
    const float2 a1 = p1[i], a2 = p2[i];
    // :167-168
    float x0 = row_dot3(cams.k1inv, 0, a1.x, a1.y, 1.0f), x1 = row_dot3(cams.k1inv, 1, a1.x, a1.y, 1.0f),
          x2 = row_dot3(cams.k1inv, 2, a1.x, a1.y, 1.0f);
    float y0 = row_dot3(cams.k2inv, 0, a2.x, a2.y, 1.0f), y1 = row_dot3(cams.k2inv, 1, a2.x, a2.y, 1.0f),
          y2 = row_dot3(cams.k2inv, 2, a2.x, a2.y, 1.0f);

    // ---- LindstromOptimisation, :354-381 ------------------------------------
    {
        const float e00 = E[0], e01 = E[1], e10 = E[3], e11 = E[4];
        float n0 = row_dot3_seq(E, 0, y0, y1, y2), n1 = row_dot3_seq(E, 1, y0, y1, y2);
        float m0 = col_dot3_seq(E, 0, x0, x1, x2), m1 = col_dot3_seq(E, 1, x0, x1, x2);
        const float q0 = col_dot3(E, 0, x0, x1, x2), q1 = col_dot3(E, 1, x0, x1, x2), q2 = col_dot3(E, 2, x0, x1, x2);
        const float r0 = n0 * e00 + n1 * e10, r1 = n0 * e01 + n1 * e11;
        const float a = r0 * m0 + r1 * m1;
        const float b = 0.5f * ((n0 * n0 + n1 * n1) + (m0 * m0 + m1 * m1));
        const float c = dot3(q0, q1, q2, y0, y1, y2);
        const float d = sqrtf(b * b - a * c);
        const float lambda = c / (b + d);
        float xd0 = lambda * n0, xd1 = lambda * n1;
        float pd0 = lambda * m0, pd1 = lambda * m1;
        n0 = n0 - (e00 * pd0 + e01 * pd1);
        n1 = n1 - (e10 * pd0 + e11 * pd1);
        m0 = m0 - (e00 * xd0 + e01 * xd1);
        m1 = m1 - (e10 * xd0 + e11 * xd1);
        const float sx = (xd0 * n0 + xd1 * n1) / (n0 * n0 + n1 * n1);
        xd0 = sx * n0;
        xd1 = sx * n1;
        const float sp = (pd0 * m0 + pd1 * m1) / (m0 * m0 + m1 * m1);
        pd0 = sp * m0;
        pd1 = sp * m1;
        x0 = x0 - xd0;
        x1 = x1 - xd1;
        y0 = y0 - pd0;
        y1 = y1 - pd1;
    }

    // ---- Triangulate, :398-431 ------------------------------------------------
    bool ok = r[33] != 0.0f;
    float d0 = nanv, d1 = nanv;
    if (ok) {
        float nx0 = x0, nx1 = x1;
        divide2(nx0, nx1, x2);
        const float nx2 = self_quotient(x2);
        float v0 = y0, v1 = y1;
        divide2(v0, v1, y2);
        float v2 = self_quotient(y2);
        float u0 = row_dot3(Rt, 0, nx0, nx1, nx2), u1 = row_dot3(Rt, 1, nx0, nx1, nx2),
              u2 = row_dot3(Rt, 2, nx0, nx1, nx2);
        const float un = sqrtf(dot3(u0, u1, u2, u0, u1, u2)), vn = sqrtf(dot3(v0, v1, v2, v0, v1, v2));
        divide3(u0, u1, u2, un);
        divide3(v0, v1, v2, vn);
        const float a = dot3(u0, u1, u2, tt[0], tt[1], tt[2]), b = dot3(u0, u1, u2, u0, u1, u2),
                    c = dot3(u0, u1, u2, v0, v1, v2), d = dot3(v0, v1, v2, tt[0], tt[1], tt[2]),
                    e = dot3(v0, v1, v2, v0, v1, v2);
        const float g = c * c - b * e;
        if ((double)fabsf(g) < 1e-9) {  // :418
            ok = false;
        } else if ((double)fabsf(c) < 1e-9) {  // :422
            d1 = (a * c - b * d) / (c * c - b * e);
            d0 = (c * d1 - a) / b;
        } else {
            d0 = (a * e - c * d) / (c * c - b * e);
            d1 = (a + b * d0) / c;
        }
    }
    if (depths) depths[o] = ok ? make_float2(d0, d1) : make_float2(nanv, nanv);
    if (!ok) {
        errors[o] = 1000.0f;  // :172
        return;
    }

    // ---- ReprojectionError, :175-179 ----------------------------------------
    const float s0 = x0 * d1, s1 = x1 * d1, s2 = x2 * d1;
    float t0 = row_dot3(Rinv, 0, s0, s1, s2) - Rinvt[0];
    float t1 = row_dot3(Rinv, 1, s0, s1, s2) - Rinvt[1];
    float t2 = row_dot3(Rinv, 2, s0, s1, s2) - Rinvt[2];
    const float z = t2;
    divide2(t0, t1, z);
    t2 = self_quotient(z);
    const float w0 = row_dot3(cams.k2, 0, t0, t1, t2) - a1.x, w1 = row_dot3(cams.k2, 1, t0, t1, t2) - a1.y,
                w2 = row_dot3(cams.k2, 2, t0, t1, t2) - 1.0f;
    errors[o] = sqrtf(dot3(w0, w1, w2, w0, w1, w2));
}

// One workgroup of two waves per hypothesis: localInliers and the running avgError of :232-236.
// fp32 addition does not reassociate, so the sum is a serial chain in pair order -- but only the
// inliers are on it.  Per tile of RD_TILE errors (a) one wave compacts the inliers, in order,
// into LDS with ballot / mbcnt prefix counts, all lanes working, while (b) the other wave runs
// the chain over the previous tile's compacted values, every lane redundantly, sixteen values
// per step with the next step's LDS reads in flight.  The next tile's global loads are issued
// before (a), so neither wave waits on HBM once the pipeline runs.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (`sum` is tied into both asm blocks so that the adds between them stay between them)
// reads the group that starts OFF bytes behind addr
template <int OFF>
__device__ __forceinline__ void lds_issue4(unsigned addr, f32x4 &r0, f32x4 &r1, f32x4 &r2, f32x4 &r3, float &sum)
{
    asm volatile("ds_read_b128 %0, %5 offset:%6\n\tds_read_b128 %1, %5 offset:%7\n\t"
                 "ds_read_b128 %2, %5 offset:%8\n\tds_read_b128 %3, %5 offset:%9"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "+v"(sum)
                 : "v"(addr), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48)
                 : "memory");
}
// the reads above are complete after this
__device__ __forceinline__ void lds_wait(f32x4 &r0, f32x4 &r1, f32x4 &r2, f32x4 &r3, float &sum)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(sum));
}
__device__ __forceinline__ void chain16(float &sum, const f32x4 &v0, const f32x4 &v1, const f32x4 &v2,
                                        const f32x4 &v3)
{
    sum = sum + v0.x; sum = sum + v0.y; sum = sum + v0.z; sum = sum + v0.w;
    sum = sum + v1.x; sum = sum + v1.y; sum = sum + v1.z; sum = sum + v1.w;
    sum = sum + v2.x; sum = sum + v2.y; sum = sum + v2.z; sum = sum + v2.w;
    sum = sum + v3.x; sum = sum + v3.y; sum = sum + v3.z; sum = sum + v3.w;
}

constexpr int RD_K = 8;               // values per lane per tile
constexpr int RD_TILE = 64 * RD_K;

__global__ __launch_bounds__(128) void ransac_reduce_kernel(const float *__restrict__ errors, int n, float threshold,
                                                            int32_t *__restrict__ inliers,
                                                            float *__restrict__ err_sum)
{
    // two waves per hypothesis: wave 1 compacts tile t into list[t & 1] while wave 0 runs the
    // chain over tile t - 1 in the other buffer; one barrier per tile
    __shared__ __align__(16) float list[2][RD_TILE + 48];
    __shared__ int fill[2];
    const int h = blockIdx.x, lane = threadIdx.x & 63;
    const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 1;
    const float *e = errors + (size_t)h * n;
    const int tiles = (n + RD_TILE - 1) / RD_TILE;
    float sum = 0.0f;
    int count = 0;
    float next[RD_K];
    if (producer) {
#pragma unroll
        for (int k = 0; k < RD_K; ++k) {
            const int j = k * 64 + lane;
            next[k] = j < n ? e[j] : -1.0f;
        }
    }
    for (int t = 0; t <= tiles; ++t) {
        if (producer) {
            if (t < tiles) {
                float cur[RD_K];
#pragma unroll
                for (int k = 0; k < RD_K; ++k) cur[k] = next[k];
#pragma unroll
                for (int k = 0; k < RD_K; ++k) {
                    const int j = (t + 1) * RD_TILE + k * 64 + lane;
                    next[k] = j < n ? e[j] : -1.0f;
                }
                // (a) in-order compaction of this tile's inliers
                float *dst = list[t & 1];
                int m = 0;
#pragma unroll
                for (int k = 0; k < RD_K; ++k) {
                    const bool in = cur[k] >= 0.0f && cur[k] < threshold;
                    const unsigned long long mask = __ballot(in);
                    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    if (in) dst[m + before] = cur[k];
                    m += __popcll(mask);
                }
                // whole groups are read; a group past the end is all +0, which leaves the
                // (non-negative) sum as it is
                if (lane < 48) dst[m + lane] = 0.0f;
                if (lane == 0) fill[t & 1] = m;
            }
        } else if (t > 0) {
            // (b) the chain, 16 values per group.  The next group's four LDS reads are issued
            // BEFORE the current group's adds and waited for after them: written as asm because
            // the compiler otherwise sinks the reads to their uses and every group pays the LDS
            // round trip.
            const int m = fill[(t - 1) & 1];
            if (m > 0) {
                const unsigned lbase =
                    (unsigned)(size_t)(__attribute__((address_space(3))) float *)list[(t - 1) & 1];
                const int groups = (m + 15) >> 4;
                f32x4 a0, a1, a2, a3, b0, b1, b2, b3;
                // (three groups of zeros follow the list, so the reads one and two groups ahead
                // need no clamping: immediate offsets from one address that advances per step)
                unsigned at = lbase;
                lds_issue4<0>(at, a0, a1, a2, a3, sum);
                lds_wait(a0, a1, a2, a3, sum);
                for (int g = 0; g < groups; g += 2) {
                    lds_issue4<64>(at, b0, b1, b2, b3, sum);
                    chain16(sum, a0, a1, a2, a3);
                    lds_wait(b0, b1, b2, b3, sum);
                    lds_issue4<128>(at, a0, a1, a2, a3, sum);
                    chain16(sum, b0, b1, b2, b3);
                    lds_wait(a0, a1, a2, a3, sum);
                    at += 128u;
                }
                count += m;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        inliers[h] = count;
        err_sum[h] = sum;
    }
}

float host_row_dot3(const float *m, int r, const float v[3])
{
    volatile float a = m[r * 3] * v[0], b = m[r * 3 + 1] * v[1], c = m[r * 3 + 2] * v[2];
    volatile float s = b + c;
    return a + s;
}

}  // namespace

int stx_launch_ransac_score(stx_ctx *ctx, const float *hyp, int n_hyp, const float K1[9], const float K2[9],
                            const float *d_p1, const float *d_p2, int n, const int32_t *exclude, float threshold,
                            int32_t *d_inliers, float *d_err_sum, float *d_errors, float *d_depths)
{
    if (n_hyp <= 0) return STX_OK;
    int err;
    // [records n_hyp*RS_REC][errors n_hyp*n when the caller brings none]
    const size_t rec_bytes = sizeof(float) * RS_REC * (size_t)n_hyp;
    const size_t err_bytes = d_errors ? 0 : sizeof(float) * (size_t)n_hyp * (size_t)(n > 0 ? n : 1);
    float *d_rec = (float *)stx_workspace(ctx, STX_SLOT_RANSAC, rec_bytes + err_bytes, &err);
    if (!d_rec) return err;
    if (!d_errors) d_errors = d_rec + RS_REC * (size_t)n_hyp;

    Cams cams;
    int rc = stx_invert_homography(K1, cams.k1inv);  // K1.inverse(), :167
    if (rc) return rc;
    rc = stx_invert_homography(K2, cams.k2inv);  // :168
    if (rc) return rc;
    for (int k = 0; k < 9; ++k) cams.k2[k] = K2[k];
    // no exception may cross the C boundary: nothrow allocation, value-initialised
    std::unique_ptr<float[]> rec(new (std::nothrow) float[(size_t)RS_REC * n_hyp]());
    if (!rec) return STX_ERR_HOST_OOM;
    for (int hh = 0; hh < n_hyp; ++hh) {
        const float *src = hyp + (size_t)hh * STX_HYP_FLOATS;
        float *dst = rec.get() + (size_t)hh * RS_REC;
        for (int k = 0; k < 9; ++k) dst[k] = src[k];
        rc = stx_invert_homography(src + 9, dst + 9);  // R.inverse(), :175
        if (rc) return rc;
        for (int k = 0; k < 3; ++k) dst[18 + k] = host_row_dot3(dst + 9, k, src + 18);  // R.inverse() * t
        for (int k = 0; k < 12; ++k) dst[21 + k] = src[21 + k];
        dst[33] = src[33];
        int32_t *ex = reinterpret_cast<int32_t *>(dst + 34);
        for (int k = 0; k < 8; ++k) ex[k] = exclude ? exclude[(size_t)hh * 8 + k] : -1;
    }
    STX_HIP(hipMemcpyAsync(d_rec, rec.get(), rec_bytes, hipMemcpyHostToDevice, ctx->stream));
    // the records were staged from pageable memory; wait so `rec` may go out of scope
    STX_HIP(hipStreamSynchronize(ctx->stream));
    if (n > 0) {
        const dim3 grid((unsigned)((n + RS_THREADS - 1) / RS_THREADS), (unsigned)n_hyp);
        hipLaunchKernelGGL(ransac_error_kernel, grid, dim3(RS_THREADS), 0, ctx->stream, d_rec, cams,
                           reinterpret_cast<const float2 *>(d_p1), reinterpret_cast<const float2 *>(d_p2), n,
                           d_errors, reinterpret_cast<float2 *>(d_depths));
        STX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(ransac_reduce_kernel, dim3((unsigned)n_hyp), dim3(128), 0, ctx->stream, d_errors, n, threshold,
                       d_inliers, d_err_sum);
    return (int)hipGetLastError();
}
