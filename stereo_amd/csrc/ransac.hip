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
// hypothesis is the one the serial loop gives, every addition rounded in pair order (fp32 addition does
// not reassociate), but computed by parallel scans: see ransac_reduce_kernel.
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
constexpr int RS_HYP = 4;   // hypotheses per workgroup of the error kernel (RS_HYP * RS_REC <= RS_THREADS)

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
// guard's limits); the reciprocal and its Newton step are shared.  Lanes outside the range take the
// compiler's full divide.  Scalar fp32 throughout, and the file is built with -fno-slp-vectorize: on gfx950
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue at 2.77 cycles per SIMD against 1.49 - 1.63 for the
// scalar forms (profiles/r03_ubench_instruction_rates_more.txt), so a packed pair saves nothing and the
// v_mov_b32 that assemble its operands are pure cost: 0.49 -> 0.42 ms for the scoring call.
__device__ __forceinline__ bool div_in_range(float lo, float hi) { return lo >= 0x1p-40f && hi <= 0x1p40f; }

__device__ __forceinline__ float shared_rcp(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
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
        const float r = shared_rcp(d);
        n0 = quot1(n0, d, r);
        n1 = quot1(n1, d, r);
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
        n0 = quot1(n0, d, r);
        n1 = quot1(n1, d, r);
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

// One (hypothesis, pair): ReprojectionError (:163-182) from the pair's rays x = K1^-1 p, y = K2^-1 p'
// (:167-168, the same for every hypothesis) and the hypothesis record r; depth = Triangulate's two depths.
__device__ __forceinline__ float pair_error(const float *r, const Cams &cams, float2 a1, float x0, float x1, float x2,
                                            float y0, float y1, float y2, float2 &depth)
{
    const float *E = r, *Rinv = r + 9, *Rinvt = r + 18, *Rt = r + 21, *tt = r + 30;
    const float nanv = __builtin_nanf("");
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
    depth = ok ? make_float2(d0, d1) : make_float2(nanv, nanv);
    if (!ok) return 1000.0f;  // :172

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
    return sqrtf(dot3(w0, w1, w2, w0, w1, w2));
}

__global__ __launch_bounds__(RS_THREADS) void ransac_error_kernel(
    const float *__restrict__ rec, int n_hyp, Cams cams, const float2 *__restrict__ p1,
    const float2 *__restrict__ p2, int n, float *__restrict__ errors, float2 *__restrict__ depths)
{
    // RS_HYP hypotheses per workgroup: the pair and its two rays are loaded and computed once, and a
    // quarter as many workgroups stage records and start up (0.44 -> 0.41 ms at 200 x 200 000; 2, 4 and 5
    // per workgroup measure the same).  The early `continue` below is worth 5 %: written as
    // `err = skip ? -1 : pair_error(...)` with one store behind it, the kernel was that much slower.
    __shared__ float recs[RS_HYP][RS_REC];
    const int h0 = blockIdx.y * RS_HYP;
    const int nh = n_hyp - h0 < RS_HYP ? n_hyp - h0 : RS_HYP;
    if ((int)threadIdx.x < nh * RS_REC) (&recs[0][0])[threadIdx.x] = rec[(size_t)h0 * RS_REC + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * RS_THREADS + threadIdx.x;
    if (i >= n) return;
    const float2 a1 = p1[i], a2 = p2[i];
    const float nanv = __builtin_nanf("");
    // :167-168
    const float x0 = row_dot3(cams.k1inv, 0, a1.x, a1.y, 1.0f), x1 = row_dot3(cams.k1inv, 1, a1.x, a1.y, 1.0f),
                x2 = row_dot3(cams.k1inv, 2, a1.x, a1.y, 1.0f);
    const float y0 = row_dot3(cams.k2inv, 0, a2.x, a2.y, 1.0f), y1 = row_dot3(cams.k2inv, 1, a2.x, a2.y, 1.0f),
                y2 = row_dot3(cams.k2inv, 2, a2.x, a2.y, 1.0f);
#pragma unroll 1
    for (int hh = 0; hh < nh; ++hh) {
        const int h = h0 + hh;
        // the chosen eight, which the reference erases before its loop (:205-211): read as scalars (a
        // uniform address) and compared per lane only in the few workgroups whose 256 pairs hold one
        // (the per-lane compare of all eight was 7 % of the kernel)
        const int *excl = reinterpret_cast<const int *>(rec + (size_t)h * RS_REC + 34);
        bool here = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) here |= excl[k] >= 0 && (unsigned)excl[k] / RS_THREADS == blockIdx.x;
        bool skip = false;
        if (here) {
#pragma unroll
            for (int k = 0; k < 8; ++k) skip |= excl[k] == i;
        }
        const size_t o = (size_t)h * n + i;
        if (skip) {
            errors[o] = -1.0f;
            if (depths) depths[o] = make_float2(nanv, nanv);
            continue;
        }
        float2 depth;
        const float err = pair_error(recs[hh], cams, a1, x0, x1, x2, y0, y1, y2, depth);
        if (depths) depths[o] = depth;
        errors[o] = err;
    }
}

// localInliers and the running avgError of :232-236, one workgroup per hypothesis.
//
// fp32 addition does not reassociate, so the sum avgError += error, taken in pair order, looks like a
// serial chain of n dependent additions (rounds 1 and 2 ran it as one: 0.4 ms for 200 x 200 000, as long
// as the error kernel itself).  It is not.  While the running sum s stays inside one binade
// [2^e, 2^(e+1)), s = M ulp with an integer M < 2^24, and fl(s + v) = (M + r) ulp where r is v / ulp
// rounded to nearest -- a number that depends on v and e only, except on an exact tie (v / ulp =
// k + 1/2), where round-to-even makes it k or k + 1 by the parity of M + k.  So every value is a map
// "parity of M coming in -> advance of M", two integers (s0, s1); maps compose associatively (rd_then)
// and a stretch of values collapses to one map by a parallel scan.  The assumption is checked, not
// trusted: the first value at which M + r would reach 2^24 (the sum leaves the binade, or v is too large
// for it) is found from the prefix maps, the eight values of that lane's run are added as floats, one
// after the other, and the rest of the step is mapped again with the new exponent -- from the registers
// it already sits in.  A sum of non-negative values leaves a binade only upwards, so at most 254 times
// whatever the data are (a few dozen times for reprojection errors, most of them within the first few
// hundred values): the kernel has no slow input.  Zero and denormal sums share the grid of the first
// normal binade (ulp 2^-149; the float's bits ARE M there), an infinite sum absorbs every finite value
// (all maps zero).  Ties are rare, so a wave whose tile has none scans plain integers instead of maps.
// Results are those of the serial chain, bit for bit (tests/test_gpu_parity.py::test_inlier_sum_*: ties,
// binade walks, zeros, denormals, overflow).
constexpr int RD_WAVES = 8;                    // waves per hypothesis
constexpr int RD_K = 8;                        // consecutive values per lane
constexpr int RD_TILE = 64 * RD_K;             // values per wave and step
constexpr int RD_STEP = RD_WAVES * RD_TILE;    // values per workgroup and step
constexpr unsigned RD_OVER = 1u << 24;         // M + r >= RD_OVER: the binade ends here
constexpr unsigned RD_SAT = 1u << 27;          // advances saturate (nothing behind an overflow is used)

struct RdMap {
    unsigned s0, s1;  // advance of M when M comes in even / odd
};

// a first, then b
__device__ __forceinline__ RdMap rd_then(RdMap a, RdMap b)
{
    RdMap c;
    c.s0 = a.s0 + ((a.s0 & 1u) ? b.s1 : b.s0);
    c.s1 = a.s1 + ((a.s1 & 1u) ? b.s0 : b.s1);
    return c;
}

__device__ __forceinline__ RdMap rd_saturated(RdMap c)
{
    c.s0 = c.s0 < RD_SAT ? c.s0 : RD_SAT;
    c.s1 = c.s1 < RD_SAT ? c.s1 : RD_SAT;
    return c;
}

// q = v / ulp = (v * sa) * sb exactly (two powers of two: 1 / ulp alone is not always a float), capped
// at 2^24, which ends the binade whatever M is; values that are no inliers leave M alone.
__device__ __forceinline__ float rd_quotient(float v, bool in, float sa, float sb)
{
    const float q = fminf((fabsf(v) * sa) * sb, 0x1p24f);
    return in ? q : 0.0f;
}

// the map of one value from its quotient and rn = rint(q), the even neighbour on a tie
__device__ __forceinline__ RdMap rd_value(float q, float rn)
{
    const float other = (q + q) - rn;  // on a tie: the odd neighbour (2 q = 2 k + 1)
    RdMap m;
    m.s0 = (unsigned)rn;
    m.s1 = (unsigned)(fabsf(q - rn) == 0.5f ? other : rn);
    return m;
}

// the value of lane - 2^k (0 where there is none): row shifts, then the row broadcasts of a wave scan
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned rd_from_left(unsigned x)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xF, false);
}

template <int CTRL, int ROWS>
__device__ __forceinline__ void rd_scan_step(RdMap &a)
{
    RdMap l;
    l.s0 = rd_from_left<CTRL, ROWS>(a.s0);
    l.s1 = rd_from_left<CTRL, ROWS>(a.s1);
    a = rd_saturated(rd_then(l, a));  // (0, 0) is the identity
}

// inclusive scan over the wave, in lane order: row_shr:1,2,4,8, row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3
__device__ __forceinline__ void rd_scan(RdMap &a)
{
    rd_scan_step<0x111, 0xF>(a);
    rd_scan_step<0x112, 0xF>(a);
    rd_scan_step<0x114, 0xF>(a);
    rd_scan_step<0x118, 0xF>(a);
    rd_scan_step<0x142, 0xA>(a);
    rd_scan_step<0x143, 0xC>(a);
}

__device__ __forceinline__ unsigned rd_scan_sum(unsigned x, unsigned cap)
{
    auto sat = [cap](unsigned v) { return v < cap ? v : cap; };
    x = sat(x + rd_from_left<0x111, 0xF>(x));
    x = sat(x + rd_from_left<0x112, 0xF>(x));
    x = sat(x + rd_from_left<0x114, 0xF>(x));
    x = sat(x + rd_from_left<0x118, 0xF>(x));
    x = sat(x + rd_from_left<0x142, 0xA>(x));
    x = sat(x + rd_from_left<0x143, 0xC>(x));
    return x;
}

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // rows of `errors` start anywhere

__global__ __launch_bounds__(RD_WAVES * 64) void ransac_reduce_kernel(const float *__restrict__ errors, int n,
                                                                      float threshold,
                                                                      int32_t *__restrict__ inliers,
                                                                      float *__restrict__ err_sum)
{
    // inclusive prefix maps per lane, double buffered by pass (one barrier per pass)
    __shared__ unsigned pm0[2][RD_WAVES][64], pm1[2][RD_WAVES][64];
    __shared__ int wave_count[RD_WAVES];
    const int h = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const float *e = errors + (size_t)h * n;
    const int myrun = wave * 64 + lane;  // this lane's run of RD_K values inside a step

    auto load = [&](int base, float (&dst)[RD_K]) __attribute__((always_inline)) {
        if (base + RD_K <= n) {
            const f32x4u a = *reinterpret_cast<const f32x4u *>(e + base);
            const f32x4u b = *reinterpret_cast<const f32x4u *>(e + base + 4);
            dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
            dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < RD_K; ++k) dst[k] = base + k < n ? e[base + k] : -1.0f;
        }
    };

    // every thread carries the same (s, skip): functions of shared data only; the inliers are simply
    // counted, each thread its own
    float s = 0.0f;
    int count = 0, pass = 0;
    // One step: RD_STEP values from pos on, this thread's run of them in cur.
    auto step = [&](int pos, const float (&cur)[RD_K]) __attribute__((always_inline)) {
        // One pass per binade the sum visits inside this step (nearly always one): runs below `skip`
        // have been added already.
        int skip = 0;
        bool in[RD_K];
#pragma unroll
        for (int k = 0; k < RD_K; ++k) {
            in[k] = cur[k] >= 0.0f && cur[k] < threshold;
            count += in[k] ? 1 : 0;
        }
        while (skip < RD_WAVES * 64) {
            const unsigned sbits = __float_as_uint(s);
            const int eb = (int)((sbits >> 23) & 0xffu);
            const int grid = eb > 1 ? eb : 1;  // zero and denormal sums live on the grid of binade 1
            // 1 / ulp = 2^(150 - grid) as two factors that are floats; inf and NaN sums absorb: q = 0
            const int ea = 150 - grid > 100 ? 100 : 150 - grid < -100 ? -100 : 150 - grid;
            const float sa = eb == 255 ? 0.0f : __uint_as_float((unsigned)(127 + ea) << 23);
            const float sb = __uint_as_float((unsigned)(127 + (150 - grid - ea)) << 23);
            const bool live = myrun >= skip;
            float q[RD_K], rn[RD_K], far = 0.0f;
#pragma unroll
            for (int k = 0; k < RD_K; ++k) {
                q[k] = rd_quotient(cur[k], live && in[k], sa, sb);
                rn[k] = __builtin_rintf(q[k]);
                far = fmaxf(far, fabsf(q[k] - rn[k]));
            }
            RdMap a = {0u, 0u};
            if (!__any(far == 0.5f)) {  // no tie in this wave's tile: plain sums
#pragma unroll
                for (int k = 0; k < RD_K; ++k) a.s0 += (unsigned)rn[k];
                a.s0 = a.s1 = rd_scan_sum(a.s0, RD_SAT);
            } else {
#pragma unroll
                for (int k = 0; k < RD_K; ++k) a = rd_then(a, rd_value(q[k], rn[k]));
                rd_scan(a);
            }
            const int buf = pass & 1;
            ++pass;
            pm0[buf][wave][lane] = a.s0;
            pm1[buf][wave][lane] = a.s1;
            __syncthreads();

            // M: the sum in units of its grid (the implicit bit only where there is one)
            unsigned M = (sbits & 0x7fffffu) | (eb ? 0x800000u : 0u);
            const unsigned below = (unsigned)(grid - 1) << 23;  // bits = below + M, for denormals too
            int cw = -1;  // the wave whose tile holds the end of the binade
#pragma unroll
            for (int w = 0; w < RD_WAVES; ++w) {
                if (cw < 0) {
                    const unsigned adv = (M & 1u) ? pm1[buf][w][63] : pm0[buf][w][63];
                    if (M + adv >= RD_OVER)
                        cw = w;
                    else
                        M += adv;
                }
            }
            if (cw < 0) {
                s = __uint_as_float(below + M);
                break;
            }
            const unsigned adv = (M & 1u) ? pm1[buf][cw][lane] : pm0[buf][cw][lane];
            const unsigned long long over = __ballot(M + adv >= RD_OVER);  // lane 63 at least
            const int ls = __builtin_ctzll(over);
            if (ls > 0) M += (M & 1u) ? pm1[buf][cw][ls - 1] : pm0[buf][cw][ls - 1];
            s = __uint_as_float(below + M);
            // that lane's run as the reference adds it (an outlier adds +0: s + 0 = s)
            const int run = __builtin_amdgcn_readfirstlane(cw * 64 + ls);
            float v[RD_K];
#pragma unroll
            for (int k = 0; k < RD_K; ++k) {
                const int j = pos + run * RD_K + k;
                v[k] = j < n ? e[j] : -1.0f;
            }
#pragma unroll
            for (int k = 0; k < RD_K; ++k) s = s + (v[k] >= 0.0f && v[k] < threshold ? v[k] : 0.0f);
            skip = run + 1;
        }
    };
    // three register sets in rotation: the loads of a step are issued two steps ahead of their use
    // (copying the sets along would make every step wait for the loads it has just issued)
    float b0[RD_K], b1[RD_K], b2[RD_K];
    load(myrun * RD_K, b0);
    load(RD_STEP + myrun * RD_K, b1);
    for (int pos = 0; pos < n;) {
        load(pos + 2 * RD_STEP + myrun * RD_K, b2);
        step(pos, b0);
        pos += RD_STEP;
        if (pos >= n) break;
        load(pos + 2 * RD_STEP + myrun * RD_K, b0);
        step(pos, b1);
        pos += RD_STEP;
        if (pos >= n) break;
        load(pos + 2 * RD_STEP + myrun * RD_K, b1);
        step(pos, b2);
        pos += RD_STEP;
    }
    count = (int)rd_scan_sum((unsigned)count, 0xffffffffu);  // lane 63: the wave's inliers
    if (lane == 63) wave_count[wave] = count;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < RD_WAVES; ++w) total += wave_count[w];
        inliers[h] = total;
        err_sum[h] = s;
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
        const dim3 grid((unsigned)((n + RS_THREADS - 1) / RS_THREADS), (unsigned)((n_hyp + RS_HYP - 1) / RS_HYP));
        hipLaunchKernelGGL(ransac_error_kernel, grid, dim3(RS_THREADS), 0, ctx->stream, d_rec, n_hyp, cams,
                           reinterpret_cast<const float2 *>(d_p1), reinterpret_cast<const float2 *>(d_p2), n,
                           d_errors, reinterpret_cast<float2 *>(d_depths));
        STX_HIP(hipGetLastError());
    }
    return stx_launch_inlier_sum(ctx, d_errors, n_hyp, n, threshold, d_inliers, d_err_sum);
}

int stx_launch_inlier_sum(stx_ctx *ctx, const float *d_errors, int n_hyp, int n, float threshold,
                          int32_t *d_inliers, float *d_err_sum)
{
    if (n_hyp <= 0) return STX_OK;
    if (n > 0x7fffffff - 4 * RD_STEP) return STX_ERR_TOO_LARGE;  // (the kernel looks two steps ahead in int)
    hipLaunchKernelGGL(ransac_reduce_kernel, dim3((unsigned)n_hyp), dim3(RD_WAVES * 64), 0, ctx->stream, d_errors, n,
                       threshold, d_inliers, d_err_sum);
    return (int)hipGetLastError();
}
