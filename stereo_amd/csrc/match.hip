// Brute-force descriptor matching for gfx950: MatchDescriptors
// (stereo/Features.cpp:1020-1079) with DistanceBetweenDescriptors (:1007-1018)
// and L2_norm (:806-815).
//
// Bit-exactness rules (this file is compiled with -ffp-contract=off):
//   dist(a,b) = sqrt( ((..((d0*d0) + d1*d1) + ..) + d127*d127) ),  dk = a[k] - b[k]
//   every -, *, + is one rounded fp32 operation, k ascending, one accumulator;
//   sqrt is the correctly rounded fp32 square root.
//   closest = first j attaining the minimum of dist; second = the previous
//   holder of the running minimum (first argmin over j < closest).
#include <cstdlib>

#include "stx_internal.h"

namespace {

constexpr int K = STX_DESC_LEN;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// squared distance in the reference's operation order (the value L2_norm holds
// in `norm` just before its sqrt, Features.cpp:809-813)
__device__ __forceinline__ float ref_norm2(const float *__restrict__ a, const float *__restrict__ b)
{
    float norm = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float d = a[k] - b[k];
        const float sq = d * d;
        norm = norm + sq;
    }
    return norm;
}

// Features.cpp:1049-1075: decide the outcome of query i from the scan result.
__device__ __forceinline__ uint8_t match_status(int closest, int second, float min_dist,
                                                float second_dist, float score_i,
                                                const float *__restrict__ score2,
                                                float dist_limit)
{
    if (closest == -1 || second == -1 || closest == second) return STX_MATCH_FAILED;
    if (fabsf(score_i - score2[closest]) > dist_limit) return STX_MATCH_SCORE;
    const float ratio = min_dist / second_dist;
    // NN_RATIO is the double 0.8 (Features.h:33); (double)ratio < 0.8 <=> ratio < 0.8f
    // because 0.8f is the smallest float above 0.8.  NaN fails.
    return ratio < 0.8f ? STX_MATCH_ACCEPTED : STX_MATCH_RATIO;
}

// ---- rows of list2 through the scalar data path, double buffered ------------------------
// Every lane of a wave needs the same row, so a row arrives as eight s_load_dwordx16 and its
// elements are scalar operands of the lanes' vector instructions.  Scalar loads return out of order:
// the only wait is lgkmcnt(0), and the compiler placed each load right in front of its wait --
// every 64-byte chunk's full latency exposed, eight times per row (42 % vector utilisation in the
// overflow pass, PMC).  Here the NEXT chunk is requested as soon as the current one has arrived
// and travels while the current one is consumed (32 vector instructions); the chunk behind a
// row's last one is the next row's first.  (Loads only: nothing is ever written through the
// scalar cache.)
typedef float f32x16s __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16s row_chunk_request(const float *p)
{
    f32x16s v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p));
    return v;
}

// the chunk requested last has arrived after this (ties the wait to the value the consumer reads)
__device__ __forceinline__ void row_chunk_arrived(f32x16s &v)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v));
}

// Squared distance of the query (64 packed pairs in registers) to the row whose first chunk has been
// requested into `cur`; `next_row` is where the following request goes once this row's chunks are
// through (the next row, or any valid row behind the last one).  Reference operation order:
// differences and squares as packed fp32 (each half rounded as the scalar instruction rounds it), the
// sum ONE accumulator in ascending k.
__device__ __forceinline__ float row_norm2(const f32x2 (&a)[K / 2], const float *row, const float *next_row,
                                           f32x16s &cur)
{
    float norm = 0.0f;
#pragma unroll
    for (int c = 0; c < K / 16; ++c) {
        row_chunk_arrived(cur);
        f32x16s now = cur;
        // (`now` passes through the request so that its consumers cannot be scheduled in front of it)
        asm volatile("s_load_dwordx16 %0, %2, 0x0"
                     : "=&s"(cur), "+s"(now)
                     : "s"(c + 1 < K / 16 ? row + 16 * (c + 1) : next_row));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 b = {now[2 * k], now[2 * k + 1]};
            const f32x2 d = a[8 * c + k] - b;
            const f32x2 sq = d * d;
            norm = norm + sq.x;
            norm = norm + sq.y;
        }
    }
    return norm;
}

// ---------------------------------------------------------------------------
// Exact scan: one lane per query, the query's 128 floats in registers, list2
// rows streamed through the scalar data path (every lane reads the same row).
// The sqrt is taken only when the squared norm beats the squared norm of the
// current minimum: sqrt is monotonic, so a candidate whose squared norm is not
// smaller cannot have a strictly smaller distance.
// ---------------------------------------------------------------------------
constexpr int EXACT_THREADS = 64;

__global__ __launch_bounds__(EXACT_THREADS) void match_exact_kernel(
    const float *__restrict__ desc1, const float *__restrict__ score1, int n1,
    const float *__restrict__ desc2, const float *__restrict__ score2, int n2, float dist_limit,
    const int32_t *__restrict__ query_list, const int *__restrict__ n_queries_dev, int n_queries_host,
    int32_t *__restrict__ closest_out,
    int32_t *__restrict__ second_out, float *__restrict__ min_dist_out,
    float *__restrict__ second_dist_out, uint8_t *__restrict__ status_out)
{
    // the overflow pass reads its query count from device memory (written by the replay kernel)
    const int n_queries = n_queries_dev ? *n_queries_dev : n_queries_host;
    if ((int)(blockIdx.x * EXACT_THREADS) >= n_queries) return;
    const int slot = blockIdx.x * EXACT_THREADS + threadIdx.x;
    const bool active = slot < n_queries;
    // inactive lanes shadow the last query so that the row loop stays uniform
    const int qslot = active ? slot : n_queries - 1;
    const int i = query_list ? query_list[qslot] : qslot;

    // differences and squares two elements at a time (packed fp32, rounded as the scalar instructions
    // round), the sum one accumulator in ascending k: 4 vector instructions per two elements, not 6
    f32x2 a[K / 2];
    const float4 *arow = reinterpret_cast<const float4 *>(desc1 + (size_t)i * K);
#pragma unroll
    for (int k = 0; k < K / 4; ++k) {
        const float4 v = arow[k];
        a[2 * k + 0] = f32x2{v.x, v.y};
        a[2 * k + 1] = f32x2{v.z, v.w};
    }

    int closest = -1, second = -1;
    float min_dist = -1.0f;  // Features.cpp:1035
    float min_norm2 = 0.0f;
    f32x16s chunk = {};
    if (n2 > 0) chunk = row_chunk_request(desc2);  // (n2 == 0: nothing to read, desc2 may be null)
    for (int j = 0; j < n2; ++j) {
        const float *row = desc2 + (size_t)j * K;
        const float norm = row_norm2(a, row, j + 1 < n2 ? row + K : row, chunk);
        if (j == 0) {  // minDist == -1 branch of :1041
            closest = 0;
            min_norm2 = norm;
            min_dist = sqrtf(norm);
        } else if (norm < min_norm2) {
            const float dist = sqrtf(norm);
            if (dist < min_dist) {  // :1041
                second = closest;   // :1043
                closest = j;        // :1044
                min_dist = dist;    // :1045
                min_norm2 = norm;
            }
        }
    }
    if (n2 > 0) row_chunk_arrived(chunk);  // the request behind the last row: nothing in flight at the end
    if (!active) return;

    float second_dist = -1.0f;
    if (second >= 0) second_dist = sqrtf(ref_norm2(reinterpret_cast<const float *>(a), desc2 + (size_t)second * K));  // :1064
    if (closest_out) closest_out[i] = closest;
    if (second_out) second_out[i] = second;
    if (min_dist_out) min_dist_out[i] = min_dist;
    if (second_dist_out) second_dist_out[i] = second_dist;
    if (status_out)
        status_out[i] = match_status(closest, second, min_dist, second_dist, score1[i], score2, dist_limit);
}


// ---------------------------------------------------------------------------
// Overflow pass of the prefilter: the exact scan, cut along list2.
//
// A query whose candidate lists overflow must be scanned in full.  One lane per query (above) makes
// that one dependent chain of n2 x 384 operations -- 0.4 s for ONE query at n2 = 200 000, next to a
// 7.6 ms call.  Here a wave still holds 64 queries (rows arrive through the scalar data path, three
// VALU operations per element: the scan runs at the VALU rate once the chip is full), but list2 is cut
// into S contiguous segments scanned by S waves, S chosen ON THE DEVICE from the overflow count so
// that the launch's OVF_WAVES waves are all busy: a handful of overflowing queries get thousands of
// short segments, thousands of queries get a few long ones.
//
// A segment's scan is the reference's scan started from "no record yet" (Features.cpp:1035-1047) and
// yields (closest_s, minDist_s, second_s, dist(second_s)).  The merge walks the segments in order.
// The full scan's records inside segment s are exactly the segment's own records that beat the minimum
// G held before it -- a suffix of the segment's strictly decreasing record chain -- so after a segment
// with minDist_s < G:  closest = closest_s, and the previous record holder is second_s if it beat G
// too, else the closest held before the segment.  Row 0 is taken unconditionally (minDist == -1), so a
// NaN there sticks: no later `<` is ever true.  Rows after the first compare against +inf when the
// segment has no record yet: a row at +inf or NaN is never a record there, and never one of the full
// scan either (nothing is < a minimum that row 0 has already set).
// ---------------------------------------------------------------------------
constexpr int OVF_WAVES = 3072;      // waves per launch: 3 per SIMD on 256 CUs -- the kernel holds the query in 128 VGPRs
                                     // (165 in all), so three waves share a SIMD and a fourth would wait for a second round
constexpr int OVF_MAX_SEGS = 2048;   // segments per query group at most

struct OvfPartial {
    int closest;
    float min_dist;
    int second;
    float second_dist;
};

// the split the seg kernel and the merge kernel both derive from the overflow count
__device__ __forceinline__ void ovf_split(int n_queries, int n2, int &groups, int &segs, int &rows_per_seg)
{
    groups = (n_queries + 63) / 64;
    segs = groups > 0 ? OVF_WAVES / groups : 1;
    segs = segs < 1 ? 1 : segs > OVF_MAX_SEGS ? OVF_MAX_SEGS : segs;
    rows_per_seg = (n2 + segs - 1) / segs;
    if (rows_per_seg < 16) rows_per_seg = 16;    // shorter segments only add merge steps
    segs = (n2 + rows_per_seg - 1) / rows_per_seg;
    if (segs < 1) segs = 1;
}

__global__ __launch_bounds__(64) void match_exact_seg_kernel(
    const float *__restrict__ desc1, const float *__restrict__ desc2, int n2,
    const int32_t *__restrict__ query_list, const int *__restrict__ n_queries_dev,
    OvfPartial *__restrict__ partial)
{
    const int n_queries = *n_queries_dev;
    if (n_queries <= 0) return;
    int groups, segs, rows_per_seg;
    ovf_split(n_queries, n2, groups, segs, rows_per_seg);
    const int slots = OVF_WAVES / segs > 0 ? OVF_WAVES / segs : 1;  // query groups in flight
    const int seg = blockIdx.x % segs;
    if ((int)blockIdx.x >= slots * segs) return;
    const int j_begin = seg * rows_per_seg;
    const int j_end = j_begin + rows_per_seg < n2 ? j_begin + rows_per_seg : n2;
#pragma unroll 1
    for (int grp = blockIdx.x / segs; grp < groups; grp += slots) {
        const int slot = grp * 64 + threadIdx.x;
        const bool active = slot < n_queries;
        // inactive lanes shadow the last query so that the row loop stays uniform
        const int i = query_list[active ? slot : n_queries - 1];
        // the query as 64 pairs: differences and squares of two elements at a time are packed fp32
        // operations (v_pk_add_f32, v_pk_mul_f32: each lane's two results rounded exactly as the scalar
        // instructions round them); the sum stays ONE accumulator, k ascending -- 4 vector
        // instructions per two elements instead of 6
        f32x2 a[K / 2];
        const float4 *arow = reinterpret_cast<const float4 *>(desc1 + (size_t)i * K);
#pragma unroll
        for (int k = 0; k < K / 4; ++k) {
            const float4 v = arow[k];
            a[2 * k + 0] = f32x2{v.x, v.y};
            a[2 * k + 1] = f32x2{v.z, v.w};
        }
        int closest = -1, second = -1;
        float min_dist = __builtin_inff(), second_dist = -1.0f;
        float min_norm2 = __builtin_inff();
        f32x16s chunk = row_chunk_request(desc2 + (size_t)j_begin * K);
        for (int j = j_begin; j < j_end; ++j) {
            const float *row = desc2 + (size_t)j * K;
            const float norm = row_norm2(a, row, j + 1 < j_end ? row + K : row, chunk);
            if (j == 0) {  // minDist == -1 branch of :1041: taken whatever the value
                closest = 0;
                min_norm2 = norm;
                min_dist = sqrtf(norm);
            } else if (norm < min_norm2) {  // sqrt is monotonic: only then can dist be smaller
                const float dist = sqrtf(norm);
                if (dist < min_dist) {  // :1041
                    second = closest;   // :1043
                    second_dist = min_dist;
                    closest = j;        // :1044
                    min_dist = dist;    // :1045
                    min_norm2 = norm;
                }
            }
        }
        row_chunk_arrived(chunk);  // the request behind the segment's last row
        if (active) {
            OvfPartial r;
            r.closest = closest;
            r.min_dist = min_dist;
            r.second = second;
            r.second_dist = second_dist;
            partial[(size_t)slot * segs + seg] = r;
        }
    }
}

__global__ __launch_bounds__(256) void match_exact_merge_kernel(
    const float *__restrict__ score1, const float *__restrict__ score2, int n2, float dist_limit,
    const int32_t *__restrict__ query_list, const int *__restrict__ n_queries_dev,
    const OvfPartial *__restrict__ partial, int32_t *__restrict__ closest_out, int32_t *__restrict__ second_out,
    float *__restrict__ min_dist_out, float *__restrict__ second_dist_out, uint8_t *__restrict__ status_out)
{
    const int n_queries = *n_queries_dev;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= n_queries) return;
    int groups, segs, rows_per_seg;
    ovf_split(n_queries, n2, groups, segs, rows_per_seg);
    const int i = query_list[slot];
    const OvfPartial *p = partial + (size_t)slot * segs;
    // segment 0 holds row 0: the scan's state after it is the segment's own
    int closest = p[0].closest, second = p[0].second;
    float min_dist = p[0].min_dist, second_dist = p[0].second_dist;
    for (int s = 1; s < segs; ++s) {
        const OvfPartial r = p[s];
        if (r.closest < 0 || !(r.min_dist < min_dist)) continue;  // no record of the full scan in this segment
        if (r.second >= 0 && r.second_dist < min_dist) {          // its previous record holder was one too
            second = r.second;
            second_dist = r.second_dist;
        } else {
            second = closest;
            second_dist = min_dist;
        }
        closest = r.closest;
        min_dist = r.min_dist;
    }
    if (closest_out) closest_out[i] = closest;
    if (second_out) second_out[i] = second;
    if (min_dist_out) min_dist_out[i] = min_dist;
    if (second_dist_out) second_dist_out[i] = second_dist;
    if (status_out)
        status_out[i] = match_status(closest, second, min_dist, second_dist, score1[i], score2, dist_limit);
}


// ---------------------------------------------------------------------------
// MFMA prefilter + exact replay.
//
// The reference's scan only changes state at strict running minima ("records")
// of dist(j), j ascending.  Phase 1 computes approximate squared distances
//     d~(i,j) = (|a_i|^2 + |b_j|^2) - 2 a_i.b_j
// with fp32 MFMA (v_mfma_f32_32x32x2_f32: an exact-fp32 fma chain) and emits, per
// query, every j that COULD be a record: with eps(i,j) = C*(|a|^2+|b|^2) + tiny a
// bound on |d~ - norm_ref| (norm_ref = the fp32 value the reference holds before
// its sqrt; see DESIGN.md for the derivation of C), j is emitted unless
//     d~ - eps  >  min over the j' < j already seen by this lane of (d~' + eps').
// A true record has norm_ref(j) < norm_ref(j') for every j' < j (sqrt is
// monotonic), hence d~-eps <= norm_ref(j) < norm_ref(j') <= d~'+eps': it is never
// filtered.  NaN/inf anywhere makes the comparison false, i.e. emits.  Phase 2
// sorts each query's candidates by j, recomputes their distances in the
// reference's exact operation order and replays the reference scan over them;
// since the prefix minimum is always attained at a record and every record is a
// candidate, closest / second / minDist come out bit-identical.  A query whose
// candidate list overflows is redone by the exact kernel above.
// ---------------------------------------------------------------------------

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_WAVES = 4;            // waves per workgroup
constexpr int MF_QW = 64;              // queries per wave (two 32-wide MFMA column blocks)
constexpr int MF_QB = MF_WAVES * MF_QW;  // queries per workgroup
constexpr int MF_JB = 32;              // list2 rows per MFMA row block
constexpr int MF_TJ = 2;               // row blocks per LDS tile
constexpr int MF_TILE_ROWS = MF_JB * MF_TJ;
constexpr int MF_BLOCK_FLOATS = MF_JB * K;  // 4096 floats = 16 KiB per row block
constexpr int MF_CAP = 256;            // candidates per query that the replay kernel sorts
constexpr size_t MF_LIST_BYTES = (size_t)2 << 30;  // what the x1 kernel's candidate lists may take
constexpr float MF_C = 6.103515625e-05f;  // 2^-14 >= 523 * 2^-24
constexpr float MF_TINY = 1e-37f;      // covers underflow in either computation


// Epilogue shared by both prefilter kernels.  After an MFMA row block this lane holds, for
// its two queries (column blocks 0/1), the dot products with rows
//   j = j0 + (r&3) + 8*(r>>2) + 4*half, r = 0..15 (ascending in r),
// tests each against the running bound and appends survivors to the query's list.
__device__ __forceinline__ void prefilter_epilogue(const f32x16 &acc0, const f32x16 &acc1, int j0, int half,
                                                   const float *__restrict__ bnorm, int n2, float na0,
                                                   float na1, bool live0, bool live1, int q0i, int q1i,
                                                   float c, float &m0, float &m1,
                                                   int *__restrict__ cand_count, int *__restrict__ cand)
{
    const float4 *nb4 = reinterpret_cast<const float4 *>(bnorm + j0 + 4 * half);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 nb = nb4[2 * q];  // rows j0 + 8q + 4half + 0..3
        const float nbv[4] = {nb.x, nb.y, nb.z, nb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            const int j = j0 + 8 * q + 4 * half + e;
            {
                const float s = na0 + nbv[e];
                const float d = fmaf(-2.0f, acc0[r], s);
                const float eps = fmaf(c, s, MF_TINY);
                if (!(d - eps > m0) && live0 && j < n2) {
                    const int pos = atomicAdd(cand_count + q0i, 1);
                    if (pos < MF_CAP) cand[(size_t)q0i * MF_CAP + pos] = j;
                }
                m0 = fminf(m0, d + eps);
            }
            {
                const float s = na1 + nbv[e];
                const float d = fmaf(-2.0f, acc1[r], s);
                const float eps = fmaf(c, s, MF_TINY);
                if (!(d - eps > m1) && live1 && j < n2) {
                    const int pos = atomicAdd(cand_count + q1i, 1);
                    if (pos < MF_CAP) cand[(size_t)q1i * MF_CAP + pos] = j;
                }
                m1 = fminf(m1, d + eps);
            }
        }
    }
}

// list2 -> MFMA A-operand fragment order, rows padded with zeros to n2pad:
//   frag[((jb*16 + g)*64 + lane)*4 + t] = desc2[jb*32 + (lane&31)][2*(4g+t) + (lane>>5)]
__global__ __launch_bounds__(256) void match_prep_frag_kernel(const float *__restrict__ desc2, int n2,
                                                              int n2pad, float *__restrict__ frag)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // one float4 of frag
    const size_t total = (size_t)n2pad * K / 4;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int g = (int)((idx >> 6) & 15);
    const size_t jb = idx >> 10;
    const size_t j = jb * MF_JB + (lane & 31);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < (size_t)n2) {
        const float *row = desc2 + j * K + 8 * g + (lane >> 5);
        v = make_float4(row[0], row[2], row[4], row[6]);
    }
    reinterpret_cast<float4 *>(frag)[idx] = v;
}

// squared norms (any summation order: covered by the error bound); padding rows get 0
__global__ __launch_bounds__(256) void match_norm_kernel(const float *__restrict__ desc, int n, int npad,
                                                         float *__restrict__ norm2)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    if (r >= npad) return;
    if (r < n) {
        const float4 *row = reinterpret_cast<const float4 *>(desc + (size_t)r * K);
#pragma unroll 8
        for (int k = 0; k < K / 4; ++k) {
            const float4 v = row[k];
            s = fmaf(v.x, v.x, s);
            s = fmaf(v.y, v.y, s);
            s = fmaf(v.z, v.z, s);
            s = fmaf(v.w, v.w, s);
        }
    }
    // keep sums of two norms finite; anything larger (or inf/NaN) becomes NaN, which the
    // prefilter treats as "cannot be bounded": the row / query is always a candidate
    norm2[r] = s <= 1e37f ? s : __builtin_nanf("");
}

__global__ __launch_bounds__(MF_WAVES * 64, 2) void match_mfma_kernel(
    const float *__restrict__ desc1, const float *__restrict__ qnorm, int n1,
    const float *__restrict__ frag, const float *__restrict__ bnorm, int n2, int tiles_total,
    int tiles_per_seg, int *__restrict__ cand_count, int *__restrict__ cand)
{
    __shared__ __attribute__((aligned(16))) float tile[2][MF_TJ * MF_BLOCK_FLOATS];  // 64 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int q0i = blockIdx.x * MF_QB + wave * MF_QW + col;  // column block 0
    const int q1i = q0i + 32;                                   // column block 1

    // query fragments: B operand, lane holds q[i][2s + half], s = 0..63
    float qa[64], qb[64];
    {
        const bool ok0 = q0i < n1, ok1 = q1i < n1;
        const float4 *r0 = reinterpret_cast<const float4 *>(desc1 + (size_t)(ok0 ? q0i : 0) * K);
        const float4 *r1 = reinterpret_cast<const float4 *>(desc1 + (size_t)(ok1 ? q1i : 0) * K);
#pragma unroll
        for (int v = 0; v < 32; ++v) {
            const float4 f0 = r0[v], f1 = r1[v];
            qa[2 * v] = ok0 ? (half ? f0.y : f0.x) : 0.f;
            qa[2 * v + 1] = ok0 ? (half ? f0.w : f0.z) : 0.f;
            qb[2 * v] = ok1 ? (half ? f1.y : f1.x) : 0.f;
            qb[2 * v + 1] = ok1 ? (half ? f1.w : f1.z) : 0.f;
        }
    }
    const float na0 = q0i < n1 ? qnorm[q0i] : 0.f;
    const float na1 = q1i < n1 ? qnorm[q1i] : 0.f;
    const bool live0 = q0i < n1, live1 = q1i < n1;
    float m0 = __builtin_inff(), m1 = __builtin_inff();  // running min of (d~ + eps)

    const int t_begin = blockIdx.y * tiles_per_seg;
    const int t_end = min(t_begin + tiles_per_seg, tiles_total);
    if (t_begin >= t_end) return;

    // Tile staging: 32 KiB = 32 wave-instructions of global_load_lds_dwordx4 (1 KiB each,
    // LDS destination = wave-uniform base + lane*16, no VGPRs).  The loads of tile t+1 are
    // issued before the MFMAs of tile t and retire at the barrier that ends the iteration.
    auto stage = [&](int t, int buf) {
        const float *src = frag + (size_t)t * MF_TJ * MF_BLOCK_FLOATS;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int chunk = (k * MF_WAVES + wave) * 256;  // floats; 64 lanes x 4 floats per chunk
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + chunk + 4 * lane),
                (__attribute__((address_space(3))) void *)(&tile[buf][chunk]), 16, 0, 0);
        }
    };

    stage(t_begin, 0);
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) stage(t + 1, buf ^ 1);

        for (int jb = 0; jb < MF_TJ; ++jb) {
            const float4 *bf = reinterpret_cast<const float4 *>(tile[buf] + jb * MF_BLOCK_FLOATS) + lane;
            f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 b = bf[g * 64];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, qa[4 * g + 0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, qb[4 * g + 0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, qa[4 * g + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, qb[4 * g + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, qa[4 * g + 2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, qb[4 * g + 2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, qa[4 * g + 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, qb[4 * g + 3], acc1, 0, 0, 0);
            }
            prefilter_epilogue(acc0, acc1, (t * MF_TJ + jb) * MF_JB, half, bnorm, n2, na0, na1, live0, live1,
                               q0i, q1i, MF_C, m0, m1, cand_count, cand);
            // the two half-waves see disjoint rows of the same queries: share the bound
            m0 = fminf(m0, __shfl_xor(m0, 32));
            m1 = fminf(m1, __shfl_xor(m1, 32));
        }

        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// bf16 x 3 variant of the prefilter.  Every fp32 value is split a = hi + lo + r with
// hi = bf16(a), lo = bf16(a - hi), |r| <= 2^-18 |a|, and the dot product is taken as
// hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate, 3 products =
// 5.3x fewer matrix cycles).  The extra error (dropped lo.lo and residual terms
// <= 3*2^-18 sum|a||b|, 384 fp32 accumulations) is covered by the wider bound MF_C_BF16
// (DESIGN.md); everything downstream -- candidate lists, exact replay -- is unchanged, so the
// outputs stay bit-identical to the reference.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr float MF_C_BF16 = 1.220703125e-04f;  // 2^-13

__device__ __forceinline__ unsigned bf16_rne(float a)
{
    unsigned u = __float_as_uint(a);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;  // NaN stays NaN
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bf16_split(float a, unsigned &hi, unsigned &lo)
{
    hi = bf16_rne(a);
    lo = bf16_rne(a - __uint_as_float(hi << 16));
}
// 8 consecutive floats -> packed hi (8 x bf16) and lo (8 x bf16)
__device__ __forceinline__ void bf16_split8(const float4 &x, const float4 &y, uint4 &hi, uint4 &lo)
{
    const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
    unsigned h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bf16_split(v[i], h[i], l[i]);
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}


// ---- pieces of the pipelined bf16 prefilter --------------------------------
struct BoundState {
    float naL, naU;  // |a|^2 (1 -+ c) -+ tiny for this lane's query
    float m;         // running min of the upper bounds U seen so far
    unsigned mask;   // bit (15 - r) set: row r of the current block may be a record
};

// mask = 2*mask + !(L > m)   (v_cmp_ngt: true for NaN), then m = min(m, U)
__device__ __forceinline__ void bound_test(BoundState &b, float L, float U)
{
    asm("v_cmp_ngt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
        : "+v"(b.mask) : "v"(L), "v"(b.m) : "vcc");
    asm("v_min_f32 %0, %0, %1" : "+v"(b.m) : "v"(U));
}

// rows of this lane in a 32-row block, in accumulator-register order r = 0..15:
//   j = j0 + (r&3) + 8*(r>>2) + 4*half
__device__ __forceinline__ void load_bnorm(const float *__restrict__ bnorm, int j0, int half, float (&nbv)[16])
{
    const float4 *nb4 = reinterpret_cast<const float4 *>(bnorm + j0 + 4 * half);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = nb4[2 * q];
        nbv[4 * q] = v.x; nbv[4 * q + 1] = v.y; nbv[4 * q + 2] = v.z; nbv[4 * q + 3] = v.w;
    }
}

// epilogue slice for k-step s: accumulator registers 2s and 2s+1 of both column blocks
__device__ __forceinline__ void bound_step(const f32x16 &p0, const f32x16 &p1, int s, const float (&nbv)[16],
                                           float cL, float cU, BoundState &b0, BoundState &b1)
{
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * s + e;
        const float sl0 = fmaf(nbv[r], cL, b0.naL), su0 = fmaf(nbv[r], cU, b0.naU);
        const float sl1 = fmaf(nbv[r], cL, b1.naL), su1 = fmaf(nbv[r], cU, b1.naU);
        bound_test(b0, fmaf(-2.0f, p0[r], sl0), fmaf(-2.0f, p0[r], su0));
        bound_test(b1, fmaf(-2.0f, p1[r], sl1), fmaf(-2.0f, p1[r], su1));
    }
}

// one 32-row block: 48 MFMAs into (n0, n1); if EPI, the epilogue of the previous block
// (accumulators p0, p1) is sliced between the k-steps
template <bool EPI>
__device__ __forceinline__ void mfma_block(const uint4 *bf, const bf16x8 (&qh0)[8], const bf16x8 (&ql0)[8],
                                           const bf16x8 (&qh1)[8], const bf16x8 (&ql1)[8], f32x16 &n0,
                                           f32x16 &n1, const f32x16 &p0, const f32x16 &p1,
                                           const float (&nbv)[16], float cL, float cU, BoundState &b0,
                                           BoundState &b1)
{
    f32x16 a0 = {0}, a1 = {0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bf[(2 * s) * 64]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bf[(2 * s + 1) * 64]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, qh0[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, qh1[s], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ql0[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ql1[s], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, qh0[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, qh1[s], a1, 0, 0, 0);
        if (EPI) bound_step(p0, p1, s, nbv, cL, cU, b0, b1);
    }
    n0 = a0;
    n1 = a1;
}

// append the flagged rows of one block to query qi's candidate list (few lanes, rare)
__device__ __forceinline__ void emit_flagged(unsigned mask, int qi, int j0, int half, int n2,
                                             int *__restrict__ cand_count, int *__restrict__ cand)
{
    while (mask) {
        const int bit = 31 - __clz((int)mask);
        mask &= ~(1u << bit);
        const int r = 15 - bit;
        const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (j < n2) {
            const int pos = atomicAdd(cand_count + qi, 1);
            if (pos < MF_CAP) cand[(size_t)qi * MF_CAP + pos] = j;
        }
    }
}

// list2 -> A-operand fragment order of v_mfma_f32_32x32x16_bf16 (lane holds
// A[row = lane&31][k = 16s + 8*(lane>>5) + 0..7]):
//   frag16[((jb*8 + s)*2 + part)*64 + lane]  (uint4 = 8 bf16), part 0 = hi, 1 = lo
__global__ __launch_bounds__(256) void match_prep_frag_bf16_kernel(const float *__restrict__ desc2, int n2,
                                                                   int n2pad, uint4 *__restrict__ frag16)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // one (jb, s, lane)
    const size_t total = (size_t)n2pad / MF_JB * 8 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int s = (int)((idx >> 6) & 7);
    const size_t jb = idx >> 9;
    const size_t j = jb * MF_JB + (lane & 31);
    uint4 hi = make_uint4(0, 0, 0, 0), lo = hi;
    if (j < (size_t)n2) {
        const float4 *row = reinterpret_cast<const float4 *>(desc2 + j * K + 16 * s + 8 * (lane >> 5));
        bf16_split8(row[0], row[1], hi, lo);
    }
    uint4 *dst = frag16 + ((jb * 8 + s) * 2) * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
}

__global__ __launch_bounds__(MF_WAVES * 64, 2) void match_mfma_bf16_kernel(
    const float *__restrict__ desc1, const float *__restrict__ qnorm, int n1,
    const uint4 *__restrict__ frag16, const float *__restrict__ bnorm, int n2, int tiles_total,
    int tiles_per_seg, int *__restrict__ cand_count, int *__restrict__ cand)
{
    // one row block = 8 k-steps x {hi, lo} x 64 lanes x 16 B = 16 KiB; tile = MF_TJ row blocks
    __shared__ uint4 tile[2][MF_TJ * 1024];  // 64 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int q0i = blockIdx.x * MF_QB + wave * MF_QW + col;
    const int q1i = q0i + 32;
    const bool live0 = q0i < n1, live1 = q1i < n1;

    // query fragments: B operand, lane holds q[i][16s + 8*half + 0..7] as hi and lo
    bf16x8 qh0[8], ql0[8], qh1[8], ql1[8];
    {
        const float4 *r0 = reinterpret_cast<const float4 *>(desc1 + (size_t)(live0 ? q0i : 0) * K + 8 * half);
        const float4 *r1 = reinterpret_cast<const float4 *>(desc1 + (size_t)(live1 ? q1i : 0) * K + 8 * half);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint4 h, l;
            bf16_split8(r0[4 * s], r0[4 * s + 1], h, l);
            if (!live0) h = l = make_uint4(0, 0, 0, 0);
            qh0[s] = __builtin_bit_cast(bf16x8, h);
            ql0[s] = __builtin_bit_cast(bf16x8, l);
            bf16_split8(r1[4 * s], r1[4 * s + 1], h, l);
            if (!live1) h = l = make_uint4(0, 0, 0, 0);
            qh1[s] = __builtin_bit_cast(bf16x8, h);
            ql1[s] = __builtin_bit_cast(bf16x8, l);
        }
    }
    // Bound in product form: with s = |a|^2 + |b|^2, L = s(1-c) - tiny - 2 a.b and
    // U = s(1+c) + tiny - 2 a.b, so the per-element work is two fma to form s(1-+c), two
    // fma for L and U, one compare and one min.  Norms above 1e37 or non-finite were
    // replaced by NaN (match_norm_kernel), which makes L NaN and therefore emits.
    const float cL = 1.0f - MF_C_BF16, cU = 1.0f + MF_C_BF16;
    const float na0 = live0 ? qnorm[q0i] : 0.f;
    const float na1 = live1 ? qnorm[q1i] : 0.f;
    BoundState b0, b1;
    b0.naL = na0 * cL - MF_TINY; b0.naU = na0 * cU + MF_TINY; b0.m = __builtin_inff(); b0.mask = 0;
    b1.naL = na1 * cL - MF_TINY; b1.naU = na1 * cU + MF_TINY; b1.m = __builtin_inff(); b1.mask = 0;

    const int t_begin = blockIdx.y * tiles_per_seg;
    const int t_end = min(t_begin + tiles_per_seg, tiles_total);
    if (t_begin >= t_end) return;

    auto stage = [&](int t, int buf) {
        const uint4 *src = frag16 + (size_t)t * MF_TJ * 1024;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int chunk = (k * MF_WAVES + wave) * 64;  // uint4 units, 1 KiB per wave-instruction
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + chunk + lane),
                (__attribute__((address_space(3))) void *)(&tile[buf][chunk]), 16, 0, 0);
        }
    };
    // close a row block whose epilogue has run: share the bound between the two half-waves
    // (they saw disjoint rows of the same queries) and append the flagged rows
    auto finish = [&](int j0) {
        b0.m = fminf(b0.m, __shfl_xor(b0.m, 32));
        b1.m = fminf(b1.m, __shfl_xor(b1.m, 32));
        const unsigned k0 = live0 ? b0.mask : 0u, k1 = live1 ? b1.mask : 0u;
        if (__any((k0 | k1) != 0u)) {
            emit_flagged(k0, q0i, j0, half, n2, cand_count, cand);
            emit_flagged(k1, q1i, j0, half, n2, cand_count, cand);
        }
        b0.mask = 0;
        b1.mask = 0;
    };

    stage(t_begin, 0);
    __syncthreads();

    // Software pipeline over row blocks: while the 48 MFMAs of block k are issued into one
    // accumulator pair, the epilogue of block k-1 runs on the other pair, four elements per
    // k-step, so that the VALU work sits in the gaps between MFMA issues of the same wave.
    f32x16 x0, x1, y0, y1;
    float nbv[16];
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) stage(t + 1, buf ^ 1);
        const uint4 *bf = tile[buf] + lane;
        const int j0 = t * MF_TILE_ROWS;
        if (t == t_begin) {
            mfma_block<false>(bf, qh0, ql0, qh1, ql1, x0, x1, y0, y1, nbv, cL, cU, b0, b1);
        } else {
            load_bnorm(bnorm, j0 - MF_JB, half, nbv);
            mfma_block<true>(bf, qh0, ql0, qh1, ql1, x0, x1, y0, y1, nbv, cL, cU, b0, b1);
            finish(j0 - MF_JB);
        }
        load_bnorm(bnorm, j0, half, nbv);
        mfma_block<true>(bf + 1024, qh0, ql0, qh1, ql1, y0, y1, x0, x1, nbv, cL, cU, b0, b1);
        finish(j0);
        __syncthreads();
    }
    {   // drain: epilogue of the last block
        const int j0 = (t_end - 1) * MF_TILE_ROWS + MF_JB;
        load_bnorm(bnorm, j0, half, nbv);
#pragma unroll
        for (int s = 0; s < 8; ++s) bound_step(y0, y1, s, nbv, cL, cU, b0, b1);
        finish(j0);
    }
}

// ---------------------------------------------------------------------------
// bf16 x 1 variant: only hi.hi.  Dropping both lo terms costs a relative error of
// (2^-8 + 2^-18) on sum|a_k b_k| <= S/2, i.e. |d~ - D| <= 3.92e-3 S once doubled, plus the
// fp32 accumulation and norm terms (< 3e-5 S together): MF_C_X1 = 4.5e-3 covers it with 14 %
// to spare.  The looser bound only lengthens the candidate lists (26 instead of 13 per query
// on the 200k x 200k benchmark set) while the matrix work drops to a third; the exact replay
// is unchanged, so the outputs stay bit-identical.
//   tile (128 rows, MF_TILE_U4 uint4): 4 blocks of [s*64 + lane] uint4 = 8 bf16 (8 KiB per
//   32-row block), then the 128 seeds -h of the tile as floats (padded to 1 KiB)
//
// The whole bound lives in the accumulator.  With h_j = |b_j|^2 / 2 the MFMA chain of a block
// starts from C = -h_j instead of 0, so it ends in y_j = a.b_j - h_j = -(|b_j|^2 - 2 a.b_j)/2:
// larger y = closer row, and |a|^2 is common to all rows of a query.  With
// e = C (|a|^2 + max_j |b_j|^2) + tiny, row j cannot be a record if an earlier row j' has
// d~_j' + e_j' < d~_j - e_j, which holds whenever y_j < Y - e, Y = max of y over (a subset
// of) the earlier rows.  Records are rare (tens per query in 200 000 rows), so the test runs
// per BLOCK: one v_max3_f32 per two rows folds the block's 16 y values of a lane into M, and
// only if some lane of the wave has !(M < Y - e) does the wave run the per-row tests of that
// block (same accumulators, still live).  Half a VALU instruction per (row, query) element
// instead of four: the kernel is bound by the matrix pipe.
//
// v_max3_f32 drops NaNs, so nothing that must be emitted may hide behind one: a row whose
// norm is not finite (or above 1e37) gets h = -inf and a zeroed fragment, hence y = +inf --
// always a candidate, and excluded from Y in the per-row pass; a query with a non-finite norm
// has e = NaN, every comparison fails and every block takes the per-row pass, which emits
// all rows (the query overflows to the exact kernel).
// ---------------------------------------------------------------------------
constexpr float MF_C_X1 = 4.5e-3f;
constexpr int MF_TJ1 = 4;
constexpr int MF_TILE_ROWS1 = MF_JB * MF_TJ1;
constexpr int MF_TILE_U4 = MF_TJ1 * 512 + 64;  // uint4 units per staged tile: fragments + seeds
constexpr int X1_WAVES = 8;                    // waves per workgroup: one workgroup per CU, 2 waves per SIMD
constexpr int X1_QB = X1_WAVES * MF_QW;        // queries per workgroup

// list2's whole preparation for the x1 prefilter in ONE pass over the descriptors (round 2 used two
// kernels that each read all of list2, the first with one lane per row, i.e. 64 rows and 64 cache
// lines per load instruction: 107 + 24 us at 200 000 rows).  A workgroup takes one 32-row block; eight
// lanes share a row, 16 consecutive floats each, so a wave reads 4 KiB of contiguous memory.  Per row:
// the squared norm (any summation order: covered by the error bound; above 1e37 or non-finite -> NaN =
// "cannot be bounded"), its seed -|b|^2 / 2 (+inf for an unbounded row) behind the fragments of the
// row's tile, the bf16 fragments themselves (zeros for an unbounded or padding row), and the largest
// finite norm of the block (match_max_kernel folds the blocks' maxima into the list's).
__global__ __launch_bounds__(256) void match_prep_x1_kernel(const float *__restrict__ desc2, int n2, int n2pad,
                                                            float *__restrict__ bnorm, float *__restrict__ block_max,
                                                            uint4 *__restrict__ frag16)
{
    const int t = threadIdx.x;
    const size_t jb = blockIdx.x;
    const int rl = t >> 3, c = t & 7;           // row of the block, 16-float chunk of the row
    const size_t j = jb * MF_JB + rl;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < (size_t)n2) {
        const float4 *row = reinterpret_cast<const float4 *>(desc2 + j * K + 16 * c);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = row[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s = fmaf(v[i].x, v[i].x, s);
        s = fmaf(v[i].y, v[i].y, s);
        s = fmaf(v[i].z, v[i].z, s);
        s = fmaf(v[i].w, v[i].w, s);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);                      // all eight lanes of the row hold its squared norm
    const float nv = s <= 1e37f ? s : __builtin_nanf("");
    float m = nv == nv ? nv : 0.f;              // largest finite norm: NaN rows enter no bound
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    // (no atomic on one word from 25 000 waves, not even a look at it first: the same-address traffic
    // alone took 95 of this kernel's 119 us; the per-block maxima are folded by match_max_kernel)
    __shared__ float wmax[4];
    if ((t & 63) == 0) wmax[t >> 6] = m;
    __syncthreads();
    if (t == 0) block_max[jb] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    uint4 *tile = frag16 + (jb / MF_TJ1) * MF_TILE_U4;
    if (c == 0) {
        bnorm[j] = nv;                          // (j < n2pad: the grid is n2pad / 32 blocks)
        reinterpret_cast<float *>(tile + MF_TJ1 * 512)[(jb % MF_TJ1) * MF_JB + rl] = nv == nv ? -0.5f * nv : __builtin_inff();
    }
    uint4 h0 = make_uint4(0, 0, 0, 0), h1 = h0, lo;
    if (nv == nv) {                             // unbounded rows enter the matrix as zeros
        bf16_split8(v[0], v[1], h0, lo);
        bf16_split8(v[2], v[3], h1, lo);
    }
    // fragment order: frag16[tile][block][k-step s = c][lane = row + 32 * (upper 8 of the 16 floats)]
    uint4 *dst = tile + (jb % MF_TJ1) * 512 + c * 64 + rl;
    dst[0] = h0;
    dst[32] = h1;
}

// largest finite squared norm of list2 from the per-block maxima (one workgroup; 0 if there is none)
__global__ __launch_bounds__(1024) void match_max_kernel(const float *__restrict__ block_max, int n, float *__restrict__ out)
{
    __shared__ float part[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, block_max[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, part[i]);
        *out = m;
    }
}

// The first MF_WARM_ROWS rows of list2 are handled apart, per query, in fp32 (match_warm_kernel):
//   * they are scanned in order with the fp32 prefilter's bound (C = 2^-14): row j is a candidate
//     unless an earlier row is clearly closer -- a handful of candidates instead of the whole
//     first block a cold bound would emit (the x1 kernel never emits rows below MF_WARM_ROWS);
//   * yw[q] = max over those rows of a.b_j - |b_j|^2 / 2 is a valid bound for every later row,
//     so every piece of a query block, also one that starts in the middle of list2, starts warm.
// Rows with a non-finite norm are always candidates and never part of the bound; a query that
// yields nothing finite keeps -inf (= cold).  The candidates go to the query's last sub-list.
constexpr int MF_WARM_ROWS = 32;
__global__ __launch_bounds__(256) void match_warm_kernel(const float *__restrict__ desc1, float *__restrict__ qnorm,
                                                         int n1, const float *__restrict__ desc2, int n2,
                                                         const float *__restrict__ bnorm, const float *__restrict__ bmax,
                                                         float *__restrict__ yw, int nsub, int subcap,
                                                         int *__restrict__ cand_count, int *__restrict__ cand,
                                                         int *__restrict__ overflow_count)
{
    __shared__ float rows[MF_WARM_ROWS][K];
    __shared__ float hn[MF_WARM_ROWS];
    const int nr = n2 < MF_WARM_ROWS ? n2 : MF_WARM_ROWS;
    for (int i = threadIdx.x; i < nr * K; i += 256) rows[i / K][i % K] = desc2[i];
    if ((int)threadIdx.x < nr) hn[threadIdx.x] = 0.5f * bnorm[threadIdx.x];
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n1) return;
    float a[K];
    const float4 *src = reinterpret_cast<const float4 *>(desc1 + (size_t)q * K);
#pragma unroll
    for (int k = 0; k < K / 4; ++k) {
        const float4 v = src[k];
        a[4 * k] = v.x; a[4 * k + 1] = v.y; a[4 * k + 2] = v.z; a[4 * k + 3] = v.w;
    }
    // This kernel is the first to touch the query, so it also does the per-query set-up that used to be
    // launches of their own: the squared norm (as match_norm_kernel sums it) for itself and the x1
    // kernel, and the zeroing of the query's sub-list counts (the x1 kernel stores only the ones it uses).
    float qn = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) qn = fmaf(a[k], a[k], qn);
    qn = qn <= 1e37f ? qn : __builtin_nanf("");
    qnorm[q] = qn;
    for (int s = 0; s + 1 < nsub; ++s) cand_count[(size_t)q * nsub + s] = 0;
    if (q == 0) *overflow_count = 0;
    // on y = -(d~ - |a|^2) / 2 the test d~' + e < d~ - e reads y < y' - e
    const float e = MF_C * (qn + *bmax) + MF_TINY;
    int *list = cand + ((size_t)q * nsub + (nsub - 1)) * subcap;
    int cnt = 0;
    float best = -__builtin_inff();
    for (int j = 0; j < nr; ++j) {
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) dot = fmaf(a[k], rows[j][k], dot);
        const float y = dot - hn[j];
        if (!(y < best - e)) {  // NaN (non-finite norms) is always a candidate
            if (cnt < subcap) list[cnt] = j;
            ++cnt;
        }
        if (y > best && y < __builtin_inff()) best = y;  // NaN and +inf never enter the bound
    }
    cand_count[(size_t)q * nsub + (nsub - 1)] = cnt;
    yw[q] = best;
}

// The folds below are inline asm, which the compiler's hazard recognizer does not cover: an ordinary
// VALU read of an MFMA result gets the wait states the ISA requires (the hardware does not interlock),
// an asm operand does not.  Where a fold follows the MFMAs that write its operands directly, this
// pads by hand (24 wait states, more than any 8-pass MFMA needs); taking both accumulators in and out
// orders it behind their last MFMAs and ahead of every later read.
__device__ __forceinline__ void mfma_results_settle(f32x16 &a, f32x16 &b)
{
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b));
}

struct BoundX1 {
    float Y;        // running max of y over the rows seen so far
    float T;        // Y - e: rows with y < T cannot be records
    float e;        // C (|a|^2 + max |b|^2) + tiny for this lane's query
    float M;        // max of y over the block being closed
    unsigned mask;  // bit (15 - r) set: row r of the current block may be a record
    int cnt;        // rows appended to this lane's private candidate list so far
};

// Candidate lists of the x1 kernel.  A lane owns one (query, segment of list2, half of the
// 32-row blocks) and appends to a private sub-list of that query: nsub = 2 * segments
// sub-lists of subcap slots each.  No atomics, so nothing in the MFMA loop ever waits for a
// memory round trip; the count is kept in a register and stored once at the end.
__device__ __forceinline__ void x1_emit(unsigned mask, int &cnt, int *__restrict__ list, int subcap, int j0,
                                        int half, int n2)
{
    while (mask) {
        const int bit = 31 - __clz((int)mask);
        mask &= ~(1u << bit);
        const int r = 15 - bit;
        const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (j < n2 && j >= MF_WARM_ROWS) {  // the first rows are match_warm_kernel's
            if (cnt < subcap) list[cnt] = j;
            ++cnt;
        }
    }
}

__device__ __forceinline__ void x1_test(BoundX1 &b, float y)
{
    // mask = 2*mask + !(y < T)   (v_cmp_nlt: true for NaN)
    asm("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
        : "+v"(b.mask) : "v"(y), "v"(b.T) : "vcc");
}

// per-row pass over one block (rare): every row is tested against the bound the block
// started with (rows that a record earlier in the same block would have excluded are kept:
// a slightly larger superset for 16 fewer dependent instructions).  y = +inf marks an
// unbounded row: emitted, never part of Y (SPECIAL: some lane of the wave has such a row in
// this block; b.M is then rebuilt from the finite values).
template <bool SPECIAL>
__device__ __forceinline__ void x1_rows(const f32x16 &p, BoundX1 &b)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) x1_test(b, p[r]);
    if (SPECIAL) {
        float m = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float f = p[r] < __builtin_inff() ? p[r] : -__builtin_inff();
            asm("v_max_f32 %0, %0, %1" : "+v"(m) : "v"(f));
        }
        b.M = m;
    }
}

// seeds of this lane's rows of block kb of a staged tile, in accumulator-register order
// r = 0..15:  j = j0 + (r&3) + 8*(r>>2) + 4*half  (4 broadcast ds_read_b128)
__device__ __forceinline__ f32x16 load_seeds(const uint4 *tile, int kb, int half)
{
    const float4 *p = reinterpret_cast<const float4 *>(tile + MF_TJ1 * 512) + kb * 8 + half;
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = p[2 * q];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    return v;
}

// the 8 A fragments (k-steps) of one 32-row block: 8 conflict-free ds_read_b128
__device__ __forceinline__ void load_frag(const uint4 *bf, uint4 (&f)[8])
{
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = bf[s * 64];
}

// Y takes the block maximum, the two half-waves (which hold the two halves of each query's
// rows) exchange it, and the threshold follows
__device__ __forceinline__ void x1_update(BoundX1 &b)
{
    float y;
    asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(b.Y), "v"(b.M));
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(b.Y) : "v"(__uint_as_float(sw[0])), "v"(__uint_as_float(sw[1])));
    b.T = b.Y - b.e;
    b.M = -__builtin_inff();
}

// Close a block whose 16 y values per chain (p0, p1) have been folded into b0.M, b1.M: if some
// lane of the wave may hold a record, run the per-row pass and append the flagged rows; then
// bring the bound up to date.  (Refreshing the bound only once per tile was tried: the staler
// bound costs more in extra candidates than the dozen instructions per block it saves.)
__device__ __forceinline__ void x1_close(const f32x16 &p0, const f32x16 &p1, BoundX1 &b0, BoundX1 &b1,
                                         bool live0, bool live1, int j0, int half, int n2,
                                         int *__restrict__ list0, int *__restrict__ list1, int subcap)
{
    const bool hit = !(b0.M < b0.T) || !(b1.M < b1.T);
    if (__any(hit)) {
        if (__any(b0.M == __builtin_inff() || b1.M == __builtin_inff())) {
            x1_rows<true>(p0, b0);
            x1_rows<true>(p1, b1);
        } else {
            x1_rows<false>(p0, b0);
            x1_rows<false>(p1, b1);
        }
        const unsigned k0 = live0 ? b0.mask : 0u, k1 = live1 ? b1.mask : 0u;
        if (__any((k0 | k1) != 0u)) {
            x1_emit(k0, b0.cnt, list0, subcap, j0, half, n2);
            x1_emit(k1, b1.cnt, list1, subcap, j0, half, n2);
        }
        b0.mask = b1.mask = 0;
    }
    x1_update(b0);
    x1_update(b1);
}

// One 32-row block: 16 MFMAs into (x0, x1), both chains starting from c = the block's seeds
// -h.  Unless FIRST, (p0, p1) hold the previous block (rows from jprev): it is folded and closed
// between the first MFMAs, off the matrix pipe's critical path.  As soon as the two first
// MFMAs have read c, the seeds of the NEXT block (block seeds_kb of the staged tile at seeds)
// are read into it.
template <bool FIRST>
__device__ __forceinline__ void x1_block(const uint4 (&f)[8], const bf16x8 (&qh0)[8], const bf16x8 (&qh1)[8],
                                         f32x16 &x0, f32x16 &x1, const f32x16 &p0, const f32x16 &p1, f32x16 &c,
                                         const uint4 *seeds, int seeds_kb, BoundX1 &b0, BoundX1 &b1,
                                         bool live0, bool live1, int jprev, int half, int n2,
                                         int *__restrict__ list0, int *__restrict__ list1, int subcap)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 fs = __builtin_bit_cast(bf16x8, f[s]);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, qh0[s], s == 0 ? c : x0, 0, 0, 0);
        if (!FIRST) {
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(b0.M) : "v"(p0[4 * s]), "v"(p0[4 * s + 1]));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(b0.M) : "v"(p0[4 * s + 2]), "v"(p0[4 * s + 3]));
        }
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, qh1[s], s == 0 ? c : x1, 0, 0, 0);
        if (!FIRST) {
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(b1.M) : "v"(p1[4 * s]), "v"(p1[4 * s + 1]));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(b1.M) : "v"(p1[4 * s + 2]), "v"(p1[4 * s + 3]));
        }
        if (s == 0) c = load_seeds(seeds, seeds_kb, half);
    }
    if (!FIRST) x1_close(p0, p1, b0, b1, live0, live1, jprev, half, n2, list0, list1, subcap);
#pragma unroll
    for (int s = 4; s < 8; ++s) {
        const bf16x8 fs = __builtin_bit_cast(bf16x8, f[s]);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, qh0[s], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, qh1[s], x1, 0, 0, 0);
    }
}

// Work distribution.  The (query block, tile of list2) plane is flattened query block major
// and cut into one contiguous range of `per` tiles per workgroup, so every workgroup does the
// same amount of matrix work whatever n1 and n2 are, and a query block is cut into few
// pieces (two when there are more query blocks than workgroups).  Every piece starts with a
// cold bound -- its first rows are always candidates -- so few pieces also mean few
// candidates.  Piece p of a query block owns sub-lists 2p and 2p + 1 (one per half-wave).
__global__ __launch_bounds__(X1_WAVES * 64, 2) void match_mfma_bf16x1_kernel(
    const float *__restrict__ desc1, const float *__restrict__ qnorm, int n1,
    const uint4 *__restrict__ frag16, const float *__restrict__ bmax, int n2,
    int tiles_total, long long per, long long flat_total, int grid_segs, int nsub, int subcap,
    int *__restrict__ cand_count, int *__restrict__ cand, const float *__restrict__ ywarm)
{
    // Two separate LDS objects, not one [2][..] array: the tile staging is an LDS DMA
    // (global_load_lds) tracked by vmcnt, and the compiler only lets a ds_read pass an
    // outstanding DMA when it can prove that the read's object is not the DMA's target.  With
    // one array every fragment read of the other buffer waited for the staging just issued
    // (s_waitcnt vmcnt(0) right behind stage(tt + 2)), i.e. the double buffering hid nothing.
    __shared__ uint4 tile0[MF_TILE_U4];  // 33 KiB
    __shared__ uint4 tile1[MF_TILE_U4];  // 33 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    const float nbmax = *bmax;

    // The per-lane part of every staging address is rebuilt from the lane number on each call
    // (the empty asm hides that it is loop invariant): held across the loop, these addresses
    // were spilled, and a scratch reload is a vector memory load that completes in order BEHIND
    // the staging loads -- its s_waitcnt vmcnt(0) waited for the tile just requested.
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto stage = [&](int t, uint4 *dst) __attribute__((always_inline)) {
        const uint4 *src = frag16 + (size_t)t * MF_TILE_U4;
        int l;  // lane number, from the hardware: nothing to keep live (or spill) across the loop
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
#pragma unroll
        for (int k = 0; k < 32 / X1_WAVES; ++k) {
            const int chunk = (k * X1_WAVES + wave_s) * 64;  // uint4 units, 1 KiB per wave-instruction
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + chunk + l),
                (__attribute__((address_space(3))) void *)(&dst[chunk]), 16, 0, 0);
        }
        if (wave_s == 0)  // the tile's seeds
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + MF_TJ1 * 512 + l),
                (__attribute__((address_space(3))) void *)(&dst[MF_TJ1 * 512]), 16, 0, 0);
    };
    // A wave's own staging loads have landed (what the barrier behind it then extends to the
    // whole workgroup).  Explicit: correctness must not rest on waits the compiler happens to add.
    // (the builtin, not inline asm: the compiler's own wait-count bookkeeping sees it and stops
    // treating the staged tile as pending; 0x0F70 = vmcnt(0), expcnt and lgkmcnt untouched)
    auto staged = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0x0F70); };

    long long flat = (long long)blockIdx.x * per;
    long long flat_end = flat + per < flat_total ? flat + per : flat_total;
    int grid_piece = -1;
    if (grid_segs > 0) {
        // Few query blocks (list1 sharded over GPUs): a (segment of list2, query block) grid instead of
        // the flat cut.  Workgroups b, b + 8, b + 16, ... share an XCD and its L2; numbered XCD by XCD,
        // neighbours take the SAME segment for consecutive query blocks, so an XCD's 32 workgroups
        // stream one stretch of list2 in step and all but the first reader of a tile hit in L2 -- where
        // the flat cut has them at 32 different positions of list2 (14 % slower per tile, measured).
        const int per_xcd = (int)gridDim.x / 8;
        const int logical = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
        const int qblocks = (int)(flat_total / tiles_total);
        if (logical >= qblocks * grid_segs) return;
        grid_piece = logical / qblocks;
        const int qb = logical - grid_piece * qblocks;
        flat = (long long)qb * tiles_total + (long long)grid_piece * per;
        flat_end = flat + per;
        const long long row_end = (long long)(qb + 1) * tiles_total;
        flat_end = flat_end < row_end ? flat_end : row_end;
    }
#pragma unroll 1
    while (flat < flat_end) {
        const int qb = (int)(flat / tiles_total);
        const int t_begin = (int)(flat - (long long)qb * tiles_total);
        const int t_end = (int)((long long)t_begin + (flat_end - flat) < (long long)tiles_total
                                    ? (long long)t_begin + (flat_end - flat) : (long long)tiles_total);
        const int ntiles = t_end - t_begin;
        flat += ntiles;
        // pieces of a query block are numbered from the first workgroup that touches it
        const int piece = grid_piece >= 0 ? grid_piece : (int)blockIdx.x - (int)(((long long)qb * tiles_total) / per);

        const int q0i = qb * X1_QB + wave * MF_QW + col;
        const int q1i = q0i + 32;
        const bool live0 = q0i < n1, live1 = q1i < n1;
        bf16x8 qh0[8], qh1[8];
        {
            const float4 *r0 = reinterpret_cast<const float4 *>(desc1 + (size_t)(live0 ? q0i : 0) * K + 8 * half);
            const float4 *r1 = reinterpret_cast<const float4 *>(desc1 + (size_t)(live1 ? q1i : 0) * K + 8 * half);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                uint4 h, l;
                bf16_split8(r0[4 * s], r0[4 * s + 1], h, l);
                if (!live0) h = make_uint4(0, 0, 0, 0);
                qh0[s] = __builtin_bit_cast(bf16x8, h);
                bf16_split8(r1[4 * s], r1[4 * s + 1], h, l);
                if (!live1) h = make_uint4(0, 0, 0, 0);
                qh1[s] = __builtin_bit_cast(bf16x8, h);
            }
        }
        // e = C (|a|^2 + max |b|^2) + tiny in the units of d~ = |a|^2 - 2y; on y the test
        // d~_j' + e < d~_j - e reads y_j < y_j' - e.  A non-finite query norm (NaN) makes T NaN:
        // every row is emitted and the query goes to the exact kernel.
        const float na0 = live0 ? qnorm[q0i] : 0.f;
        const float na1 = live1 ? qnorm[q1i] : 0.f;
        BoundX1 b0, b1;
        b0.e = MF_C_X1 * (na0 + nbmax) + MF_TINY;
        b1.e = MF_C_X1 * (na1 + nbmax) + MF_TINY;
        b0.Y = b1.Y = b0.M = b1.M = -__builtin_inff();
        // every piece starts from the bound of list2's first rows (match_warm_kernel), which are
        // that kernel's to emit
        b0.Y = live0 ? ywarm[q0i] : b0.Y;
        b1.Y = live1 ? ywarm[q1i] : b1.Y;
        b0.T = b0.Y - b0.e;
        b1.T = b1.Y - b1.e;
        b0.mask = b1.mask = 0;
        b0.cnt = b1.cnt = 0;
        const int sub = 2 * piece + half;  // this lane's sub-list of its two queries
        int *count0 = cand_count + (size_t)(live0 ? q0i : 0) * nsub + sub;
        int *count1 = cand_count + (size_t)(live1 ? q1i : 0) * nsub + sub;
        int *list0 = cand + ((size_t)(live0 ? q0i : 0) * nsub + sub) * subcap;
        int *list1 = cand + ((size_t)(live1 ? q1i : 0) * nsub + sub) * subcap;

        // Everything a block needs -- its 8 A fragments and its 16 seeds -h, both from the
        // staged tile -- is read into registers one whole block (>= 512 matrix cycles) ahead of
        // its MFMA chain; two register sets alternate.  The loop has no global loads besides
        // the staging (a younger global load that is waited for would drag the staging loads'
        // latency in).  LDS tiles are double buffered with one barrier per tile, placed before
        // the tile's last block: by then every wave has its last fragments of this tile in
        // registers, so the buffer can be refilled (tile + 2) and the next tile's first block
        // read.  (After the last tile's barrier nobody reads LDS any more, so the next piece
        // may stage at once.)
        stage(t_begin, tile0);
        staged();
        __syncthreads();
        if (ntiles > 1) stage(t_begin + 1, tile1);
        uint4 fA[8], fB[8];
        f32x16 x0, x1, y0, y1;  // two accumulator pairs: one computes while the other is closed
        load_frag(tile0 + lane, fA);
        f32x16 c = load_seeds(tile0, 0, half);
#define X1_BLOCK(FIRST, F, X0, X1, P0, P1, SEEDS, KB, JPREV)                                         \
    x1_block<FIRST>(F, qh0, qh1, X0, X1, P0, P1, c, SEEDS, KB, b0, b1, live0, live1, JPREV, half, n2, \
                    list0, list1, subcap)
        // one tile: `cur` is the buffer being consumed, `oth` the one tile tt + 1 lands in
        auto tile_step = [&](int tt, uint4 *cur, uint4 *oth) __attribute__((always_inline)) {
            const uint4 *bf = cur + lane;
            const int j0 = (t_begin + tt) * MF_TILE_ROWS1;
            load_frag(bf + 512, fB);
            __builtin_amdgcn_sched_barrier(0);
            if (tt == 0)
                X1_BLOCK(true, fA, x0, x1, y0, y1, cur, 1, 0);
            else
                X1_BLOCK(false, fA, x0, x1, y0, y1, cur, 1, j0 - MF_JB);
            load_frag(bf + 1024, fA);
            __builtin_amdgcn_sched_barrier(0);
            X1_BLOCK(false, fB, y0, y1, x0, x1, cur, 2, j0);
            load_frag(bf + 1536, fB);
            __builtin_amdgcn_sched_barrier(0);
            X1_BLOCK(false, fA, x0, x1, y0, y1, cur, 3, j0 + MF_JB);
            staged();         // this wave's share of tile tt + 1, issued one tile ago
            __syncthreads();  // tile tt + 1 has landed; nobody reads tile tt from LDS any more
            if (tt + 2 < ntiles) stage(t_begin + tt + 2, cur);
            if (tt + 1 < ntiles) load_frag(oth + lane, fA);
            __builtin_amdgcn_sched_barrier(0);
            // (after the piece's last tile the other buffer holds stale seeds: read, never used)
            X1_BLOCK(false, fB, y0, y1, x0, x1, oth, 0, j0 + 2 * MF_JB);
        };
#pragma unroll 1
        for (int tt = 0; tt < ntiles; tt += 2) {
            tile_step(tt, tile0, tile1);
            if (tt + 1 < ntiles) tile_step(tt + 1, tile1, tile0);
        }
#undef X1_BLOCK
        {   // drain: the last block (in y0, y1)
            const int j0 = (t_end - 1) * MF_TILE_ROWS1 + 3 * MF_JB;
            mfma_results_settle(y0, y1);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(b0.M) : "v"(y0[r]), "v"(y0[r + 1]));
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(b1.M) : "v"(y1[r]), "v"(y1[r + 1]));
            }
            x1_close(y0, y1, b0, b1, live0, live1, j0, half, n2, list0, list1, subcap);
        }
        if (live0) *count0 = b0.cnt;
        if (live1) *count1 = b1.cnt;
    }
}

// Phase 2: exact replay, RP_G queries per wave (RP_L lanes each).  A query has a few tens of
// candidates, so a whole wave per query would leave most lanes idle through the 384-operation
// distance chains and all but one through the final scan.
constexpr int RP_G = 4;
constexpr int RP_L = 64 / RP_G;

__global__ __launch_bounds__(64) void match_replay_kernel(
    const float *__restrict__ desc1, const float *__restrict__ score1, int n1,
    const float *__restrict__ desc2, const float *__restrict__ score2, int n2, float dist_limit,
    const int *__restrict__ cand_count, const int *__restrict__ cand, int nsub, int subcap,
    int *__restrict__ overflow_count,
    int *__restrict__ overflow_list, int32_t *__restrict__ closest_out, int32_t *__restrict__ second_out,
    float *__restrict__ min_dist_out, float *__restrict__ second_dist_out, uint8_t *__restrict__ status_out)
{
    __shared__ int cj_[RP_G][MF_CAP];
    __shared__ float dist_[RP_G][MF_CAP];
    __shared__ float q_[RP_G][K];
    const int g = threadIdx.x / RP_L;
    const int l = threadIdx.x % RP_L;
    const int i = blockIdx.x * RP_G + g;
    int *cj = cj_[g];
    float *dist = dist_[g], *q = q_[g];
    // the query's candidates: nsub sub-lists of at most subcap entries (one list for the atomic
    // kernels, one per (segment, half) for the x1 kernel)
    int count = 0;
    bool live = i < n1;
    if (live) {
        bool overflow = false;
        for (int s = 0; s < nsub; ++s) {
            const int c = cand_count[(size_t)i * nsub + s];
            overflow |= c > subcap;
            count += c;
        }
        if (overflow) {  // a sub-list ran out of slots (too many near-ties): exact scan for this query
            if (l == 0) overflow_list[atomicAdd(overflow_count, 1)] = i;
            live = false;
        }
    }
    if (!live) count = 0;
    // more candidates than the LDS arrays hold (but every sub-list complete): walked in place below
    const bool in_lds = count <= MF_CAP;
    if (live && in_lds) {
        for (int s = 0, base = 0; s < nsub; ++s) {
            const int c = cand_count[(size_t)i * nsub + s];
            const int *list = cand + ((size_t)i * nsub + s) * subcap;
            for (int p = l; p < c; p += RP_L) cj[base + p] = list[p];
            base += c;
        }
    }
    if (live)
        for (int k = l; k < K; k += RP_L) q[k] = desc1[(size_t)i * K + k];
    __syncthreads();
    // Exact distances of the candidates, in any order.  The reference's scan over ascending j
    // (Features.cpp:1036-1047) needs no sort to replay: its `closest` is the first index of the
    // smallest distance and its `second` the first index of the smallest distance among j <
    // closest -- both are records, hence candidates, and every other candidate is no smaller --
    // so two lexicographic (distance, index) minima give them.  NaN: list2[0] is taken
    // unconditionally (:1041, minDist == -1), so a NaN there sticks; any later NaN never
    // satisfies dist < minDist and counts as +inf.  (list2[0] is always a candidate: a piece
    // starts with a cold bound that passes its whole first block.)
    const float inf = __builtin_inff();
    float bd = inf, d0 = 0.0f;
    int bj = 0x7FFFFFFF;
    bool nan0 = false;
    auto exact_dist = [&](int j) {
        const float4 *b = reinterpret_cast<const float4 *>(desc2 + (size_t)j * K);
        float norm = 0.0f;
#pragma unroll 8
        for (int k4 = 0; k4 < K / 4; ++k4) {  // reference order: k ascending, one accumulator
            const float4 bv = b[k4];
            const float4 qv = reinterpret_cast<const float4 *>(q)[k4];
            float d;
            d = qv.x - bv.x; norm = norm + d * d;
            d = qv.y - bv.y; norm = norm + d * d;
            d = qv.z - bv.z; norm = norm + d * d;
            d = qv.w - bv.w; norm = norm + d * d;
        }
        return sqrtf(norm);
    };
    if (in_lds) {
#pragma unroll 1
        for (int p = l; p < count; p += RP_L) {
            const int j = cj[p];
            const float dt = exact_dist(j);
            dist[p] = dt;
            if (j == 0) { nan0 = dt != dt; d0 = dt; }
            const float dd = dt == dt ? dt : inf;
            if (dd < bd || (dd == bd && j < bj)) { bd = dd; bj = j; }
        }
    } else {
#pragma unroll 1
        for (int s = 0; s < nsub; ++s) {
            const int c = cand_count[(size_t)i * nsub + s];
            const int *list = cand + ((size_t)i * nsub + s) * subcap;
#pragma unroll 1
            for (int p = l; p < c; p += RP_L) {
                const int j = list[p];
                const float dt = exact_dist(j);
                if (j == 0) { nan0 = dt != dt; d0 = dt; }
                const float dd = dt == dt ? dt : inf;
                if (dd < bd || (dd == bd && j < bj)) { bd = dd; bj = j; }
            }
        }
    }
    // reduce over the query's RP_L lanes (xor butterflies stay inside the aligned group)
#pragma unroll
    for (int m = RP_L / 2; m > 0; m >>= 1) {
        const float od = __shfl_xor(bd, m);
        const int oj = __shfl_xor(bj, m);
        if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
        nan0 |= (bool)__shfl_xor((int)nan0, m);
        const float o0 = __shfl_xor(d0, m);
        d0 = (o0 != 0.0f || o0 != o0) ? o0 : d0;  // only the lane that met j == 0 holds a value
    }
    __syncthreads();
    int closest = -1, second = -1;
    float min_dist = -1.0f, second_dist = -1.0f;
    if (count > 0) {
        if (nan0) {
            closest = 0;
            min_dist = d0;
        } else {
            closest = bj;
            // the true value (bd is it unless every candidate is NaN-or-inf, where +inf is also right
            // only if list2[0] is +inf: then closest == 0 and dist of j == 0 is what the scan holds)
            min_dist = bd;
            float sd = inf;
            int sj = 0x7FFFFFFF;
            if (in_lds) {
#pragma unroll 1
                for (int p = l; p < count; p += RP_L) {
                    const int j = cj[p];
                    if (j < closest) {
                        const float dt = dist[p];
                        const float dd = dt == dt ? dt : inf;
                        if (dd < sd || (dd == sd && j < sj)) { sd = dd; sj = j; }
                    }
                }
            } else {  // long list: the distances of the candidates below `closest` once more
#pragma unroll 1
                for (int s = 0; s < nsub; ++s) {
                    const int c = cand_count[(size_t)i * nsub + s];
                    const int *list = cand + ((size_t)i * nsub + s) * subcap;
#pragma unroll 1
                    for (int p = l; p < c; p += RP_L) {
                        const int j = list[p];
                        if (j < closest) {
                            const float dt = exact_dist(j);
                            const float dd = dt == dt ? dt : inf;
                            if (dd < sd || (dd == sd && j < sj)) { sd = dd; sj = j; }
                        }
                    }
                }
            }
#pragma unroll
            for (int m = RP_L / 2; m > 0; m >>= 1) {
                const float od = __shfl_xor(sd, m);
                const int oj = __shfl_xor(sj, m);
                if (od < sd || (od == sd && oj < sj)) { sd = od; sj = oj; }
            }
            if (sj != 0x7FFFFFFF) {
                second = sj;
                second_dist = sd;
            }
        }
    }
    if (l != 0 || !live) return;
    if (closest_out) closest_out[i] = closest;
    if (second_out) second_out[i] = second;
    if (min_dist_out) min_dist_out[i] = min_dist;
    if (second_dist_out) second_dist_out[i] = second_dist;
    if (status_out)
        status_out[i] = match_status(closest, second, min_dist, second_dist, score1[i], score2, dist_limit);
}

// ---------------------------------------------------------------------------
// Small problems (a few hundred features per image, BASELINE config 1): the scan of the exact
// kernel is one dependent chain of n2 x 384 operations per query, ~2 us per row, so 200 rows
// take 0.4 ms whatever the GPU.  Here RP_L lanes share a query and take every RP_L-th row of
// list2; the two lexicographic (distance, index) minima of the replay kernel -- the same
// argument: the scan's `closest` is the first index of the smallest distance, its `second` the
// first index of the smallest distance among j < closest -- then replay the scan over ALL rows.
// Distances are in the reference's operation order; they are kept in LDS for the second minimum
// when n2 fits (STORE), else recomputed.
// ---------------------------------------------------------------------------
constexpr int SMALL_STORE_MAX = 8192;  // rows whose distances fit the LDS next to the queries (4 x 32 KiB)

__device__ __forceinline__ float small_dist(const float *__restrict__ q, const float *__restrict__ brow)
{
    const float4 *b = reinterpret_cast<const float4 *>(brow);
    float norm = 0.0f;
#pragma unroll 8
    for (int k4 = 0; k4 < K / 4; ++k4) {  // reference order: k ascending, one accumulator
        const float4 bv = b[k4];
        const float4 qv = reinterpret_cast<const float4 *>(q)[k4];
        float d;
        d = qv.x - bv.x; norm = norm + d * d;
        d = qv.y - bv.y; norm = norm + d * d;
        d = qv.z - bv.z; norm = norm + d * d;
        d = qv.w - bv.w; norm = norm + d * d;
    }
    return sqrtf(norm);
}

// the same for rows j, j + RP_L, j + 2 RP_L, j + 3 RP_L at once (rows beyond n2 repeat row j; unused)
__device__ __forceinline__ void small_dist4(const float *__restrict__ q, const float *__restrict__ desc2, int j, int n2,
                                            float (&out)[4])
{
    const float4 *b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int jr = j + r * RP_L < n2 ? j + r * RP_L : j;
        b[r] = reinterpret_cast<const float4 *>(desc2 + (size_t)jr * K);
    }
    float norm[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (int k4 = 0; k4 < K / 4; ++k4) {  // reference order per row: k ascending, one accumulator
        const float4 qv = reinterpret_cast<const float4 *>(q)[k4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 bv = b[r][k4];
            float d;
            d = qv.x - bv.x; norm[r] = norm[r] + d * d;
            d = qv.y - bv.y; norm[r] = norm[r] + d * d;
            d = qv.z - bv.z; norm[r] = norm[r] + d * d;
            d = qv.w - bv.w; norm[r] = norm[r] + d * d;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = sqrtf(norm[r]);
}

template <bool STORE>
__global__ __launch_bounds__(64) void match_small_kernel(
    const float *__restrict__ desc1, const float *__restrict__ score1, int n1,
    const float *__restrict__ desc2, const float *__restrict__ score2, int n2, float dist_limit,
    int32_t *__restrict__ closest_out, int32_t *__restrict__ second_out,
    float *__restrict__ min_dist_out, float *__restrict__ second_dist_out, uint8_t *__restrict__ status_out)
{
    extern __shared__ float small_lds[];  // STORE: [RP_G][n2] distances
    __shared__ __attribute__((aligned(16))) float q_[RP_G][K];
    const int g = threadIdx.x / RP_L;
    const int l = threadIdx.x % RP_L;
    const int i = blockIdx.x * RP_G + g;
    const bool live = i < n1;
    float *q = q_[g];
    float *dist = small_lds + (size_t)g * (STORE ? n2 : 0);
    if (live)
        for (int k = l; k < K; k += RP_L) q[k] = desc1[(size_t)i * K + k];
    __syncthreads();
    const float inf = __builtin_inff();
    float bd = inf, d0 = 0.0f;
    int bj = 0x7FFFFFFF;
    bool nan0 = false;
    if (live) {
        // four rows per lane at a time: each distance is a dependent chain of 128 additions, and four
        // independent chains fill the latency of one
#pragma unroll 1
        for (int j = l; j < n2; j += 4 * RP_L) {
            float dt4[4];
            small_dist4(q, desc2, j, n2, dt4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jr = j + r * RP_L;
                if (jr < n2) {
                    const float dt = dt4[r];
                    if (STORE) dist[jr] = dt;
                    if (jr == 0) { nan0 = dt != dt; d0 = dt; }  // list2[0] is taken unconditionally (:1041)
                    const float dd = dt == dt ? dt : inf;      // a later NaN never satisfies dist < minDist
                    if (dd < bd || (dd == bd && jr < bj)) { bd = dd; bj = jr; }  // (+inf rows too: the first one stays)
                }
            }
        }
    }
#pragma unroll
    for (int m = RP_L / 2; m > 0; m >>= 1) {
        const float od = __shfl_xor(bd, m);
        const int oj = __shfl_xor(bj, m);
        if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
        nan0 |= (bool)__shfl_xor((int)nan0, m);
        const float o0 = __shfl_xor(d0, m);
        d0 = (o0 != 0.0f || o0 != o0) ? o0 : d0;  // only the lane that met j == 0 holds a value
    }
    __syncthreads();  // (STORE) the group's distances are visible to all its lanes
    int closest = -1, second = -1;
    float min_dist = -1.0f, second_dist = -1.0f;
    if (live && n2 > 0) {
        if (nan0) {
            closest = 0;
            min_dist = d0;
        } else {
            closest = bj;
            min_dist = bd;
            float sd = inf;
            int sj = 0x7FFFFFFF;
#pragma unroll 1
            for (int j = l; j < closest; j += RP_L) {
                const float dt = STORE ? dist[j] : small_dist(q, desc2 + (size_t)j * K);
                const float dd = dt == dt ? dt : inf;
                if (dd < sd || (dd == sd && j < sj)) { sd = dd; sj = j; }
            }
#pragma unroll
            for (int m = RP_L / 2; m > 0; m >>= 1) {
                const float od = __shfl_xor(sd, m);
                const int oj = __shfl_xor(sj, m);
                if (od < sd || (od == sd && oj < sj)) { sd = od; sj = oj; }
            }
            if (sj != 0x7FFFFFFF) {
                second = sj;
                second_dist = sd;
            }
        }
    }
    if (l != 0 || !live) return;
    if (closest_out) closest_out[i] = closest;
    if (second_out) second_out[i] = second;
    if (min_dist_out) min_dist_out[i] = min_dist;
    if (second_dist_out) second_dist_out[i] = second_dist;
    if (status_out)
        status_out[i] = match_status(closest, second, min_dist, second_dist, score1[i], score2, dist_limit);
}

}  // namespace

static int launch_exact(stx_ctx *ctx, const float *d_desc1, const float *d_score1, int n1,
                        const float *d_desc2, const float *d_score2, int n2, float dist_limit,
                        const int32_t *query_list, const int *n_queries_dev, int32_t *d_closest,
                        int32_t *d_second, float *d_min_dist, float *d_second_dist, uint8_t *d_status)
{
    const int blocks = (n1 + EXACT_THREADS - 1) / EXACT_THREADS;
    hipLaunchKernelGGL(match_exact_kernel, dim3(blocks), dim3(EXACT_THREADS), 0, ctx->stream, d_desc1,
                       d_score1, n1, d_desc2, d_score2, n2, dist_limit, query_list, n_queries_dev, n1,
                       d_closest, d_second, d_min_dist, d_second_dist, d_status);
    return (int)hipGetLastError();
}

static int launch_small(stx_ctx *ctx, const float *d_desc1, const float *d_score1, int n1, const float *d_desc2,
                        const float *d_score2, int n2, float dist_limit, int32_t *d_closest, int32_t *d_second,
                        float *d_min_dist, float *d_second_dist, uint8_t *d_status)
{
    const int blocks = (n1 + RP_G - 1) / RP_G;
    if (n2 <= SMALL_STORE_MAX) {
        const int lds = (int)(sizeof(float) * RP_G * (size_t)n2);
        const int rc = stx_set_lds_attr(ctx, reinterpret_cast<const void *>(match_small_kernel<true>), lds);
        if (rc != STX_OK) return rc;
        hipLaunchKernelGGL(match_small_kernel<true>, dim3(blocks), dim3(64), lds, ctx->stream, d_desc1, d_score1, n1,
                           d_desc2, d_score2, n2, dist_limit, d_closest, d_second, d_min_dist, d_second_dist, d_status);
    } else {
        hipLaunchKernelGGL(match_small_kernel<false>, dim3(blocks), dim3(64), 0, ctx->stream, d_desc1, d_score1, n1,
                           d_desc2, d_score2, n2, dist_limit, d_closest, d_second, d_min_dist, d_second_dist, d_status);
    }
    return (int)hipGetLastError();
}

int stx_launch_match(stx_ctx *ctx, const float *d_desc1, const float *d_score1, int n1,
                     const float *d_desc2, const float *d_score2, int n2, float dist_limit,
                     int32_t *d_closest, int32_t *d_second, float *d_min_dist,
                     float *d_second_dist, uint8_t *d_status)
{
    ctx->m_last_n1 = 0;  // stx_match_stats describes prefilter calls only
    if (n1 <= 0) return STX_OK;
    int algo = ctx->match_algo;
    if (algo == STX_MATCH_SMALL && n2 > 0)
        return launch_small(ctx, d_desc1, d_score1, n1, d_desc2, d_score2, n2, dist_limit, d_closest, d_second,
                            d_min_dist, d_second_dist, d_status);
    if (algo == STX_MATCH_SMALL) algo = STX_MATCH_EXACT;
    if (algo == STX_MATCH_AUTO) {
        // measured (tools/exp/match_crossover.py): the small kernel takes ~9 us + 0.1 us per row of list2
        // while its workgroups fit the chip at once, the prefilter + replay ~60 us + the matrix work
        if (n2 == 0)
            algo = STX_MATCH_EXACT;
        else if (n2 <= 640 && (double)n1 * (double)n2 <= 4.0e6)
            return launch_small(ctx, d_desc1, d_score1, n1, d_desc2, d_score2, n2, dist_limit, d_closest, d_second,
                                d_min_dist, d_second_dist, d_status);
        else
            algo = STX_MATCH_MFMA_BF16X1;
    }
    if (algo == STX_MATCH_EXACT || n2 == 0)
        return launch_exact(ctx, d_desc1, d_score1, n1, d_desc2, d_score2, n2, dist_limit, nullptr,
                            nullptr, d_closest, d_second, d_min_dist, d_second_dist, d_status);

    // ---- MFMA prefilter + replay ------------------------------------------
    const bool x1 = algo == STX_MATCH_MFMA_BF16X1;
    const int tile_rows = x1 ? MF_TILE_ROWS1 : MF_TILE_ROWS;
    int tiles_total = (n2 + tile_rows - 1) / tile_rows;
    // x1: the workgroups' ranges start at multiples of tiles_total / 32 (see below)
    if (x1 && tiles_total >= 256) tiles_total = (tiles_total + 31) / 32 * 32;
    const int n2pad = tiles_total * tile_rows;
    const int qblocks = x1 ? (n1 + X1_QB - 1) / X1_QB : (n1 + MF_QB - 1) / MF_QB;
    // segments of list2: fill the CUs evenly (2 workgroups per CU), keep segments long
    const int slots = (x1 ? 1 : 2) * (ctx->num_cus > 0 ? ctx->num_cus : 256);
    int segs = 1;
    double best_eff = 0;
    for (int s = 1; s <= 6; ++s) {
        if (s > 1 && tiles_total / s < 16) break;
        const long wgs = (long)qblocks * s;
        const double eff = (double)wgs / (double)(((wgs + slots - 1) / slots) * slots);
        if (eff > best_eff + 0.02) { best_eff = eff; segs = s; }
    }
    const int tiles_per_seg = (tiles_total + segs - 1) / segs;
    segs = (tiles_total + tiles_per_seg - 1) / tiles_per_seg;

    int err;
    // [frag: n2pad*K floats, or x1 tiles of MF_TILE_U4 uint4][bnorm n2pad][bmax 1 (+pad)]
    const size_t frag_floats = x1 ? (size_t)tiles_total * MF_TILE_U4 * 4 : (size_t)n2pad * K;
    // (+ x1: the 32-row blocks' largest norms, n2pad / 32 floats)
    float *frag = (float *)stx_workspace(ctx, STX_SLOT_M_FRAG,
                                         sizeof(float) * (frag_floats + (size_t)n2pad + 4 + (size_t)n2pad / MF_JB), &err);
    if (!frag) return err;
    float *bnorm = frag + frag_floats;
    float *block_max = bnorm + n2pad + 4;
    // [qnorm n1][ywarm n1]
    float *qnorm = (float *)stx_workspace(ctx, STX_SLOT_M_QNORM, sizeof(float) * 2 * (size_t)n1, &err);
    if (!qnorm) return err;
    float *ywarm = qnorm + n1;
    // x1 kernel: one contiguous range of `per` tiles of the flattened (query block, tile) plane
    // per workgroup, 2 workgroups per CU; a query block is cut into at most max_pieces pieces
    const long long flat_total = (long long)qblocks * tiles_total;
    long long per = (flat_total + slots - 1) / slots;
    if (per < (tiles_total + 6) / 7) per = (tiles_total + 6) / 7;
    // Workgroups b and b + 8 share an XCD and its L2.  With every range starting at a multiple
    // of tiles_total / 32, the 64 workgroups of an XCD stream list2 at no more than 4 distinct
    // positions, in step, so all but the first reader of a tile hit in L2.
    // (Not when the rounding would leave CUs without work: with few query blocks -- list1 sharded
    // over GPUs -- a range is a fraction of list2 and rounding it up to the next 1/32 can idle an
    // eighth of the chip, more than the alignment is worth.)
    if (tiles_total % 32 == 0 && tiles_total >= 256) {
        const long long u = tiles_total / 32;
        const long long per_aligned = (per + u - 1) / u * u;
        const long long wgs_aligned = (flat_total + per_aligned - 1) / per_aligned;
        if (wgs_aligned * 100 >= (long long)slots * 94) per = per_aligned;
    }
    int x1_wgs = (int)((flat_total + per - 1) / per);
    int max_pieces = (int)((tiles_total + per - 1) / per) + 1;
    // few query blocks: the (segment, query block) grid of the kernel's comment, when it fills the chip
    int grid_segs = 0;
    if (x1 && slots % 8 == 0 && qblocks * 2 <= slots && tiles_total >= 256) {
        const int S = slots / qblocks;
        if ((long long)qblocks * S * 100 >= (long long)slots * 94 && getenv("STX_MATCH_FLAT_CUT") == nullptr) {
            grid_segs = S;
            per = (tiles_total + S - 1) / S;
            x1_wgs = slots;
            max_pieces = S;
        }
    }
    // candidate lists: one list of MF_CAP slots per query filled with atomics (fp32 / bf16x3
    // kernels), or two private sub-lists per piece (x1 kernel)
    const int nsub = x1 ? 2 * max_pieces + 1 : 1;  // x1: + the warm kernel's list
    // Slots per sub-list.  A sub-list that overflows sends its query to the overflow pass (an exact scan
    // of all of list2: 2.4 us per such query at n2 = 200 000), a slot that stays empty costs nothing but
    // address space, so the lists take what MF_LIST_BYTES allows (512 slots at 200k x 200k, 2 GB of the
    // 288): a cluster of ~500 near-identical rows then stays in its lists (clustered 200k x 200k:
    // 1 223 overflowing queries and 2.9 ms of overflow pass before).  STX_MATCH_SUBCAP (tests only)
    // forces a value, so that small shapes still reach the overflow pass.
    int subcap = MF_CAP;
    if (x1) {
        subcap = nsub <= 8 ? 128 : nsub <= 16 ? 64 : 32;
        const size_t fit = MF_LIST_BYTES / (sizeof(int) * (size_t)n1 * (size_t)nsub);
        while (subcap < 1024 && (size_t)subcap * 2 <= fit) subcap *= 2;
        const char *forced = getenv("STX_MATCH_SUBCAP");
        if (forced && atoi(forced) >= 8 && atoi(forced) <= 4096) subcap = atoi(forced);
    }
    // [count n1*nsub][overflow_count 1 (+pad)][overflow_list n1][cand n1*nsub*subcap]
    const size_t n1a = ((size_t)n1 + 3) & ~(size_t)3;
    const size_t ncnt = ((size_t)n1 * nsub + 3) & ~(size_t)3;
    int *cnt = (int *)stx_workspace(ctx, STX_SLOT_M_CAND,
                                    sizeof(int) * (ncnt + 4 + n1a + (size_t)n1 * nsub * subcap), &err);
    if (!cnt) return err;
    int *ovf_count = cnt + ncnt;
    int *ovf_list = ovf_count + 4;
    int *cand = ovf_list + n1a;
    hipStream_t st = ctx->stream;
    // (the x1 kernel stores the counts of the sub-lists it uses; the others must read 0: its warm
    // kernel zeroes them, per query, and the overflow count)
    if (!x1) STX_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (ncnt + 4), st));
    const bool bf16 = algo == STX_MATCH_MFMA_BF16;
    unsigned *bmax = reinterpret_cast<unsigned *>(bnorm + n2pad);
    if (x1)  // norms, seeds, fragments and the largest norm in one pass over list2
    {
        hipLaunchKernelGGL(match_prep_x1_kernel, dim3((unsigned)(n2pad / MF_JB)), dim3(256), 0, st, d_desc2, n2, n2pad,
                           bnorm, block_max, reinterpret_cast<uint4 *>(frag));
        hipLaunchKernelGGL(match_max_kernel, dim3(1), dim3(1024), 0, st, block_max, n2pad / MF_JB,
                           reinterpret_cast<float *>(bmax));
    }
    else
        hipLaunchKernelGGL(match_norm_kernel, dim3((n2pad + 255) / 256), dim3(256), 0, st, d_desc2, n2, n2pad, bnorm);
    if (x1)
        ;  // (fragments written above)
    else if (bf16)
        hipLaunchKernelGGL(match_prep_frag_bf16_kernel, dim3((unsigned)(((size_t)n2pad / MF_JB * 512 + 255) / 256)),
                           dim3(256), 0, st, d_desc2, n2, n2pad, reinterpret_cast<uint4 *>(frag));
    else
        hipLaunchKernelGGL(match_prep_frag_kernel, dim3((unsigned)(((size_t)n2pad * K / 4 + 255) / 256)), dim3(256),
                           0, st, d_desc2, n2, n2pad, frag);
    if (!x1)
        hipLaunchKernelGGL(match_norm_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, d_desc1, n1, n1, qnorm);
    if (x1) {
        hipLaunchKernelGGL(match_warm_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, d_desc1, qnorm, n1, d_desc2, n2,
                           bnorm, reinterpret_cast<const float *>(bmax), ywarm, nsub, subcap, cnt, cand, ovf_count);
        hipLaunchKernelGGL(match_mfma_bf16x1_kernel, dim3(x1_wgs), dim3(X1_WAVES * 64), 0, st, d_desc1, qnorm,
                           n1, reinterpret_cast<const uint4 *>(frag), reinterpret_cast<const float *>(bmax), n2,
                           tiles_total, per, flat_total, grid_segs, nsub, subcap, cnt, cand, ywarm);
    }
    else if (bf16)
        hipLaunchKernelGGL(match_mfma_bf16_kernel, dim3(qblocks, segs), dim3(MF_WAVES * 64), 0, st, d_desc1, qnorm,
                           n1, reinterpret_cast<const uint4 *>(frag), bnorm, n2, tiles_total, tiles_per_seg, cnt,
                           cand);
    else
        hipLaunchKernelGGL(match_mfma_kernel, dim3(qblocks, segs), dim3(MF_WAVES * 64), 0, st, d_desc1, qnorm, n1,
                           frag, bnorm, n2, tiles_total, tiles_per_seg, cnt, cand);
    hipLaunchKernelGGL(match_replay_kernel, dim3((n1 + RP_G - 1) / RP_G), dim3(64), 0, st, d_desc1, d_score1, n1, d_desc2, d_score2,
                       n2, dist_limit, cnt, cand, nsub, subcap, ovf_count, ovf_list, d_closest, d_second, d_min_dist,
                       d_second_dist, d_status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    ctx->m_last_counts = cnt;
    ctx->m_last_overflow = ovf_count;
    ctx->m_last_n1 = n1;
    ctx->m_last_nsub = nsub;
    // queries with an overflowed sub-list (count read on the device): the exact scan cut along list2
    const size_t part_entries = (size_t)(n1 + 64 > 64 * OVF_WAVES ? n1 + 64 : 64 * OVF_WAVES);
    OvfPartial *part = (OvfPartial *)stx_workspace(ctx, STX_SLOT_M_FALLBACK, sizeof(OvfPartial) * part_entries, &err);
    if (!part) return err;
    hipLaunchKernelGGL(match_exact_seg_kernel, dim3(OVF_WAVES), dim3(64), 0, st, d_desc1, d_desc2, n2, ovf_list,
                       ovf_count, part);
    hipLaunchKernelGGL(match_exact_merge_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, d_score1, d_score2, n2,
                       dist_limit, ovf_list, ovf_count, part, d_closest, d_second, d_min_dist, d_second_dist, d_status);
    return (int)hipGetLastError();
}

// Candidate statistics of the context's last prefilter call (diagnostics: SURVEY 8d asks for the
// replay's cost as candidates per query).  Waits for the stream.
int stx_match_stats_impl(stx_ctx *ctx, long long *total_candidates, int *max_candidates, int *overflow_queries,
                         int *n_queries)
{
    *total_candidates = 0;
    *max_candidates = *overflow_queries = *n_queries = 0;
    if (!ctx->m_last_counts || ctx->m_last_n1 <= 0) return STX_OK;
    const int n1 = ctx->m_last_n1, nsub = ctx->m_last_nsub;
    int *h = (int *)malloc(sizeof(int) * ((size_t)n1 * nsub + 1));
    if (!h) return STX_ERR_HOST_OOM;
    hipError_t e = hipMemcpyAsync(h, ctx->m_last_counts, sizeof(int) * (size_t)n1 * nsub, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(h + (size_t)n1 * nsub, ctx->m_last_overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(h); return (int)e; }
    long long tot = 0;
    int mx = 0;
    for (int i = 0; i < n1; ++i) {
        int c = 0;
        for (int s = 0; s < nsub; ++s) c += h[(size_t)i * nsub + s];
        tot += c;
        if (c > mx) mx = c;
    }
    *total_candidates = tot;
    *max_candidates = mx;
    *overflow_queries = h[(size_t)n1 * nsub];
    *n_queries = n1;
    free(h);
    return STX_OK;
}

// ---------------------------------------------------------------------------
// Array-of-structures input (the reference's std::vector<Feature>, Features.h:52-84: 540-byte
// records, descriptor at byte offset 20, score at 12): the records are uploaded as they lie in
// the caller's vector and split into the descriptor and score arrays here, instead of a host
// loop that touches every float once more.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void match_unpack_aos_kernel(const unsigned char *__restrict__ rec, int n,
                                                               size_t stride, size_t desc_off, size_t score_off,
                                                               float *__restrict__ desc, float *__restrict__ score)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // (feature, k)
    const size_t i = idx / K;
    const int k = (int)(idx % K);
    if (i >= (size_t)n) return;
    const unsigned char *r = rec + i * stride;
    desc[idx] = *reinterpret_cast<const float *>(r + desc_off + 4 * (size_t)k);  // offsets are multiples of 4
    if (k == 0) score[i] = *reinterpret_cast<const float *>(r + score_off);
}
}  // namespace

int stx_launch_unpack_features(stx_ctx *ctx, const void *d_rec, int n, size_t stride, size_t desc_off,
                               size_t score_off, float *d_desc, float *d_score)
{
    if (n <= 0) return STX_OK;
    const size_t total = (size_t)n * K;
    hipLaunchKernelGGL(match_unpack_aos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const unsigned char *>(d_rec), n, stride, desc_off, score_off, d_desc, d_score);
    return (int)hipGetLastError();
}
