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
#include "stx_internal.h"

namespace {

constexpr int K = STX_DESC_LEN;

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
    const int32_t *__restrict__ query_list, int n_queries, int32_t *__restrict__ closest_out,
    int32_t *__restrict__ second_out, float *__restrict__ min_dist_out,
    float *__restrict__ second_dist_out, uint8_t *__restrict__ status_out)
{
    const int slot = blockIdx.x * EXACT_THREADS + threadIdx.x;
    const bool active = slot < n_queries;
    // inactive lanes shadow the last query so that the row loop stays uniform
    const int qslot = active ? slot : n_queries - 1;
    const int i = query_list ? query_list[qslot] : qslot;

    float a[K];
    const float4 *arow = reinterpret_cast<const float4 *>(desc1 + (size_t)i * K);
#pragma unroll
    for (int k = 0; k < K / 4; ++k) {
        const float4 v = arow[k];
        a[4 * k + 0] = v.x;
        a[4 * k + 1] = v.y;
        a[4 * k + 2] = v.z;
        a[4 * k + 3] = v.w;
    }

    int closest = -1, second = -1;
    float min_dist = -1.0f;  // Features.cpp:1035
    float min_norm2 = 0.0f;
    for (int j = 0; j < n2; ++j) {
        const float *__restrict__ b = desc2 + (size_t)j * K;
        float norm = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d = a[k] - b[k];
            const float sq = d * d;
            norm = norm + sq;
        }
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
    if (!active) return;

    float second_dist = -1.0f;
    if (second >= 0) second_dist = sqrtf(ref_norm2(a, desc2 + (size_t)second * K));  // :1064
    if (closest_out) closest_out[i] = closest;
    if (second_out) second_out[i] = second;
    if (min_dist_out) min_dist_out[i] = min_dist;
    if (second_dist_out) second_dist_out[i] = second_dist;
    if (status_out)
        status_out[i] = match_status(closest, second, min_dist, second_dist, score1[i], score2, dist_limit);
}

}  // namespace

int stx_launch_match(stx_ctx *ctx, const float *d_desc1, const float *d_score1, int n1,
                     const float *d_desc2, const float *d_score2, int n2, float dist_limit,
                     int32_t *d_closest, int32_t *d_second, float *d_min_dist,
                     float *d_second_dist, uint8_t *d_status)
{
    if (n1 <= 0) return STX_OK;
    const int blocks = (n1 + EXACT_THREADS - 1) / EXACT_THREADS;
    hipLaunchKernelGGL(match_exact_kernel, dim3(blocks), dim3(EXACT_THREADS), 0, ctx->stream,
                       d_desc1, d_score1, n1, d_desc2, d_score2, n2, dist_limit,
                       (const int32_t *)nullptr, n1, d_closest, d_second, d_min_dist,
                       d_second_dist, d_status);
    return (int)hipGetLastError();
}
