// Internal declarations shared by the translation units of libstereo_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/stereo_hip.h"

// Grow-only device workspace slots owned by a context.
enum stx_slot {
    STX_SLOT_IMG0 = 0,   // host-API staging: first input image(s)
    STX_SLOT_IMG1,       // host-API staging: second input image / rectify dst
    STX_SLOT_OUT,        // host-API staging: output
    STX_SLOT_DISP_KEYS,  // disparity: per-pixel best keys carried between passes (D > 256)
    STX_SLOT_DESC1,      // host-API staging: list1 descriptors
    STX_SLOT_DESC2,      // host-API staging: list2 descriptors
    STX_SLOT_SMALL,      // host-API staging: scores + per-query outputs
    STX_SLOT_HINV,       // rectify: per-image inverse homographies
    STX_SLOT_M_FRAG,     // match: list2 in MFMA fragment order + norms
    STX_SLOT_M_QNORM,    // match: query norms
    STX_SLOT_M_CAND,     // match: candidate lists + counters
    STX_SLOT_M_FALLBACK, // match: overflow query list
    STX_SLOT_FAST,       // FAST: corner bit words + prefix sums
    STX_SLOT_SCORE,      // scoring + NMS: scores, index lists, flags, counts
    STX_SLOT_SCORE_IO,   // host-API staging of the scoring call: gradient images, features, outputs
    STX_SLOT_SIFT_TABLES, // descriptors: atan table + Gaussian kernels (built once per context)
    STX_SLOT_RANSAC,     // hypothesis scoring: records + per-pair errors
    STX_SLOT_COUNT
};

struct stx_ctx {
    int device;
    hipStream_t stream;
    int match_algo;
    int force_bruteforce;  // tests: route every D1-ext call to the tap-by-tap kernel
    int num_cus;
    void *slot_ptr[STX_SLOT_COUNT];
    size_t slot_bytes[STX_SLOT_COUNT];
};

// Returns a device buffer of at least `bytes` bytes for `slot` (grow-only;
// growing synchronises the stream, frees and reallocates).  nullptr + *err on failure.
void *stx_workspace(stx_ctx *ctx, int slot, size_t bytes, int *err);

#define STX_HIP(call)                                   \
    do {                                                \
        hipError_t stx_e_ = (call);                     \
        if (stx_e_ != hipSuccess) return (int)stx_e_;   \
    } while (0)

// Kernel launchers (device pointers, asynchronous on ctx->stream).
int stx_launch_disparity(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1,
                         int n_pairs, int w, int h, size_t stride, size_t pair_stride,
                         int row0, int rows, int win, int max_disp, int cost_mode,
                         void *d_out, size_t out_stride, size_t out_pair_stride,
                         int out_bytes);

int stx_launch_rectify(stx_ctx *ctx, const uint8_t *d_src, int src_w, int src_h,
                       size_t src_stride, size_t src_image_stride, uint8_t *d_dst,
                       int dst_w, int dst_h, size_t dst_stride, size_t dst_image_stride,
                       const float *d_hinv, int n_images, int flags, int dst_y0 = 0);

int stx_launch_match(stx_ctx *ctx, const float *d_desc1, const float *d_score1, int n1,
                     const float *d_desc2, const float *d_score2, int n2,
                     float dist_limit, int32_t *d_closest, int32_t *d_second,
                     float *d_min_dist, float *d_second_dist, uint8_t *d_status);

int stx_launch_fast(stx_ctx *ctx, const uint8_t *d_img, int w, int h, size_t stride, int threshold,
                    float *d_xy, int max_out, int *d_count);

int stx_launch_score_cluster(stx_ctx *ctx, const uint8_t *d_gx, const uint8_t *d_gy, int w, int h,
                             size_t stride, const float *d_xy, int n, const float kernel3x3[9],
                             float score_threshold, float cluster_dist, float *d_xy_out,
                             float *d_score_out, int32_t *d_index_out, int *d_count);

int stx_launch_sift(stx_ctx *ctx, const uint8_t *d_gx, const uint8_t *d_gy, int w, int h, size_t stride,
                    const float *d_xy, int n, float *d_angle, float *d_desc);

int stx_launch_cluster(stx_ctx *ctx, const float *d_xy, const float *d_score, int n, float window,
                       int32_t *d_index_out, int *d_count);

int stx_launch_ransac_score(stx_ctx *ctx, const float *hyp, int n_hyp, const float K1[9], const float K2[9],
                            const float *d_p1, const float *d_p2, int n, const int32_t *exclude, float threshold,
                            int32_t *d_inliers, float *d_err_sum, float *d_errors, float *d_depths);
