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
    STX_SLOT_RES0,       // resident image 0 (stx_rectify_u8_keep -> stx_disparity_resident_u8)
    STX_SLOT_RES1,       // resident image 1
    STX_SLOT_AOS,        // stx_match_features: the caller's feature records as uploaded
    STX_SLOT_COUNT
};

struct stx_host_pool;  // stx_host.hip: worker threads for the host-pointer staging copies
#define STX_HOST_MAX_CHUNKS 8

struct stx_ctx {
    int device;
    hipStream_t stream;
    int match_algo;
    int force_bruteforce;  // tests: route every D1-ext call to the tap-by-tap kernel
    int force_colsum;      // tests: windows 17..31 through the column-sum kernel
    int num_cus;
    void *slot_ptr[STX_SLOT_COUNT];
    size_t slot_bytes[STX_SLOT_COUNT];
    // static device scratch of stx_disparity_dev_noalloc on this context's device
    // (hipGetSymbolAddress at stx_create; disparity.hip), and its size in bytes
    void *disp_scratch;
    size_t disp_scratch_bytes;
    int res_w[2], res_h[2];  // dimensions of the resident images (0: slot empty)
    // the last prefilter call's candidate counters (stx_match_stats): device pointers into
    // STX_SLOT_M_CAND, valid until the next matching call on this context
    const int *m_last_counts, *m_last_overflow;
    int m_last_n1, m_last_nsub;
    // Host-pointer entry points: pinned staging (grow-only), a second stream and events for
    // the chunked upload / compute / download pipeline, worker threads for the host copies.
    void *pin_ptr[2];  // [0] uploads, [1] downloads
    size_t pin_bytes[2];
    hipStream_t s_in, s_out;  // upload / download streams (kernels stay on `stream`)
    hipEvent_t ev_in[STX_HOST_MAX_CHUNKS], ev_k[STX_HOST_MAX_CHUNKS], ev_out[STX_HOST_MAX_CHUNKS];
    stx_host_pool *pool;
    int host_pipeline;  // 1 on, 0 plain hipMemcpyAsync of the caller's buffers, -1 ask the environment
    int host_threads;   // staging threads incl. the caller (0: STX_HOST_THREADS or a default)
    int host_chunks;    // chunks per call (0: STX_HOST_CHUNKS or by size)
};

// Makes ctx's device current for the calling thread and restores the caller's device when it
// goes out of scope: an entry point never leaves the thread on another GPU than it found it on.
struct stx_device_guard {
    int prev;
    int err;
    explicit stx_device_guard(int device) : prev(-1), err(STX_OK)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) { prev = -1; }
        if (prev != device) {
            e = hipSetDevice(device);
            if (e != hipSuccess) { err = (int)e; prev = -1; }
        } else {
            prev = -1;  // nothing to restore
        }
    }
    ~stx_device_guard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    stx_device_guard(const stx_device_guard &) = delete;
    stx_device_guard &operator=(const stx_device_guard &) = delete;
};
#define STX_ENTER(ctx)                       \
    stx_device_guard stx_guard_((ctx)->device); \
    if (stx_guard_.err) return stx_guard_.err

// Returns a device buffer of at least `bytes` bytes for `slot` (grow-only; growing waits for
// the whole device -- work enqueued on ANY stream may still use the old buffer -- then frees
// and reallocates).  nullptr + *err on failure.
void *stx_workspace(stx_ctx *ctx, int slot, size_t bytes, int *err);

// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, lds), issued once per (device, kernel).
int stx_set_lds_attr(stx_ctx *ctx, const void *fn, int lds);

// ---- stx_host.hip: the host-pointer data path (pinned staging, three streams, chunks) ----
// Releases the pinned staging, copy streams, events and worker threads.
void stx_host_release(stx_ctx *ctx);
bool stx_host_pipeline_on(stx_ctx *ctx);
int stx_host_prepare(stx_ctx *ctx, size_t in_bytes, size_t out_bytes);
bool stx_host_zero_copy_on();  // kernels may store small results straight into page-locked host memory
int stx_host_chunks(stx_ctx *ctx, size_t bytes, size_t units);
int stx_host_fence(stx_ctx *ctx);
bool stx_host_is_pinned(const void *p, size_t bytes);
void *stx_host_kernel_ptr(const void *p, size_t bytes);  // device-visible address on the current device, or nullptr
int stx_host_upload(stx_ctx *ctx, void *d_dst, void *pin, const void *src, size_t spitch, size_t wbytes,
                    size_t rows, hipEvent_t ev);
int stx_host_download_begin(stx_ctx *ctx, void *pin, const void *d_src, size_t bytes, hipEvent_t after,
                            hipEvent_t ev);
int stx_host_download_end(stx_ctx *ctx, void *dst, size_t dpitch, const void *pin, size_t wbytes, size_t rows,
                          hipEvent_t ev);
int stx_host_upload_flat(stx_ctx *ctx, void *d_dst, const void *src, size_t bytes, size_t pin_off);
int stx_disparity_host(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int n_pairs, int w, int h,
                       size_t stride, size_t pair_stride, int row0, int rows, int win, int max_disp,
                       int cost_mode, void *out, size_t out_stride, size_t out_pair_stride, int out_bytes);
// d_keep: device buffer (dense, dst_w x dst_h) that receives the warped image instead of the
// staging slot, e.g. a resident-image slot; nullptr = staging slot.
int stx_rectify_host(stx_ctx *ctx, const uint8_t *src, int src_w, int src_h, size_t src_stride, uint8_t *dst,
                     int dst_w, int dst_h, size_t dst_stride, const float Hinv[9], int flags,
                     uint8_t *d_keep = nullptr);
int stx_disparity_resident_host(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1, int w, int h, int win,
                                int max_disp, int cost_mode, void *out, size_t out_stride, int out_bytes);

// disparity.hip: device address and size of this device's static scratch array.
int stx_disparity_scratch(void **ptr, size_t *bytes);

#define STX_HIP(call)                                   \
    do {                                                \
        hipError_t stx_e_ = (call);                     \
        if (stx_e_ != hipSuccess) return (int)stx_e_;   \
    } while (0)

// Kernel launchers (device pointers, asynchronous on ctx->stream).
// keys/keys_bytes: caller-provided buffer for the per-pixel keys carried between passes
// (D > 256); nullptr = take it from the context's grow-only workspace.
int stx_launch_disparity(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1,
                         int n_pairs, int w, int h, size_t stride, size_t pair_stride,
                         int row0, int rows, int win, int max_disp, int cost_mode,
                         void *d_out, size_t out_stride, size_t out_pair_stride,
                         int out_bytes, void *keys = nullptr, size_t keys_bytes = 0);

int stx_launch_rectify(stx_ctx *ctx, const uint8_t *d_src, int src_w, int src_h,
                       size_t src_stride, size_t src_image_stride, uint8_t *d_dst,
                       int dst_w, int dst_h, size_t dst_stride, size_t dst_image_stride,
                       const float *d_hinv, int n_images, int flags, int dst_y0 = 0,
                       const float *host_hinv = nullptr);  // host_hinv (n_images <= 2): passed as kernel arguments

int stx_launch_match(stx_ctx *ctx, const float *d_desc1, const float *d_score1, int n1,
                     const float *d_desc2, const float *d_score2, int n2,
                     float dist_limit, int32_t *d_closest, int32_t *d_second,
                     float *d_min_dist, float *d_second_dist, uint8_t *d_status);

int stx_match_stats_impl(stx_ctx *ctx, long long *total_candidates, int *max_candidates, int *overflow_queries,
                         int *n_queries);

int stx_launch_unpack_features(stx_ctx *ctx, const void *d_rec, int n, size_t stride, size_t desc_off,
                               size_t score_off, float *d_desc, float *d_score);

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

int stx_launch_inlier_sum(stx_ctx *ctx, const float *d_errors, int n_hyp, int n, float threshold,
                          int32_t *d_inliers, float *d_err_sum);
int stx_launch_ransac_score(stx_ctx *ctx, const float *hyp, int n_hyp, const float K1[9], const float K2[9],
                            const float *d_p1, const float *d_p2, int n, const int32_t *exclude, float threshold,
                            int32_t *d_inliers, float *d_err_sum, float *d_errors, float *d_depths);
