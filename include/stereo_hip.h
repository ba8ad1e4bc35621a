/*
 * stereo_hip.h -- C ABI of libstereo_hip.so, the MI355X (gfx950) implementation
 * of the three data-parallel stages of dmckinnon/stereo:
 *
 *   M  MatchDescriptors     stereo/Features.h:195-196   (body Features.cpp:1020-1079)
 *   R  RectifyImage         stereo/Stereography.h:48-51 (body Stereography.cpp:596-631)
 *   D  ComputeDepthImage    stereo/Stereography.h:53-55 (body Stereography.cpp:649-691)
 *
 * The reference has no FFI: its boundary is those three C++ free functions,
 * called once each from main() (stereo/main.cpp:207, :363-369, :380).  The
 * drop-in adapter stereo_amd/adapter/stereo_dropin.hpp re-implements the three
 * signatures on top of this ABI.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary
 *   - return 0 on success, STX_ERR_* (< 0) for argument/usage errors, a
 *     positive hipError_t when a HIP call fails (stx_error_string decodes both)
 *   - one context per (thread, GPU); a context is bound to one device and one
 *     stream; calls on one context must not overlap
 *   - "host" entry points take host pointers, copy in, run, copy out and block;
 *     "_dev" entry points take device pointers, enqueue on the context's stream
 *     and return without synchronising (use stx_synchronize)
 *   - images are single-channel uint8, row-major, row pitch in BYTES
 */
#ifndef STEREO_HIP_H
#define STEREO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STX_VERSION 200 /* 0.2.0 */

#define STX_DESC_LEN 128 /* DESC_LENGTH, stereo/Features.h:46 */

/* error codes (negative); positive return values are hipError_t */
#define STX_OK 0
#define STX_ERR_INVALID_ARG (-1)
#define STX_ERR_UNSUPPORTED (-2)
#define STX_ERR_NO_DEVICE (-3)
#define STX_ERR_HOST_OOM (-4)
#define STX_ERR_TOO_LARGE (-5)

/* per-query outcome written to status[] by stx_match* */
#define STX_MATCH_FAILED 0   /* closest or secondClosest unset: the reference prints
                                "Error: failed to match feature <i>" (Features.cpp:1049-1054) */
#define STX_MATCH_SCORE 1    /* |score1[i]-score2[closest]| > dist_limit (Features.cpp:1057-1060) */
#define STX_MATCH_RATIO 2    /* ratio test failed (Features.cpp:1062-1067) */
#define STX_MATCH_ACCEPTED 3 /* the reference pushes the pair (Features.cpp:1069-1074) */

/* cost_mode of the disparity scan */
#define STX_COST_SAD 0
#define STX_COST_SSD 1

/* flags of stx_rectify_u8* */
#define STX_RECT_PRESERVE_DST 0 /* pixels that miss keep their prior value (reference behaviour) */
#define STX_RECT_DST_IS_ZERO 1  /* caller guarantees dst is all zero (as main.cpp:362,366 does):
                                   misses are written as 0 and dst is never read */

/* matching algorithm selector (stx_set_match_algo) */
#define STX_MATCH_AUTO 0  /* exact scan, 16 lanes per query, for small problems; bf16 MFMA prefilter + exact replay otherwise */
#define STX_MATCH_EXACT 1 /* VALU scan in reference operation order */
#define STX_MATCH_MFMA 2  /* fp32 MFMA prefilter + exact replay of the running-minimum records */
#define STX_MATCH_MFMA_BF16 3 /* same with the dot products as 3 bf16 MFMA products (hi/lo split) */
#define STX_MATCH_MFMA_BF16X1 4 /* same with one bf16 MFMA product and a wider (still rigorous) bound */
#define STX_MATCH_SMALL 5 /* exact scan with 16 lanes per query (what AUTO picks for small problems) */

/* disparity kernel selector (stx_set_disparity_algo) */
#define STX_DISP_AUTO 0       /* register sliding-sum kernel (win <= 31, max_disp <= 4096), else the LDS column-sum kernel */
#define STX_DISP_BRUTEFORCE 1 /* tap-by-tap kernel for every D1-ext call */
#define STX_DISP_COLSUM 2     /* LDS column-sum kernel (what AUTO uses beyond 4096 disparities or 65535 columns) for every window */

typedef struct stx_ctx stx_ctx;

/* ---- library / context ------------------------------------------------- */

int stx_version(void);
const char *stx_error_string(int code);

/* Number of HIP devices visible to this process (<0: error). */
int stx_device_count(void);

/* Create a context on HIP device `device`.  The launch stream defaults to the
 * NULL stream. */
int stx_create(stx_ctx **out, int device);
void stx_destroy(stx_ctx *ctx);

/* Use `hip_stream` (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream)
 * for every launch and copy of this context.  The stream is not owned. */
int stx_set_stream(stx_ctx *ctx, void *hip_stream);
int stx_synchronize(stx_ctx *ctx);

/* Choose the matching algorithm (STX_MATCH_*).  All algorithms give identical
 * outputs; this exists for testing and benchmarking. */
int stx_set_match_algo(stx_ctx *ctx, int algo);

/* Diagnostics of the context's last stx_match* call that went through the MFMA prefilter
 * ("the exact-replay overhead reported separately as candidates/query"): the sum and
 * the maximum over the queries of the candidates the exact replay looked at, and the number of
 * queries whose candidate lists overflowed and were re-scanned in full (Features.cpp:1036-1047
 * over all of list2).  n_queries = 0 when the last call did not use the prefilter.  Waits for
 * the context's stream.  No counterpart in the reference. */
int stx_match_stats(stx_ctx *ctx, long long *total_candidates, int *max_candidates, int *overflow_queries,
                    int *n_queries);

/* Device bytes the context currently holds in its grow-only workspace slots (staging, keys carried
 * between passes, matching work lists ...).  Diagnostic; no counterpart in the reference. */
size_t stx_workspace_bytes(const stx_ctx *ctx);

/* Page-locked host bytes the context currently holds as staging for the host-pointer entry points
 * (0 when every buffer it was given was already page-locked, or when the host had none to give and
 * the calls went on with plain copies).  Diagnostic; no counterpart in the reference. */
size_t stx_staging_bytes(const stx_ctx *ctx);

/* How the host-pointer entry points would treat the caller's buffer [p, p + bytes):
 *   0  pageable (or not inside ONE page-locked allocation the HIP runtime knows): staged through the
 *      library's pinned buffers;
 *   1  page-locked: read / written by the DMA engines as it lies;
 *   2  page-locked and visible to kernels on the context's device (hipHostMalloc, hipHostRegister with
 *      a mapping on that device): small results may be stored into it by the kernels themselves.
 * Negative: error.  Diagnostic; no counterpart in the reference (cv::Mat rows are pageable unless the
 * caller installs a pinned cv::MatAllocator, main.cpp:362-369). */
int stx_host_buffer_kind(stx_ctx *ctx, const void *p, size_t bytes);

/* Choose the D1-ext disparity kernel (STX_DISP_*).  Both give identical
 * outputs; this exists for testing and benchmarking. */
int stx_set_disparity_algo(stx_ctx *ctx, int algo);

/* How the host-pointer entry points move the caller's buffers.  on = 1 (default;
 * STX_HOST_PIPELINE=0 in the environment turns it off): rows are copied into pinned staging
 * memory by `threads` host threads including the caller (0: STX_HOST_THREADS, else a default
 * of at most 8) and a call is cut into `chunks` (0: STX_HOST_CHUNKS, else by size; at most 8)
 * whose upload, kernel and download overlap on three streams.  on = 0: plain hipMemcpyAsync
 * of the caller's pageable buffers around one kernel launch.  Results are identical. */
int stx_set_host_pipeline(stx_ctx *ctx, int on, int threads, int chunks);

/* ---- M: descriptor matching --------------------------------------------
 * Replaces MatchDescriptors (Features.h:195-196).  desc1/desc2 are [n][128]
 * fp32 row-major (Feature::desc, Features.h:47-50,58), score* are
 * Feature::score (Features.h:56).  dist_limit is distLimitBetweenMatches
 * (main.cpp:207 passes MAX_DIST_BETWEEN_MATCHES = 1000.f, Features.h:34).
 * Per query i of list1 (any output pointer may be NULL):
 *   closest[i], second[i]  final closest / secondClosest (Features.cpp:1033-1047), -1 = unset
 *   min_dist[i]            final minDist, -1.0f when n2 == 0
 *   second_dist[i]         distance to list2[second[i]], -1.0f when second[i] < 0
 *   status[i]              STX_MATCH_*
 * The adapter turns these into the reference's vector of (Feature, Feature)
 * pairs in ascending i and prints the reference's failure line. */
int stx_match(stx_ctx *ctx, const float *desc1, const float *score1, int n1,
              const float *desc2, const float *score2, int n2, float dist_limit,
              int32_t *closest, int32_t *second, float *min_dist,
              float *second_dist, uint8_t *status);

/* stx_match on the reference's own layout: feat1 / feat2 point at arrays of n1 / n2 feature
 * records of feature_stride bytes each -- std::vector<Feature>::data() with sizeof(Feature) = 540
 * -- holding the 128-float descriptor at desc_offset (offsetof(Feature, desc) = 20) and the score
 * at score_offset (offsetof(Feature, score) = 12); stride and offsets are multiples of 4.  The
 * records are uploaded as they lie and split into descriptor / score arrays on the GPU, so the
 * host never loops over the floats.  Outputs as stx_match.  Host pointers. */
int stx_match_features(stx_ctx *ctx, const void *feat1, int n1, const void *feat2, int n2,
                       size_t feature_stride, size_t desc_offset, size_t score_offset, float dist_limit,
                       int32_t *closest, int32_t *second, float *min_dist, float *second_dist,
                       uint8_t *status);

/* Same with every array resident in device memory; asynchronous. */
int stx_match_dev(stx_ctx *ctx, const float *d_desc1, const float *d_score1, int n1,
                  const float *d_desc2, const float *d_score2, int n2,
                  float dist_limit, int32_t *d_closest, int32_t *d_second,
                  float *d_min_dist, float *d_second_dist, uint8_t *d_status);

/* ---- R: rectification warp ---------------------------------------------
 * Host-side helper: H.inverse() as Eigen::Matrix3f computes it at
 * Stereography.cpp:617 (cofactors / determinant, fp32).  Row-major 3x3. */
int stx_invert_homography(const float H[9], float Hinv[9]);

/* Replaces the loop of RectifyImage (Stereography.cpp:612-630) for one image.
 * Hinv is the row-major INVERSE homography (rectified -> original), i.e.
 * stx_invert_homography(H) of the H the reference passes.  dst is in/out. */
int stx_rectify_u8(stx_ctx *ctx, const uint8_t *src, int src_w, int src_h,
                   size_t src_stride, uint8_t *dst, int dst_w, int dst_h,
                   size_t dst_stride, const float Hinv[9], int flags);

/* stx_rectify_u8 that also KEEPS the warped image on the GPU, in resident slot `slot` (0 or 1)
 * of the context, for stx_disparity_resident_u8: main.cpp:362-380 warps two images and scans
 * them at once, and what the scan reads is exactly what these two calls wrote.  A slot holds
 * its image until the next stx_rectify_u8_keep on it (any other call leaves it alone). */
int stx_rectify_u8_keep(stx_ctx *ctx, int slot, const uint8_t *src, int src_w, int src_h,
                        size_t src_stride, uint8_t *dst, int dst_w, int dst_h, size_t dst_stride,
                        const float Hinv[9], int flags);

/* Batched, device-resident form: n_images images, image k at
 * d_src + k*src_image_stride / d_dst + k*dst_image_stride, warped with
 * Hinv[9*k .. 9*k+8] (host array).  Asynchronous. */
int stx_rectify_u8_dev(stx_ctx *ctx, const uint8_t *d_src, int src_w, int src_h,
                       size_t src_stride, size_t src_image_stride, uint8_t *d_dst,
                       int dst_w, int dst_h, size_t dst_stride,
                       size_t dst_image_stride, const float *Hinv, int n_images,
                       int flags);

/* Row-band form for sharding one warp over several GPUs: fills
 * destination rows [dst_row0, dst_row0+dst_rows) of the image RectifyImage iterates over
 * (Stereography.cpp:612-613); d_dst points at row dst_row0.  The whole source must be
 * resident (every rank holds its own copy).  Asynchronous. */
int stx_rectify_rowband_u8_dev(stx_ctx *ctx, const uint8_t *d_src, int src_w, int src_h,
                               size_t src_stride, uint8_t *d_dst, int dst_w, int dst_row0,
                               int dst_rows, size_t dst_stride, const float Hinv[9], int flags);

/* ---- D: dense disparity scan -------------------------------------------
 * Replaces ComputeDepthImage (Stereography.h:53-55).  Reference mode is
 * win = 1, max_disp = 0, cost_mode = STX_COST_SAD: 1x1 absolute difference,
 * every column of the same row of img1 is a candidate, zero pixels are skipped,
 * first minimum wins, output |xBest - x| mod 256 (Stereography.cpp:664-688).
 *
 * win > 1 or max_disp > 0 select this library's generalisation (D1-ext,
 * DESIGN.md): odd win x win SAD/SSD window,
 * candidates x1 in [max(0, x-max_disp+1), x] scanned in ascending x1, initial
 * best 255*win^2 (65025*win^2 for SSD), strict <.  max_disp = 0 requires win = 1.
 * Both images are zero-extended outside their borders (a window tap that falls
 * outside an image reads 0).
 * The reference's size-mismatch branch (Stereography.cpp:655-659) is handled
 * by the adapter, which owns both Mats. */
int stx_disparity_u8(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int w,
                     int h, size_t stride, int win, int max_disp, int cost_mode,
                     uint8_t *out, size_t out_stride);

/* stx_disparity_u8 over the two resident images left by stx_rectify_u8_keep (img0 = slot
 * slot_img0, img1 = slot slot_img1; both must hold w x h images, else STX_ERR_INVALID_ARG):
 * no upload.  The CALLER asserts that its host copies of the two images have not changed since
 * the warp wrote them -- the library cannot see host writes. */
int stx_disparity_resident_u8(stx_ctx *ctx, int slot_img0, int slot_img1, int w, int h, int win,
                              int max_disp, int cost_mode, uint8_t *out, size_t out_stride);

/* uint16 output (value mod 65536) for max_disp > 256; out_stride in bytes. */
int stx_disparity_u16(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int w,
                      int h, size_t stride, int win, int max_disp, int cost_mode,
                      uint16_t *out, size_t out_stride);

/* Batched, device-resident form: pair k at d_img* + k*pair_stride, output k at
 * d_out + k*out_pair_stride.  out_bytes = 1 (uint8) or 2 (uint16); strides in
 * bytes.  Asynchronous. */
int stx_disparity_dev(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1,
                      int n_pairs, int w, int h, size_t stride, size_t pair_stride,
                      int win, int max_disp, int cost_mode, void *d_out,
                      size_t out_stride, size_t out_pair_stride, int out_bytes);

/* The same scan for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.  Arguments and
 * results are exactly stx_disparity_dev's.  The one scratch buffer the scan needs -- the
 * per-pixel keys carried between passes when more than 256 disparities are searched -- is a
 * static device array inside the library, so this call never calls hipMalloc / hipFree (nor any
 * allocating library routine).  Limit: a scan of more than 256 disparities may cover at most
 * STX_NOALLOC_MAX_PIXELS pixels (n_pairs * w * h; one 3840x2160 pair fits), else
 * STX_ERR_TOO_LARGE; scans of up to 256 disparities need no scratch and have no such limit.
 * Every call on one GPU shares that GPU's array (32 MiB, held until the process ends): calls
 * that overlap on one GPU -- from any context or stream -- race.  Asynchronous. */
#define STX_NOALLOC_MAX_PIXELS (1 << 23)
int stx_disparity_dev_noalloc(stx_ctx *ctx, const uint8_t *d_img0, const uint8_t *d_img1,
                              int n_pairs, int w, int h, size_t stride, size_t pair_stride,
                              int win, int max_disp, int cost_mode, void *d_out,
                              size_t out_stride, size_t out_pair_stride, int out_bytes);

/* Row-band form for sharding one large pair over several GPUs:
 * computes output rows [row0, row0+rows) only; img pointers and h describe the
 * rows this process holds, [band_row0, band_row0+h) of the full image, which
 * must include win/2 halo rows where the full image has them.  full_h is the
 * full image height.  d_out points at output row `row0`. */
int stx_disparity_rowband_dev(stx_ctx *ctx, const uint8_t *d_img0,
                              const uint8_t *d_img1, int w, int h, int band_row0,
                              int full_h, size_t stride, int row0, int rows, int win,
                              int max_disp, int cost_mode, void *d_out,
                              size_t out_stride, int out_bytes);

/* ---- FAST-12 corner detection (upstream producer of M's inputs) -----------
 * Replaces FindFASTFeatures (stereo/Features.cpp:407-471; decl Features.h:167-169):
 * every pixel at least 3 from the border whose 16-point ring has 12 contiguous points all
 * brighter than p + threshold or all darker than p - threshold (FAST_THRESHOLD = 30,
 * Features.h:13).  xy receives up to max_out (x, y) float pairs in the reference's raster
 * order (Feature::p); *count the total number found (may exceed max_out).  Host pointers. */
int stx_fast_detect_u8(stx_ctx *ctx, const uint8_t *img, int w, int h, size_t stride, int threshold,
                       float *xy, int max_out, int *count);

/* Device-resident form (d_count is a device int); asynchronous. */
int stx_fast_detect_u8_dev(stx_ctx *ctx, const uint8_t *d_img, int w, int h, size_t stride,
                           int threshold, float *d_xy, int max_out, int *d_count);

/* ---- Shi-Tomasi scoring + non-maximum suppression (between FAST and the descriptors) ----
 * Replaces the reference's own arithmetic in ScoreAndClusterFeatures
 * (stereo/Features.cpp:672-786; decl Features.h:184-188; calls Features.cpp:1184, main.cpp:258):
 * the scoring loop (:704-746: minimum eigenvalue of the kernel-weighted 3x3 structure matrix,
 * keep score > score_threshold), the O(n^2) suppression (:752-778: a kept feature is dropped
 * if a kept feature with a strictly larger score lies within cluster_dist on both axes) and
 * the sort by score, descending (:783; equal scores keep their input order -- std::sort
 * leaves that unspecified).  gx, gy are the CV_8U gradient images the reference makes with
 * OpenCV's GaussianBlur + Sobel (:684-690); they stay with the caller.  kernel3x3 is
 * CreateGaussianKernel's output (Features.cpp:20-34), row-major as gaussKernel.at<float>(r, c);
 * stx_gaussian_kernel computes it.  xy: n (x, y) float pairs (Feature::p).  Outputs hold up
 * to n entries: position, score and index in the input list of every surviving feature;
 * *count their number.  A feature whose 3x3 window leaves the image (the reference reads out
 * of bounds there) is dropped.  Host pointers. */
int stx_score_cluster(stx_ctx *ctx, const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                      const float *xy, int n, const float kernel3x3[9], float score_threshold,
                      float cluster_dist, float *xy_out, float *score_out, int32_t *index_out,
                      int *count);

/* Device-resident form (kernel3x3 is a host array; d_count a device int); asynchronous. */
int stx_score_cluster_dev(stx_ctx *ctx, const uint8_t *d_gx, const uint8_t *d_gy, int w, int h,
                          size_t stride, const float *d_xy, int n, const float kernel3x3[9],
                          float score_threshold, float cluster_dist, float *d_xy_out,
                          float *d_score_out, int32_t *d_index_out, int *d_count);

/* Replaces ClusterFeatures (stereo/Features.cpp:48-80; decl Features.h:176-178; call :375):
 * non-maximum suppression over features that already carry scores -- feature a is dropped if a
 * feature with a strictly larger score lies within `window` on both axes ((int)abs margins,
 * <= test, :61-65).  index_out receives the input positions of the survivors, in input order;
 * *count their number.  Host pointers. */
int stx_cluster_features(stx_ctx *ctx, const float *xy, const float *score, int n, int window,
                         int32_t *index_out, int *count);

/* Device-resident form (d_count a device int); asynchronous. */
int stx_cluster_features_dev(stx_ctx *ctx, const float *d_xy, const float *d_score, int n, int window,
                             int32_t *d_index_out, int *d_count);

/* ---- SIFT-like descriptors (producer of stx_match's inputs) --------------------
 * Replaces the per-feature body of CreateSIFTDescriptors (stereo/Features.cpp:844-934; decl
 * Features.h:190-193; call Features.cpp:1204) and ComputeFeatureOrientation (:945-995), on the
 * CV_8U gradient images of :830-837 (OpenCV's GaussianBlur + Sobel stay with the caller).
 * xy: n (x, y) float pairs; angle: n floats, in/out (Feature::angle: set to the dominant
 * orientation, kept when the 9x9 window holds no gradient); desc: n x 128 floats, out
 * (Feature::desc).  atan is taken from a table that the library fills with the host's atanf
 * once per context, so results follow the host libm as the reference's would.  Host pointers. */
int stx_sift_descriptors(stx_ctx *ctx, const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                         const float *xy, int n, float *angle, float *desc);

/* Device-resident form; asynchronous (the first call of a context uploads the tables). */
int stx_sift_descriptors_dev(stx_ctx *ctx, const uint8_t *d_gx, const uint8_t *d_gy, int w, int h,
                             size_t stride, const float *d_xy, int n, float *d_angle, float *d_desc);

/* CreateGaussianKernel (stereo/Features.cpp:20-34) on the host: out[r * size + c]. */
int stx_gaussian_kernel(int size, float sigma, float *out);

/* ---- RANSAC hypothesis scoring (between M and R) -------------------------------
 * Replaces the pair loop of FindFundamentalMatrixWithRANSAC (stereo/Stereography.cpp:226-238;
 * decl Stereography.h:29; call main.cpp:220) for a batch of n_hyp hypotheses over the same n
 * pairs: per pair ReprojectionError (:163-182) -> Triangulate (:383-443; decl Stereography.h:31)
 * -> LindstromOptimisation (:354-381), then localInliers and the running avgError (:232-236).
 * The draws, FindFundamentalMatrix and the singular value decompositions stay with the caller
 * (Eigen BDCSVD in the reference); each hypothesis is one record of STX_HYP_FLOATS floats:
 *   [0..8]   E, row-major, as DecomposeEssentialMatrix at :225 leaves it (it scales E in place, :318)
 *   [9..17]  R, [18..20] t of that call (its 4th and 3rd outputs as named at :225)
 *   [21..29] R, [30..32] t of the decomposition inside Triangulate (:398): decompose a COPY of
 *            the scaled E once more -- ReprojectionError takes E by value, so every pair sees the
 *            same matrices
 *   [33]     0 if that inner decomposition failed (Triangulate then returns false), else nonzero
 *   [34..35] unused
 * K1, K2: the camera matrices (stereo.img1.K, stereo.img2.K), row-major; their inverses and
 * R.inverse() are taken as Eigen's 3x3 inverse does (stx_invert_homography).  p1, p2: n (x, y)
 * float pairs (m.first.p, m.second.p).  exclude (may be NULL): n_hyp x 8 positions, in the
 * pair list, of each hypothesis's chosen eight, which the reference erases before the loop
 * (:205-211).  threshold: FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD (70, Stereography.h:11).
 * Outputs: inliers[h] = localInliers; err_sum[h] = avgError before the division at :240, summed
 * in pair order; errors (may be NULL; n_hyp x n): every pair's error, 1000 where Triangulate
 * fails, -1 for excluded pairs; depths (may be NULL; n_hyp x n x 2): Triangulate's depth0,
 * depth1 (NaN where it fails or the pair is excluded).  hyp, K1, K2, exclude are always host
 * arrays.  fp32 throughout, no contraction; three-term sums of fixed-size Eigen expressions as
 * a0 + (a1 + a2) (Eigen's unrolled reduction); parity with Eigen itself is unpinned.
 * n_hyp <= 65535; when errors is NULL the library keeps the n_hyp x n errors in a grow-only
 * workspace of its own (4 bytes each).  Host pointers. */
#define STX_HYP_FLOATS 36
int stx_ransac_score(stx_ctx *ctx, const float *hyp, int n_hyp, const float K1[9], const float K2[9],
                     const float *p1, const float *p2, int n, const int32_t *exclude, float threshold,
                     int32_t *inliers, float *err_sum, float *errors, float *depths);

/* Device-resident form: d_* are device pointers (d_errors, d_depths may be NULL); asynchronous
 * after the hypothesis records have been uploaded. */
int stx_ransac_score_dev(stx_ctx *ctx, const float *hyp, int n_hyp, const float K1[9], const float K2[9],
                         const float *d_p1, const float *d_p2, int n, const int32_t *exclude,
                         float threshold, int32_t *d_inliers, float *d_err_sum, float *d_errors,
                         float *d_depths);

/* The second half of that loop on its own (stereo/Stereography.cpp:232-236): for each of n_hyp rows
 * of n per-pair errors, inliers[h] = the number of errors e with 0 <= e < threshold and err_sum[h] =
 * their fp32 sum taken in pair order, avgError += reprojectionError, bit for bit (negative and NaN
 * entries -- the -1 of excluded pairs -- are skipped).  stx_ransac_score* end with this step; callers
 * that compute the errors themselves can use it alone.  Device pointers; asynchronous. */
int stx_inlier_sum_dev(stx_ctx *ctx, const float *d_errors, int n_hyp, int n, float threshold,
                       int32_t *d_inliers, float *d_err_sum);

/* ---- R+D fused on the device ---------------------------------------------
 * main.cpp:362-380 in one call: both images are rectified into zero-initialised
 * destinations of the source size (RectifyImage twice, Stereography.cpp:596-631) and the
 * disparity scan (ComputeDepthImage, :649-691, or D1-ext) runs on the rectified pair
 * without leaving the GPU.  Hinv0/Hinv1 are inverse homographies as for stx_rectify_u8.
 * rect0/rect1 (optional, may be NULL) receive the rectified images.  Host pointers. */
int stx_rectify_disparity_u8(stx_ctx *ctx, const uint8_t *img0, const uint8_t *img1, int w, int h,
                             size_t stride, const float Hinv0[9], const float Hinv1[9], int win,
                             int max_disp, int cost_mode, uint8_t *out, size_t out_stride,
                             uint8_t *rect0, uint8_t *rect1, size_t rect_stride);

/* ---- several GPUs in one process -------------------------------
 * The reference is one process with one thread (main.cpp:116-388); a drop-in that wants more
 * than one GPU therefore shards inside the library: one context, stream and host thread per
 * device, no collective -- every device uploads what it needs from the caller's host buffers
 * and writes its share of the result straight back.  (The one-process-per-GPU form with an
 * RCCL gather is stereo_amd/dist.py.)  Results are bit-identical to the single-context calls.
 * device_ids = NULL: devices 0 .. n_devices-1; n_devices = 0: every visible device. */
#define STX_MULTI_MAX_DEVICES 16
typedef struct stx_multi stx_multi;
int stx_multi_create(stx_multi **out, const int *device_ids, int n_devices);
void stx_multi_destroy(stx_multi *m);
int stx_multi_size(const stx_multi *m);
/* Device k's context (owned by m), e.g. for stx_set_match_algo. */
stx_ctx *stx_multi_context(stx_multi *m, int k);

/* ComputeDepthImage (Stereography.h:53-55) for a batch of pairs (BASELINE config 4): pair p at
 * img* + p*pair_stride, map p at out + p*out_pair_stride; pairs are dealt to the devices in
 * contiguous shares.  Parameters as stx_disparity_u8.  Host pointers. */
int stx_disparity_batch_u8(stx_multi *m, const uint8_t *img0, const uint8_t *img1, int n_pairs, int w,
                           int h, size_t stride, size_t pair_stride, int win, int max_disp,
                           int cost_mode, uint8_t *out, size_t out_stride, size_t out_pair_stride);

/* ComputeDepthImage for ONE large pair (BASELINE config 3), sharded by epipolar row bands
 * (the search is row-local, Stereography.cpp:664-686): device k computes a contiguous band of
 * rows and reads the win/2 halo rows above and below it from the caller's images (SURVEY 8e
 * row 2).  Parameters as stx_disparity_u8 / _u16.  Host pointers. */
int stx_disparity_rowband_u8(stx_multi *m, const uint8_t *img0, const uint8_t *img1, int w, int h,
                             size_t stride, int win, int max_disp, int cost_mode, uint8_t *out,
                             size_t out_stride);
int stx_disparity_rowband_u16(stx_multi *m, const uint8_t *img0, const uint8_t *img1, int w, int h,
                              size_t stride, int win, int max_disp, int cost_mode, uint16_t *out,
                              size_t out_stride);

/* MatchDescriptors (Features.h:195-196) with list1's queries sharded over the devices and list2
 * replicated.  Parameters and outputs as stx_match.  Host pointers. */
int stx_match_multi(stx_multi *m, const float *desc1, const float *score1, int n1, const float *desc2,
                    const float *score2, int n2, float dist_limit, int32_t *closest, int32_t *second,
                    float *min_dist, float *second_dist, uint8_t *status);

/* RectifyImage's loop (Stereography.cpp:612-630) for ONE image over the handle's devices
 * (SURVEY 8e, row R): every device holds a copy of the source and fills a contiguous band of
 * destination rows; the warp is output-pixel-local, so there is no halo and no exchange.
 * Arguments as stx_rectify_u8.  Host pointers; blocking. */
int stx_rectify_multi_u8(stx_multi *m, const uint8_t *src, int src_w, int src_h, size_t src_stride,
                         uint8_t *dst, int dst_w, int dst_h, size_t dst_stride, const float Hinv[9],
                         int flags);

#ifdef __cplusplus
}
#endif
#endif /* STEREO_HIP_H */
