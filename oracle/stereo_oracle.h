/*
 * stereo_oracle.h -- CPU restatement of the three hot-path functions of
 * dmckinnon/stereo.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load or call anything in oracle/.  The product (libstereo_hip.so and the
 * stereo_amd package) never links, imports or falls back to this code.
 *
 * PARITY UNPINNED: the reference ships no golden vectors or known-answer tests
 * for this path and cannot be compiled in this image
 * (it needs OpenCV 3.4.1, Eigen and <Windows.h>).  The pins are hand-derived
 * known-answer cases in tests/test_oracle_kat.py that were worked out by
 * reading the reference source, plus fixtures generated from this restatement
 * (tests/golden/, generator tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile); the
 * reference's working configuration is an unoptimised MSVC x64 build
 * (stereo/stereo.vcxproj:87-105), i.e. no FMA contraction and no reassociation.
 */
#ifndef STEREO_ORACLE_H
#define STEREO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_DESC_LEN 128 /* DESC_LENGTH, stereo/Features.h:46 */

/* per-query outcome codes written to status[] by oracle_match */
#define ORACLE_MATCH_FAILED 0   /* "Error: failed to match feature i", Features.cpp:1049-1054 */
#define ORACLE_MATCH_SCORE 1    /* rejected by the score gate, Features.cpp:1057-1060 */
#define ORACLE_MATCH_RATIO 2    /* rejected by the ratio test, Features.cpp:1062-1067 */
#define ORACLE_MATCH_ACCEPTED 3 /* pair pushed, Features.cpp:1069-1074 */

/* DistanceBetweenDescriptors + L2_norm, stereo/Features.cpp:806-815, 1007-1018 */
float oracle_descriptor_distance(const float *a, const float *b);

/* MatchDescriptors, stereo/Features.cpp:1020-1079.  desc*: [n][128] row-major.
 * Outputs are per query i of list1 (any may be NULL):
 *   closest[i], second[i]  final values of closest / secondClosest (-1 = unset)
 *   min_dist[i]            final minDist (-1.0f if list2 is empty)
 *   second_dist[i]         distance to list2[second[i]] (-1.0f if second < 0)
 *   status[i]              ORACLE_MATCH_* above
 * Returns the number of accepted pairs. */
int oracle_match(const float *desc1, const float *score1, int n1,
                 const float *desc2, const float *score2, int n2,
                 float dist_limit, int32_t *closest, int32_t *second,
                 float *min_dist, float *second_dist, uint8_t *status);

/* Eigen::Matrix3f::inverse() (cofactor / determinant form, fp32), as called at
 * stereo/Stereography.cpp:617.  Row-major 3x3 in and out.  Third-party
 * arithmetic restated from Eigen's published algorithm; version unpinned. */
void oracle_invert3x3(const float H[9], float Hinv[9]);

/* BilinearInterpolatePixel, stereo/Stereography.cpp:577-594 */
uint8_t oracle_bilinear(const uint8_t *img, size_t stride, float x, float y);

/* RectifyImage, stereo/Stereography.cpp:596-631, with H.inverse() hoisted and
 * passed in as Hinv (row-major).  dst pixels that fail the bounds tests keep
 * their prior contents. */
void oracle_rectify(const uint8_t *src, int src_w, int src_h, size_t src_stride,
                    uint8_t *dst, int dst_w, int dst_h, size_t dst_stride,
                    const float Hinv[9]);

/* ComputeDepthImage, stereo/Stereography.cpp:649-691 (reference mode: 1x1
 * window, whole-row search, both directions).  out is w*h, fully written
 * (zero where the reference leaves Mat::zeros untouched). */
void oracle_disparity_ref(const uint8_t *img0, const uint8_t *img1, int w, int h,
                          size_t stride, uint8_t *out, size_t out_stride);

/* D1-ext: NOT in the reference; this repository's
 * generalisation that collapses to oracle_disparity_ref for win=1,
 * max_disp=0.  See DESIGN.md "D1-ext semantics".
 *   win       odd window side >= 1; both images are zero-extended outside
 *             their borders (a tap that falls outside an image reads 0)
 *   max_disp  D > 0: candidates x1 in [max(0, x-D+1), x]; 0: whole row
 *   cost_mode 0 = SAD, 1 = SSD
 *   out_bytes 1 = uint8 output (value mod 256), 2 = uint16 (value mod 65536)
 * Returns 0, or -1 on invalid arguments. */
int oracle_disparity_ext(const uint8_t *img0, const uint8_t *img1, int w, int h,
                         size_t stride, int win, int max_disp, int cost_mode,
                         void *out, size_t out_stride_bytes, int out_bytes);

/* Same result as oracle_disparity_ext, computed with separable sliding sums
 * (O(w*h*D) instead of O(w*h*D*win^2)); used for the cpu_baseline timing and
 * cross-checked against the brute-force form in tests. */
int oracle_disparity_ext_sliding(const uint8_t *img0, const uint8_t *img1, int w,
                                 int h, size_t stride, int win, int max_disp,
                                 int cost_mode, void *out, size_t out_stride_bytes,
                                 int out_bytes);

/* ---- FAST-12 detection -------------------------------------
 * CheckForSequential12, stereo/Features.cpp:534-602: points[16] in the reference's ring
 * order, p_b/pb the lower/upper thresholds.  Pinned by the reference's own
 * TestSequential12 vectors (Features.cpp:605-641), see tests/test_oracle_kat.py. */
int oracle_check_sequential12(const int points[16], int p_b, int pb);

/* The same predicate stated as "a circular run of >= 12 points all > pb or all < p_b"
 * (what the GPU kernel evaluates with bit masks); tests prove the two agree on all 3^16
 * bright/dark/neither patterns. */
int oracle_ring_run12(const int points[16], int p_b, int pb);

/* ThreeOfFourValuesBrighterOrDarker, stereo/Features.cpp:477-517 */
int oracle_three_of_four(int i1, int i5, int i9, int i13, int pb, int p_b);

/* FindFASTFeatures, stereo/Features.cpp:407-471, with FAST_THRESHOLD (Features.h:13) as
 * a parameter.  Writes up to max_out (x, y) pairs in the reference's raster order
 * (feature.p.x = w, feature.p.y = h); returns the total number of corners found. */
int oracle_fast_features(const uint8_t *img, int w, int h, size_t stride, int threshold,
                         float *xy_out, int max_out);

/* ---- Shi-Tomasi scoring + non-maximum suppression -------------
 * CreateGaussianKernel, stereo/Features.cpp:20-34: size x size weights, out[x*size + y]
 * (the reference writes img.at<float>(x, y); the kernel is symmetric). */
void oracle_gaussian_kernel(int size, float sigma, float *out);

/* The scoring loop body of ScoreAndClusterFeatures, stereo/Features.cpp:704-740: minimum
 * eigenvalue of the kernel-weighted structure matrix over the 3x3 window (ST_WINDOW = 3,
 * Features.h:24) of the CV_8U gradient images around ((int)px, (int)py).  The gradient images
 * are inputs: the reference makes them with OpenCV's GaussianBlur and Sobel (:684-690),
 * third-party arithmetic that is not restated here.  A window that leaves the image (the
 * reference would read out of bounds) returns NaN. */
float oracle_shi_tomasi_score(const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                              float px, float py, const float kernel3x3[9]);

/* ClusterFeatures, stereo/Features.cpp:48-80: index_out receives the input positions of the
 * features that no strictly better feature has inside its window; returns their number. */
int oracle_cluster_features(const float *xy, const float *score, int n, int window, int32_t *index_out);

/* ScoreAndClusterFeatures, stereo/Features.cpp:704-785, on given gradient images: score,
 * keep score > score_threshold (:742), drop every kept feature that has a strictly better
 * kept feature within cluster_dist on both axes (:752-778), sort by score descending
 * (:783; std::sort leaves the order of equal scores unspecified: here ties keep the input
 * order).  xy is n (x, y) pairs; outputs hold up to n entries; index_out = position in the
 * input list.  Returns the number of surviving features. */
int oracle_score_and_cluster(const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                             const float *xy, int n, const float kernel3x3[9],
                             float score_threshold, float cluster_dist, float *xy_out,
                             float *score_out, int32_t *index_out);

/* ---- SIFT-like descriptors (third part) ------------------------
 * ComputeFeatureOrientation, stereo/Features.cpp:945-995: 36-bin orientation histogram over
 * the 9x9 window (ANGLE_WINDOW, Features.h:26) of the CV_8U gradient images, weighted by
 * magnitude and by k9 = CreateGaussianKernel(9, 1); *angle takes DEG2RAD(10 * dominant bin)
 * and keeps its prior value when every bin is 0 (the reference leaves it uninitialised). */
void oracle_feature_orientation(const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                                float px, float py, const float k9[81], float *angle);

/* The per-feature body of CreateSIFTDescriptors, stereo/Features.cpp:844-934, after the
 * OpenCV filters (:830-837): orientation, 4x4 cells x 8 bins over the 16x16 window with the
 * feature at (8, 8), k4 = CreateGaussianKernel(4, 1) cell weights, normalise, clamp at 0.2
 * (ILLUMINANCE_BOUND), renormalise.  angle: n values, in/out; desc: n x 128, out.  The
 * reference mixes units at :890 (angle in degrees minus f.angle in radians); kept. */
void oracle_sift_descriptors(const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                             const float *xy, int n, const float k9[81], const float k4[16],
                             float *angle, float *desc);

/* ---- RANSAC hypothesis scoring ---------------------------------
 * PARITY UNPINNED against Eigen: the reference evaluates these 3x3 / 2x2 expressions through
 * Eigen templates whose summation order is not visible in its own source; here three-term sums
 * of fixed-size expressions follow Eigen's unrolled reduction, a0 + (a1 + a2) (as oracle_rectify
 * does), the products through the dynamic-size selector S (:356-363) are accumulated in index
 * order, all in fp32 without contraction.  The singular value
 * decompositions (DecomposeEssentialMatrix, Stereography.cpp:301-353) are NOT restated: their
 * outputs are inputs here, one record per hypothesis:
 *   hyp[0..8]   E, row-major, as the call at :225 leaves it (scaled by its first singular value)
 *   hyp[9..17]  R and hyp[18..20] t of that call (its fourth and third arguments, :225)
 *   hyp[21..29] R and hyp[30..32] t of the decomposition inside Triangulate (:398) -- the same
 *               for every pair, since ReprojectionError takes E by value (:163)
 *   hyp[33]     0 if that inner decomposition failed (Triangulate returns false), else nonzero
 *   hyp[34..35] unused */
#define ORACLE_HYP_FLOATS 36

/* LindstromOptimisation, stereo/Stereography.cpp:354-381; x and xp are updated in place. */
void oracle_lindstrom(float x[3], float xp[3], const float E[9]);

/* Triangulate, stereo/Stereography.cpp:383-443, with the outcome of the decomposition at :398
 * passed in.  x, xp are updated in place (they are references in the reference); returns 0 where
 * the reference returns false. */
int oracle_triangulate(float *d0, float *d1, float x[3], float xp[3], const float E[9],
                       const float R[9], const float t[3], int decomp_ok);

/* ReprojectionError, stereo/Stereography.cpp:163-182, for one pair under one hypothesis.
 * K1inv, K2inv are oracle_invert3x3 of the camera matrices (K.inverse(), :167-168).  d0/d1
 * (optional) receive Triangulate's depths, NaN where it fails. */
float oracle_reprojection_error(const float hyp[ORACLE_HYP_FLOATS], const float K1inv[9],
                                const float K2inv[9], const float K2[9], const float p1[2],
                                const float p2[2], float *d0, float *d1);

/* The pair loop of FindFundamentalMatrixWithRANSAC, stereo/Stereography.cpp:226-238, for H
 * hypotheses over the same n pairs.  exclude (optional): H x 8 positions of each hypothesis's
 * chosen eight in the pair list, which the reference erases before the loop (:205-211).
 * inliers[h] = localInliers, err_sum[h] = avgError before the division at :240 (summed in pair
 * order).  errors (optional, H x n): every pair's error, -1 for the excluded ones; depths
 * (optional, H x n x 2): Triangulate's d0, d1. */
void oracle_ransac_score(const float *hyp, int H, const float K1[9], const float K2[9], const float *p1,
                         const float *p2, int n, const int32_t *exclude, float threshold,
                         int32_t *inliers, float *err_sum, float *errors, float *depths);

#ifdef __cplusplus
}
#endif
#endif
