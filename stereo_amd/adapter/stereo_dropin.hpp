// stereo_dropin.hpp -- the reference's three hot-path functions (and FindFASTFeatures,
// stereo/Features.cpp:407-471), re-implemented on top of libstereo_hip.so.  Header-only; include it INSTEAD of the bodies at
//   stereo/Features.cpp:1020-1079      (MatchDescriptors)
//   stereo/Stereography.cpp:596-631    (RectifyImage)
//   stereo/Stereography.cpp:649-691    (ComputeDepthImage)
//   stereo/Stereography.cpp:184-281    (FindFundamentalMatrixWithRANSAC; its pair loop runs on the GPU)
// The including translation unit must already see the reference's `Feature`
// (stereo/Features.h:52-84), cv::Mat and Eigen::Matrix3f.  Signatures, ownership and
// printed lines are the reference's.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "stereo_hip.h"

namespace stereo_hip_dropin {

// One context per thread, created on first use on HIP device STEREO_HIP_DEVICE (default 0).
inline stx_ctx *context()
{
    thread_local stx_ctx *ctx = nullptr;
    if (!ctx) {
        const char *dev = std::getenv("STEREO_HIP_DEVICE");
        const int rc = stx_create(&ctx, dev ? std::atoi(dev) : 0);
        if (rc != STX_OK) throw std::runtime_error(std::string("stx_create: ") + stx_error_string(rc));
    }
    return ctx;
}

// What RectifyImage left on the GPU: the two most recent warps stay resident (slots 0, 1), each
// remembered by the cv::Mat it was written to.  ComputeDepthImage may scan them without an
// upload -- see STEREO_DROPIN_REUSE_RECTIFIED below.
struct ResidentImage {
    const void *data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;
};
inline ResidentImage *resident()
{
    thread_local ResidentImage res[2];
    return res;
}
inline int &next_resident_slot()
{
    thread_local int slot = 0;
    return slot;
}

inline void check(int rc, const char *what)
{
    // the reference has no error path; a GPU failure must not pass silently
    if (rc != STX_OK) throw std::runtime_error(std::string(what) + ": " + stx_error_string(rc));
}

}  // namespace stereo_hip_dropin

// Features.h:195-196.  Lists are taken by value as in the reference (callee-owned copies).
inline std::vector<std::pair<Feature, Feature> > MatchDescriptors(
    std::vector<Feature> list1, std::vector<Feature> list2, float distLimitBetweenMatches)
{
    using namespace stereo_hip_dropin;
    const int n1 = (int)list1.size(), n2 = (int)list2.size();
    if (n1 == 0) return std::vector<std::pair<Feature, Feature> >();  // the outer loop at :1028 never runs
    // The feature records go up as they lie in the vectors (540-byte Feature, descriptor at
    // offset 20, score at 12); the GPU splits them into descriptor / score arrays.
    std::vector<int32_t> closest(n1), second(n1);
    std::vector<float> minDist(n1), secondDist(n1);
    std::vector<uint8_t> status(n1);
    check(stx_match_features(context(), list1.data(), n1, list2.data(), n2, sizeof(Feature),
                             (size_t)((const char *)&list1.data()->desc - (const char *)list1.data()),
                             (size_t)((const char *)&list1.data()->score - (const char *)list1.data()),
                             distLimitBetweenMatches, closest.data(), second.data(), minDist.data(),
                             secondDist.data(), status.data()),
          "stx_match_features");
    std::vector<std::pair<Feature, Feature> > matches;
    for (int i = 0; i < n1; ++i) {
        if (status[i] == STX_MATCH_FAILED) {
            std::cout << "Error: failed to match feature " << i << std::endl;  // Features.cpp:1052
            continue;
        }
        if (status[i] != STX_MATCH_ACCEPTED) continue;
        // Features.cpp:1070-1074: both copies carry the distance
        list1[i].distFromBestMatch = minDist[i];
        list2[closest[i]].distFromBestMatch = minDist[i];
        matches.push_back(std::make_pair(list1[i], list2[closest[i]]));
    }
    return matches;
}

// Features.h:167-169 (body Features.cpp:407-471).  Appends one default-constructed Feature per
// corner with p = (x, y), in the reference's raster order; always returns true.
inline bool FindFASTFeatures(cv::Mat img, std::vector<Feature> &features)
{
    using namespace stereo_hip_dropin;
    if (img.cols <= 0 || img.rows <= 0) return true;
    int count = 0;
    std::vector<float> xy(2 * 65536);
    check(stx_fast_detect_u8(context(), img.data, img.cols, img.rows, (size_t)img.step, 30 /* FAST_THRESHOLD */,
                             xy.data(), (int)(xy.size() / 2), &count),
          "stx_fast_detect_u8");
    if ((size_t)count > xy.size() / 2) {  // more corners than the first guess: ask again with room for all
        xy.resize(2 * (size_t)count);
        check(stx_fast_detect_u8(context(), img.data, img.cols, img.rows, (size_t)img.step, 30, xy.data(), count,
                                 &count),
              "stx_fast_detect_u8");
    }
    for (int i = 0; i < count; ++i) {
        Feature feature;
        feature.p.x = xy[2 * (size_t)i];
        feature.p.y = xy[2 * (size_t)i + 1];
        features.push_back(feature);
    }
    return true;
}

// The part of ScoreAndClusterFeatures (Features.h:184-188, Features.cpp:672-786) that is the
// reference's own arithmetic, on the CV_8U gradient images of :684-690.  The drop-in for the
// reference's signature is three OpenCV calls in front of it:
//   GaussianBlur(img, s, Size(ST_WINDOW, ST_WINDOW), 1, 1, BORDER_DEFAULT);
//   Sobel(s, gx, CV_8U, 1, 0, ST_WINDOW, 1, 0, BORDER_DEFAULT);  Sobel(s, gy, CV_8U, 0, 1, ...);
//   return ScoreAndClusterFeaturesOnGradients(gx, gy, features, scoreThreshold, distance);
inline std::vector<Feature> ScoreAndClusterFeaturesOnGradients(const cv::Mat &grad_x, const cv::Mat &grad_y,
                                                               std::vector<Feature> &features,
                                                               float scoreThreshold,
                                                               float distanceForWithinCluster)
{
    using namespace stereo_hip_dropin;
    std::vector<Feature> good;
    const int n = (int)features.size();
    if (n == 0 || grad_x.cols != grad_y.cols || grad_x.rows != grad_y.rows || grad_x.step != grad_y.step)
        return good;
    float kernel[9];
    check(stx_gaussian_kernel(3, 1.f, kernel), "stx_gaussian_kernel");  // CreateGaussianKernel, :695-696
    std::vector<float> xy(2 * (size_t)n), oxy(2 * (size_t)n), score((size_t)n);
    std::vector<int32_t> index((size_t)n);
    for (int i = 0; i < n; ++i) {
        xy[2 * (size_t)i] = features[i].p.x;
        xy[2 * (size_t)i + 1] = features[i].p.y;
    }
    int count = 0;
    check(stx_score_cluster(context(), grad_x.data, grad_y.data, grad_x.cols, grad_x.rows, (size_t)grad_x.step,
                            xy.data(), n, kernel, scoreThreshold, distanceForWithinCluster, oxy.data(),
                            score.data(), index.data(), &count),
          "stx_score_cluster");
    for (int i = 0; i < count; ++i) {
        features[index[i]].score = score[i];  // f is a reference into the caller's list, :706,:739
        good.push_back(features[index[i]]);
    }
    return good;
}

// The per-feature body of CreateSIFTDescriptors (Features.h:190-193, Features.cpp:827-937) on
// the CV_8U gradient images of :830-837; the drop-in keeps the reference's three OpenCV calls
// in front of it.  Sets angle and desc of every feature, appends the
// descriptors.
inline bool CreateSIFTDescriptorsOnGradients(const cv::Mat &grad_x, const cv::Mat &grad_y,
                                             std::vector<Feature> &features,
                                             std::vector<FeatureDescriptor> &descriptors)
{
    using namespace stereo_hip_dropin;
    const int n = (int)features.size();
    if (n == 0) return true;
    if (grad_x.cols != grad_y.cols || grad_x.rows != grad_y.rows || grad_x.step != grad_y.step) return false;
    std::vector<float> xy(2 * (size_t)n), angle((size_t)n), desc(128 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        xy[2 * (size_t)i] = features[i].p.x;
        xy[2 * (size_t)i + 1] = features[i].p.y;
        angle[i] = features[i].angle;
    }
    check(stx_sift_descriptors(context(), grad_x.data, grad_y.data, grad_x.cols, grad_x.rows, (size_t)grad_x.step,
                               xy.data(), n, angle.data(), desc.data()),
          "stx_sift_descriptors");
    for (int i = 0; i < n; ++i) {
        features[i].angle = angle[i];
        for (int k = 0; k < 128; ++k) features[i].desc.vec[k] = desc[128 * (size_t)i + k];
        descriptors.push_back(features[i].desc);  // :933
    }
    return true;
}

// Stereography.h:29 (body Stereography.cpp:184-281).  The including translation unit must see the
// reference's StereoPair, FindFundamentalMatrix (Stereography.h:28), DecomposeEssentialMatrix
// (Stereography.h:35-39) and the FUNDAMENTAL_* / MIN_NUM_INLIERS macros (Stereography.h:11-13):
// the draws, the eight-point solves and the Eigen decompositions stay the reference's own host
// code.  Neither depends on the scores, so all iterations' hypotheses are made first, the pair
// loop of every iteration (:226-238) runs in ONE stx_ransac_score call, and the selection rule
// (:239-246, :275-280) is applied in iteration order afterwards.
inline bool FindFundamentalMatrixWithRANSAC(const std::vector<std::pair<Feature, Feature> > &matches,
                                            Eigen::Matrix3f &F, StereoPair &stereo)
{
    using namespace stereo_hip_dropin;
    const int n = (int)matches.size();
    if (n < 8) return false;  // the reference's draw loop takes rand() % 0 here
    const int iterations = FUNDAMENTAL_RANSAC_ITERATIONS;
    srand(F(0, 0));  // :193, kept: the seed is whatever the caller's F holds

    std::vector<Eigen::Matrix3f> fundamentals;
    std::vector<float> hyp;
    std::vector<int32_t> exclude;
    std::vector<int> left;
    for (int it = 0; it < iterations; ++it) {
        // :200-211 with positions in place of copies of the pair list
        left.resize(n);
        for (int i = 0; i < n; ++i) left[i] = i;
        std::vector<std::pair<Feature, Feature> > chosenEight;
        int32_t chosen[8];
        while (chosenEight.size() < 8) {
            const int randNum = rand() % left.size();
            chosen[chosenEight.size()] = left[randNum];
            chosenEight.push_back(matches[left[randNum]]);
            left.erase(left.begin() + randNum);
        }
        Eigen::Matrix3f fundamental;
        if (!FindFundamentalMatrix(chosenEight, fundamental)) continue;  // :214
        Eigen::Matrix3f E = stereo.img2.K.transpose() * fundamental * stereo.img1.K;  // :220
        Eigen::Matrix3f R, R2;
        R.setZero();
        R2.setZero();
        Eigen::Vector3f t(0, 0, 0);
        DecomposeEssentialMatrix(E, R2, R, t);  // :225; scales E in place (:318)
        // every ReprojectionError call decomposes its by-value copy of E again inside Triangulate
        // (:163, :398): the same matrices for all pairs, so once here
        Eigen::Matrix3f Ecopy = E, Rin, Rother;
        Rin.setZero();
        Rother.setZero();
        Eigen::Vector3f tin(0, 0, 0);
        const bool inner = DecomposeEssentialMatrix(Ecopy, Rother, Rin, tin);
        float rec[STX_HYP_FLOATS] = {0};
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                rec[r * 3 + c] = E(r, c);
                rec[9 + r * 3 + c] = R(r, c);
                rec[21 + r * 3 + c] = Rin(r, c);
            }
            rec[18 + r] = t(r);
            rec[30 + r] = tin(r);
        }
        rec[33] = inner ? 1.f : 0.f;
        hyp.insert(hyp.end(), rec, rec + STX_HYP_FLOATS);
        exclude.insert(exclude.end(), chosen, chosen + 8);
        fundamentals.push_back(fundamental);
    }

    const int n_hyp = (int)fundamentals.size();
    std::vector<float> p1(2 * (size_t)n), p2(2 * (size_t)n), err_sum((size_t)n_hyp);
    std::vector<int32_t> inliers((size_t)n_hyp);
    float k1[9], k2[9];
    for (int i = 0; i < n; ++i) {
        p1[2 * (size_t)i] = matches[i].first.p.x;
        p1[2 * (size_t)i + 1] = matches[i].first.p.y;
        p2[2 * (size_t)i] = matches[i].second.p.x;
        p2[2 * (size_t)i + 1] = matches[i].second.p.y;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            k1[r * 3 + c] = stereo.img1.K(r, c);
            k2[r * 3 + c] = stereo.img2.K(r, c);
        }
    check(stx_ransac_score(context(), hyp.data(), n_hyp, k1, k2, p1.data(), p2.data(), n, exclude.data(),
                           (float)FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD, inliers.data(), err_sum.data(),
                           nullptr, nullptr),
          "stx_ransac_score");

    // :239-246 in iteration order; minError is an int in the reference (:191)
    Eigen::Matrix3f currentBestFundamentalMatrix;
    int minError = FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD * MIN_NUM_INLIERS;
    for (int h = 0; h < n_hyp; ++h) {
        const int localInliers = inliers[h];
        float avgError = err_sum[h];
        if (localInliers > 0) avgError /= localInliers;
        if (localInliers > MIN_NUM_INLIERS && avgError < minError) {
            minError = avgError;
            currentBestFundamentalMatrix = fundamentals[h];
        }
    }
    if (minError < FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD * MIN_NUM_INLIERS) {  // :275-280
        F = currentBestFundamentalMatrix;
        return true;
    }
    return false;
}

// Stereography.h:48-51.  H is the forward homography; `rectified` is in/out.
// The warped image also stays on the GPU (the two most recent ones do): see ComputeDepthImage.
inline void RectifyImage(const cv::Mat &original, cv::Mat &rectified, const Eigen::Matrix3f &H)
{
    using namespace stereo_hip_dropin;
    float h[9], hinv[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) h[r * 3 + c] = H(r, c);
    check(stx_invert_homography(h, hinv), "stx_invert_homography");  // H.inverse(), :617
    const int slot = next_resident_slot();
    next_resident_slot() ^= 1;
    ResidentImage &res = resident()[slot];
    res = ResidentImage();
    // a Mat that is warped again is no longer what the other slot remembers
    if (resident()[slot ^ 1].data == rectified.data) resident()[slot ^ 1] = ResidentImage();
    check(stx_rectify_u8_keep(context(), slot, original.data, original.cols, original.rows, (size_t)original.step,
                              rectified.data, rectified.cols, rectified.rows, (size_t)rectified.step, hinv,
                              STX_RECT_PRESERVE_DST),
          "stx_rectify_u8_keep");
    res.data = rectified.data;
    res.rows = rectified.rows;
    res.cols = rectified.cols;
    res.step = (size_t)rectified.step;
}

// Stereography.h:53-55.
// With STEREO_DROPIN_REUSE_RECTIFIED defined before this header is included, a call whose two
// arguments are the very Mats (same data pointer, size and pitch) that the two most recent
// RectifyImage calls of this thread wrote is served from the copies those calls left on the GPU
// -- main.cpp:362-380 exactly -- and uploads nothing.  The macro is the caller's statement that
// it does not write to the rectified Mats between the warp and the scan (the reference only
// displays them, main.cpp:371-376); the library cannot see host writes, so without the macro
// both images are uploaded as usual.
inline cv::Mat ComputeDepthImage(const cv::Mat &img0, const cv::Mat &img1)
{
    using namespace stereo_hip_dropin;
    if (img0.cols != img1.cols || img0.rows != img1.rows) {
        std::cout << "Cannot compute depth!" << std::endl;  // Stereography.cpp:657
        return cv::Mat(cv::Size(0, 0), CV_8U);
    }
    cv::Mat depth = cv::Mat::zeros(cv::Size(img0.cols, img0.rows), CV_8U);
#ifdef STEREO_DROPIN_REUSE_RECTIFIED
    {
        const ResidentImage *res = resident();
        int s0 = -1, s1 = -1;
        for (int k = 0; k < 2; ++k) {
            if (res[k].data && res[k].data == img0.data && res[k].rows == img0.rows && res[k].cols == img0.cols &&
                res[k].step == (size_t)img0.step)
                s0 = k;
            if (res[k].data && res[k].data == img1.data && res[k].rows == img1.rows && res[k].cols == img1.cols &&
                res[k].step == (size_t)img1.step)
                s1 = k;
        }
        if (s0 >= 0 && s1 >= 0 && s0 != s1 && img0.cols > 0 && img0.rows > 0) {
            check(stx_disparity_resident_u8(context(), s0, s1, img0.cols, img0.rows, 1, 0, STX_COST_SAD, depth.data,
                                            (size_t)depth.step),
                  "stx_disparity_resident_u8");
            return depth;
        }
    }
#endif
    if (img0.step == img1.step) {
        check(stx_disparity_u8(context(), img0.data, img1.data, img0.cols, img0.rows, (size_t)img0.step,
                               1, 0, STX_COST_SAD, depth.data, (size_t)depth.step),
              "stx_disparity_u8");
    } else {
        // different pitches: repack img1 to img0's pitch
        std::vector<uint8_t> tmp((size_t)img0.step * img0.rows);
        for (int y = 0; y < img1.rows; ++y)
            for (int x = 0; x < img1.cols; ++x) tmp[(size_t)y * img0.step + x] = img1.data[(size_t)y * img1.step + x];
        check(stx_disparity_u8(context(), img0.data, tmp.data(), img0.cols, img0.rows, (size_t)img0.step,
                               1, 0, STX_COST_SAD, depth.data, (size_t)depth.step),
              "stx_disparity_u8");
    }
    return depth;
}
