// Drives the reference signatures of stereo_dropin.hpp on a GPU and compares
// with the CPU oracle (test infrastructure).  Built and run by tests/test_gpu_dropin.py.
#include <opencv2/core.hpp>
#include <Eigen/Dense>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>

// struct Feature as the reference declares it (stereo/Features.h:46-84), minus stream operators
#define DESC_LENGTH 128
struct FeatureDescriptor { float vec[DESC_LENGTH]; };
struct Feature {
    int scale; cv::Point2f p; float score; float angle; FeatureDescriptor desc;
    float distFromBestMatch; float depth;
};

#include "stereo_dropin.hpp"
#include "stereo_oracle.h"

static uint64_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(rng_state >> 33); }

int main()
{
    int fails = 0;
    // ---- D: pitched Mats, reference mode -------------------------------------
    {
        const int w = 301, h = 37;
        cv::Mat a(h, w, 320), b(h, w, 320);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
            a.at<uint8_t>(y, x) = rnd() % 20 == 0 ? 0 : rnd() % 256;
            b.at<uint8_t>(y, x) = rnd() % 20 == 0 ? 0 : rnd() % 256;
        }
        cv::Mat d = ComputeDepthImage(a, b);
        std::vector<uint8_t> want((size_t)w * h);
        oracle_disparity_ref(a.data, b.data, w, h, a.step, want.data(), w);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x)
            if (d.at<uint8_t>(y, x) != want[(size_t)y * w + x]) { ++fails; }
        cv::Mat c(h, w + 1, 0);
        std::stringstream cap; auto *old = std::cout.rdbuf(cap.rdbuf());
        cv::Mat e = ComputeDepthImage(a, c);
        std::cout.rdbuf(old);
        if (e.rows != 0 || e.cols != 0 || cap.str() != "Cannot compute depth!\n") ++fails;
        std::printf("D fails so far: %d\n", fails);
    }
    // ---- R --------------------------------------------------------------------
    {
        const int w = 200, h = 90;
        cv::Mat src(h, w, 0), dst = cv::Mat::zeros(cv::Size(w, h), CV_8U);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) src.at<uint8_t>(y, x) = 1 + rnd() % 255;
        Eigen::Matrix3f H;
        const float hm[9] = {0.9993f, -0.0087f, 12.5f, 0.0086f, 1.0002f, -3.25f, 1.1e-5f, -2.0e-6f, 1.0f};
        for (int i = 0; i < 9; ++i) H.m[i] = hm[i];
        RectifyImage(src, dst, H);
        float hinv[9];
        oracle_invert3x3(hm, hinv);
        std::vector<uint8_t> want((size_t)w * h, 0);
        oracle_rectify(src.data, w, h, src.step, want.data(), w, h, w, hinv);
        int hit = 0;
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
            if (dst.at<uint8_t>(y, x) != want[(size_t)y * w + x]) ++fails;
            hit += want[(size_t)y * w + x] != 0;
        }
        if (hit < w * h / 2) ++fails;
        std::printf("R fails so far: %d (hit %d)\n", fails, hit);
    }
    // ---- M --------------------------------------------------------------------
    {
        const int n1 = 70, n2 = 90;
        std::vector<Feature> l1(n1), l2(n2);
        auto fill = [](Feature &f, int i) {
            f.scale = 1; f.p.x = (float)i; f.p.y = 0; f.angle = 0; f.distFromBestMatch = 0; f.depth = 0;
            f.score = 700.f + (float)(rnd() % 4000);
            float nrm = 0;
            for (int k = 0; k < DESC_LENGTH; ++k) { f.desc.vec[k] = (rnd() % 2) ? (float)(rnd() % 1000) / 1000.f : 0.f; nrm += f.desc.vec[k] * f.desc.vec[k]; }
            nrm = std::sqrt(nrm) + 1e-6f;
            for (int k = 0; k < DESC_LENGTH; ++k) f.desc.vec[k] /= nrm;
        };
        for (int i = 0; i < n1; ++i) fill(l1[i], i);
        for (int j = 0; j < n2; ++j) fill(l2[j], j);
        for (int j = 5; j < n2; j += 2) {  // plant true matches
            l2[j].desc = l1[j % n1].desc;
            l2[j].desc.vec[3] += 0.01f;
        }
        l2[0].desc = l1[9].desc;  // nearest at j = 0 -> failure line
        std::stringstream cap; auto *old = std::cout.rdbuf(cap.rdbuf());
        auto pairs = MatchDescriptors(l1, l2, 1000.f);
        std::cout.rdbuf(old);
        std::vector<float> d1((size_t)n1 * 128), d2((size_t)n2 * 128), s1(n1), s2(n2);
        for (int i = 0; i < n1; ++i) { for (int k = 0; k < 128; ++k) d1[(size_t)i * 128 + k] = l1[i].desc.vec[k]; s1[i] = l1[i].score; }
        for (int j = 0; j < n2; ++j) { for (int k = 0; k < 128; ++k) d2[(size_t)j * 128 + k] = l2[j].desc.vec[k]; s2[j] = l2[j].score; }
        std::vector<int32_t> cl(n1), se(n1); std::vector<float> md(n1), sd(n1); std::vector<uint8_t> st(n1);
        const int acc = oracle_match(d1.data(), s1.data(), n1, d2.data(), s2.data(), n2, 1000.f, cl.data(), se.data(), md.data(), sd.data(), st.data());
        if ((int)pairs.size() != acc) ++fails;
        size_t p = 0;
        std::string wantlog;
        for (int i = 0; i < n1; ++i) {
            if (st[i] == ORACLE_MATCH_FAILED) wantlog += "Error: failed to match feature " + std::to_string(i) + "\n";
            if (st[i] != ORACLE_MATCH_ACCEPTED) continue;
            if (p >= pairs.size()) { ++fails; break; }
            if (pairs[p].first.p.x != (float)i || pairs[p].second.p.x != (float)cl[i]) ++fails;
            if (pairs[p].first.distFromBestMatch != md[i] || pairs[p].second.distFromBestMatch != md[i]) ++fails;
            ++p;
        }
        if (cap.str() != wantlog) ++fails;
        if (l1[0].distFromBestMatch != 0) ++fails;  // caller's lists untouched
        std::printf("M fails so far: %d (pairs %zu, oracle %d)\n", fails, pairs.size(), acc);
    }
    // ---- FAST ------------------------------------------------------------------
    {
        const int w = 150, h = 70;
        cv::Mat img(h, w, 160);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x)
            img.at<uint8_t>(y, x) = (uint8_t)((rnd() % 97 == 0) ? 200 : 40 + rnd() % 8);  // isolated bright dots
        std::vector<Feature> feats(1);
        feats[0].p.x = -1.f;
        if (!FindFASTFeatures(img, feats)) ++fails;
        std::vector<float> want(2 * (size_t)w * h);
        const int n = oracle_fast_features(img.data, w, h, img.step, 30, want.data(), w * h);
        if ((int)feats.size() != n + 1 || feats[0].p.x != -1.f) ++fails;
        for (int i = 0; i < n && i + 1 < (int)feats.size(); ++i)
            if (feats[i + 1].p.x != want[2 * i] || feats[i + 1].p.y != want[2 * i + 1]) ++fails;
        if (n < 10) ++fails;
        std::printf("FAST fails so far: %d (corners %d)\n", fails, n);
    }
    {   // ScoreAndClusterFeatures on gradient images
        const int w = 160, h = 90, n = 900;
        cv::Mat gx = cv::Mat::zeros(cv::Size(w, h), CV_8U), gy = cv::Mat::zeros(cv::Size(w, h), CV_8U);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                gx.at<uint8_t>(y, x) = (uint8_t)(rnd() % 3 ? rnd() % 200 : 0);
                gy.at<uint8_t>(y, x) = (uint8_t)(rnd() % 3 ? rnd() % 200 : 0);
            }
        std::vector<Feature> feats(n);
        std::vector<float> xy(2 * n);
        for (int i = 0; i < n; ++i) {
            feats[i].p.x = xy[2 * i] = (float)(rnd() % w);
            feats[i].p.y = xy[2 * i + 1] = (float)(rnd() % h);
        }
        std::vector<Feature> good = ScoreAndClusterFeaturesOnGradients(gx, gy, feats, 700.f, 2.f);
        float kernel[9];
        oracle_gaussian_kernel(3, 1.f, kernel);
        std::vector<float> oxy(2 * n), osc(n);
        std::vector<int32_t> oidx(n);
        const int m = oracle_score_and_cluster(gx.data, gy.data, w, h, gx.step, xy.data(), n, kernel, 700.f, 2.f,
                                               oxy.data(), osc.data(), oidx.data());
        if ((int)good.size() != m || m < 10) ++fails;
        for (int i = 0; i < m && i < (int)good.size(); ++i)
            if (good[i].p.x != oxy[2 * i] || good[i].p.y != oxy[2 * i + 1] || good[i].score != osc[i]) ++fails;
        std::printf("score+NMS fails so far: %d (survivors %d of %d)\n", fails, m, n);
        // descriptors of the survivors
        std::vector<FeatureDescriptor> descs;
        if (!CreateSIFTDescriptorsOnGradients(gx, gy, good, descs) || (int)descs.size() != m) ++fails;
        float k9[81], k4[16];
        oracle_gaussian_kernel(9, 1.f, k9);
        oracle_gaussian_kernel(4, 1.f, k4);
        std::vector<float> oang(m, 0.f), odesc(128 * (size_t)m);
        oracle_sift_descriptors(gx.data, gy.data, w, h, gx.step, oxy.data(), m, k9, k4, oang.data(), odesc.data());
        for (int i = 0; i < m && i < (int)good.size(); ++i) {
            if (good[i].angle != oang[i]) ++fails;
            if (std::memcmp(good[i].desc.vec, &odesc[128 * (size_t)i], 512) != 0) ++fails;
        }
        std::printf("descriptor fails so far: %d\n", fails);
    }
    std::printf(fails ? "DROPIN FAIL\n" : "DROPIN OK\n");
    return fails ? 1 : 0;
}
