// Drives the reference signatures of stereo_dropin.hpp on a GPU and compares
// with the CPU oracle (test infrastructure).  Built and run by tests/test_gpu_dropin.py.
#include <opencv2/core.hpp>
#include <Eigen/Dense>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <vector>

// struct Feature as the reference declares it (stereo/Features.h:46-84), minus stream operators
#define DESC_LENGTH 128
struct FeatureDescriptor { float vec[DESC_LENGTH]; };
struct Feature {
    int scale; cv::Point2f p; float score; float angle; FeatureDescriptor desc;
    float distFromBestMatch; float depth;
};

// What the reference's Stereography.h gives the adapter's FindFundamentalMatrixWithRANSAC
// (Stereography.h:11-39).  The two solver functions are TEST STAND-INS defined below (the real
// ones are Eigen BDCSVD code): they only have to be deterministic functions of their inputs.
#define FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD 70
#define MIN_NUM_INLIERS 20
#define FUNDAMENTAL_RANSAC_ITERATIONS 200
struct ImageDescriptor { Eigen::Matrix3f K; };
struct StereoPair { ImageDescriptor img1, img2; Eigen::Matrix3f F, E; };
bool FindFundamentalMatrix(const std::vector<std::pair<Feature, Feature> > &matches, Eigen::Matrix3f &F);
bool DecomposeEssentialMatrix(Eigen::Matrix3f &E, Eigen::Matrix3f &R1, Eigen::Matrix3f &R2, Eigen::Vector3f &t);

// the caller's statement that it does not write to the rectified Mats between RectifyImage and
// ComputeDepthImage (main.cpp:362-380 does not): the scan then reads the GPU-resident copies
#define STEREO_DROPIN_REUSE_RECTIFIED
#include "stereo_dropin.hpp"
#include "stereo_oracle.h"

// ---- two-view scene for the RANSAC test: X2 = R X1 + t ---------------------------
static const float SC_R[9] = {0.99939083f, 0.f, 0.0348995f, 0.f, 1.f, 0.f, -0.0348995f, 0.f, 0.99939083f};
static const float SC_T[3] = {-0.99503719f, 0.0497519f, 0.0870658f};  // unit length
static Eigen::Matrix3f sc_true_f;  // p2^T F p1 = 0, F(2,2) = 1

// stand-in for the eight-point solve: the true F, nudged by the first chosen pair so that
// hypotheses differ; every 7th draw "fails" (:214 takes that branch)
bool FindFundamentalMatrix(const std::vector<std::pair<Feature, Feature> > &matches, Eigen::Matrix3f &F)
{
    if (matches.size() < 8) return false;
    const int key = (int)matches[0].first.p.x;
    if (key % 7 == 0) return false;
    F = sc_true_f;
    const float bend = (float)(key % 13) * 2e-9f;
    F(0, 1) += bend;
    F(1, 0) -= bend;
    return true;
}

// stand-in for the SVD: scales E to unit singular values (:317-318, in place) and returns the
// scene's rotation and baseline; fails on every third call for a matrix whose E(0,0) bit pattern is odd
bool DecomposeEssentialMatrix(Eigen::Matrix3f &E, Eigen::Matrix3f &R1, Eigen::Matrix3f &R2, Eigen::Vector3f &t)
{
    float fro = 0;
    for (float v : E.m) fro += v * v;
    const float s = std::sqrt(fro / 2.f);
    for (float &v : E.m) v /= s;
    for (int i = 0; i < 9; ++i) R2.m[i] = SC_R[i];
    R1 = R2.transpose();
    for (int i = 0; i < 3; ++i) t(i) = SC_T[i];
    uint32_t bits;
    std::memcpy(&bits, &E.m[1], 4);
    return (bits & 7u) != 0;
}

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
    {   // FindFundamentalMatrixWithRANSAC: the batched call against the reference's sequential loop
        const int n = 600;
        StereoPair stereo;
        const float k1[9] = {2592.f, 0.f, 1440.f, 0.f, 2592.f, 994.f, 0.f, 0.f, 1.f};
        const float k2[9] = {2617.92f, 0.f, 1447.f, 0.f, 2617.92f, 990.f, 0.f, 0.f, 1.f};
        for (int i = 0; i < 9; ++i) { stereo.img1.K.m[i] = k1[i]; stereo.img2.K.m[i] = k2[i]; }
        // F = K2^-T [t]x R K1^-1, scaled to F(2,2) = 1 (double precision, then rounded)
        {
            const double tx[9] = {0, -SC_T[2], SC_T[1], SC_T[2], 0, -SC_T[0], -SC_T[1], SC_T[0], 0};
            double e[9], k1i[9] = {1 / 2592., 0, -1440. / 2592., 0, 1 / 2592., -994. / 2592., 0, 0, 1},
                   k2i[9] = {1 / 2617.92, 0, -1447. / 2617.92, 0, 1 / 2617.92, -990. / 2617.92, 0, 0, 1}, a[9], f[9];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { e[r * 3 + c] = 0; for (int k = 0; k < 3; ++k) e[r * 3 + c] += tx[r * 3 + k] * SC_R[k * 3 + c]; }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { a[r * 3 + c] = 0; for (int k = 0; k < 3; ++k) a[r * 3 + c] += k2i[k * 3 + r] * e[k * 3 + c]; }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { f[r * 3 + c] = 0; for (int k = 0; k < 3; ++k) f[r * 3 + c] += a[r * 3 + k] * k1i[k * 3 + c]; }
            for (int i = 0; i < 9; ++i) sc_true_f.m[i] = (float)(f[i] / f[8]);
        }
        std::vector<std::pair<Feature, Feature> > matches(n);
        for (int i = 0; i < n; ++i) {
            const double X[3] = {-2.5 + 5.0 * (rnd() % 10000) / 10000., -1.7 + 3.4 * (rnd() % 10000) / 10000., 4 + 8.0 * (rnd() % 10000) / 10000.};
            double Y[3];
            for (int r = 0; r < 3; ++r) Y[r] = SC_R[r * 3] * X[0] + SC_R[r * 3 + 1] * X[1] + SC_R[r * 3 + 2] * X[2] + 0.4 * SC_T[r];
            matches[i].first.p.x = (float)(k1[0] * X[0] / X[2] + k1[2]);
            matches[i].first.p.y = (float)(k1[4] * X[1] / X[2] + k1[5]);
            matches[i].second.p.x = (float)(k2[0] * Y[0] / Y[2] + k2[2]);
            matches[i].second.p.y = (float)(k2[4] * Y[1] / Y[2] + k2[5]);
            if (rnd() % 4 == 0) { matches[i].second.p.x = (float)(rnd() % 2880); matches[i].second.p.y = (float)(rnd() % 1988); }
        }
        Eigen::Matrix3f F;
        F(0, 0) = 17.f;  // the seed, :193
        const bool got = FindFundamentalMatrixWithRANSAC(matches, F, stereo);

        // the reference's loop, sequentially, with the oracle's ReprojectionError
        Eigen::Matrix3f best;
        int minError = FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD * MIN_NUM_INLIERS, solved = 0, failed_inner = 0;
        srand(17);
        float k1inv[9], k2inv[9];
        oracle_invert3x3(k1, k1inv);
        oracle_invert3x3(k2, k2inv);
        for (int it = 0; it < FUNDAMENTAL_RANSAC_ITERATIONS; ++it) {
            std::vector<std::pair<Feature, Feature> > pairs(matches), chosenEight;
            while (chosenEight.size() < 8) {
                const int randNum = rand() % pairs.size();
                chosenEight.push_back(pairs[randNum]);
                pairs.erase(pairs.begin() + randNum);
            }
            Eigen::Matrix3f fundamental;
            if (!FindFundamentalMatrix(chosenEight, fundamental)) continue;
            ++solved;
            Eigen::Matrix3f E = stereo.img2.K.transpose() * fundamental * stereo.img1.K, R, R2, Rin, Rother;
            Eigen::Vector3f t, tin;
            DecomposeEssentialMatrix(E, R2, R, t);
            Eigen::Matrix3f Ecopy = E;
            const bool inner = DecomposeEssentialMatrix(Ecopy, Rother, Rin, tin);
            failed_inner += !inner;
            float rec[ORACLE_HYP_FLOATS] = {0};
            for (int i = 0; i < 9; ++i) { rec[i] = E.m[i]; rec[9 + i] = R.m[i]; rec[21 + i] = Rin.m[i]; }
            for (int i = 0; i < 3; ++i) { rec[18 + i] = t(i); rec[30 + i] = tin(i); }
            rec[33] = inner ? 1.f : 0.f;
            int localInliers = 0;
            float avgError = 0;
            for (auto &m : pairs) {
                const float q1[2] = {m.first.p.x, m.first.p.y}, q2[2] = {m.second.p.x, m.second.p.y};
                const float e = oracle_reprojection_error(rec, k1inv, k2inv, k2, q1, q2, nullptr, nullptr);
                if (e < FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD) { localInliers++; avgError += e; }
            }
            if (localInliers > 0) avgError /= localInliers;
            if (localInliers > MIN_NUM_INLIERS && avgError < minError) { minError = avgError; best = fundamental; }
        }
        const bool want = minError < FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD * MIN_NUM_INLIERS;
        if (got != want || !want) ++fails;
        if (std::memcmp(F.m, best.m, sizeof best.m) != 0) ++fails;
        if (solved < 100 || solved == FUNDAMENTAL_RANSAC_ITERATIONS || failed_inner == 0) ++fails;  // every branch taken
        std::vector<std::pair<Feature, Feature> > few(matches.begin(), matches.begin() + 7);
        if (FindFundamentalMatrixWithRANSAC(few, F, stereo)) ++fails;
        std::printf("RANSAC fails so far: %d (solved %d, inner failures %d, minError %d)\n", fails, solved, failed_inner, minError);
    }
    // ---- main.cpp:362-380: RectifyImage x2 -> ComputeDepthImage on the resident copies -----
    {
        const int w = 333, h = 120;
        cv::Mat s0(h, w, 0), s1(h, w, 0);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
            s0.at<uint8_t>(y, x) = 1 + rnd() % 255;
            s1.at<uint8_t>(y, x) = 1 + rnd() % 255;
        }
        cv::Mat r0 = cv::Mat::zeros(cv::Size(w, h), CV_8U), r1 = cv::Mat::zeros(cv::Size(w, h), CV_8U);
        const float ha[9] = {0.9996f, -0.0061f, 7.5f, 0.0060f, 1.0001f, -2.25f, 0.9e-5f, -1.0e-6f, 1.0f};
        const float hb[9] = {1.0003f, 0.0042f, -5.5f, -0.0041f, 0.9998f, 3.75f, -0.7e-5f, 2.0e-6f, 1.0f};
        Eigen::Matrix3f HA, HB;
        for (int i = 0; i < 9; ++i) { HA.m[i] = ha[i]; HB.m[i] = hb[i]; }
        RectifyImage(s0, r0, HA);
        RectifyImage(s1, r1, HB);
        cv::Mat depth = ComputeDepthImage(r0, r1);   // both Mats are resident: nothing is uploaded
        float ia[9], ib[9];
        oracle_invert3x3(ha, ia);
        oracle_invert3x3(hb, ib);
        std::vector<uint8_t> w0((size_t)w * h, 0), w1((size_t)w * h, 0), wd((size_t)w * h, 0);
        oracle_rectify(s0.data, w, h, s0.step, w0.data(), w, h, w, ia);
        oracle_rectify(s1.data, w, h, s1.step, w1.data(), w, h, w, ib);
        oracle_disparity_ref(w0.data(), w1.data(), w, h, w, wd.data(), w);
        int nz = 0;
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
            if (r0.at<uint8_t>(y, x) != w0[(size_t)y * w + x] || r1.at<uint8_t>(y, x) != w1[(size_t)y * w + x]) ++fails;
            if (depth.at<uint8_t>(y, x) != wd[(size_t)y * w + x]) ++fails;
            nz += wd[(size_t)y * w + x] != 0;
        }
        if (nz < w * h / 2) ++fails;
        // a Mat that was NOT produced by the last two warps takes the upload path: same answer
        cv::Mat copy0(h, w, 0);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) copy0.at<uint8_t>(y, x) = r0.at<uint8_t>(y, x);
        cv::Mat depth2 = ComputeDepthImage(copy0, r1);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x)
            if (depth2.at<uint8_t>(y, x) != wd[(size_t)y * w + x]) ++fails;
        // swapped arguments use the slots the other way round
        cv::Mat depth3 = ComputeDepthImage(r1, r0);
        oracle_disparity_ref(w1.data(), w0.data(), w, h, w, wd.data(), w);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x)
            if (depth3.at<uint8_t>(y, x) != wd[(size_t)y * w + x]) ++fails;
        std::printf("R,R->D resident fails so far: %d\n", fails);
    }
    // ---- more than one GPU from one process: the multi-device handle ------------------------
    {
        const int nd = stx_device_count();
        const int w = 260, h = 40, n = 5;
        std::vector<uint8_t> a((size_t)n * w * h), b((size_t)n * w * h), got((size_t)n * w * h), want((size_t)w * h);
        for (size_t i = 0; i < a.size(); ++i) { a[i] = rnd() % 25 == 0 ? 0 : rnd() % 256; b[i] = rnd() % 25 == 0 ? 0 : rnd() % 256; }
        // n = 1, 2, 4: on a box with fewer GPUs the handle repeats device 0 (the library's test-only
        // switch), so the threaded shares, halo uploads and per-context staging pools run anyway
        setenv("STX_MULTI_ALLOW_REPEAT", "1", 1);
        for (int use = 1; use <= 4; use *= 2) {
            stx_multi *m = nullptr;
            int ids[4];
            for (int k = 0; k < use; ++k) ids[k] = k < nd ? k : 0;
            if (stx_multi_create(&m, ids, use) != STX_OK || stx_multi_size(m) != use) { ++fails; break; }
            if (stx_disparity_batch_u8(m, a.data(), b.data(), n, w, h, w, (size_t)w * h, 1, 0, STX_COST_SAD, got.data(), w,
                                       (size_t)w * h) != STX_OK) ++fails;
            for (int k = 0; k < n; ++k) {
                oracle_disparity_ref(a.data() + (size_t)k * w * h, b.data() + (size_t)k * w * h, w, h, w, want.data(), w);
                if (std::memcmp(want.data(), got.data() + (size_t)k * w * h, want.size()) != 0) ++fails;
            }
            std::fill(got.begin(), got.end(), 0);
            if (stx_disparity_rowband_u8(m, a.data(), b.data(), w, h, w, 1, 0, STX_COST_SAD, got.data(), w) != STX_OK) ++fails;
            oracle_disparity_ref(a.data(), b.data(), w, h, w, want.data(), w);
            if (std::memcmp(want.data(), got.data(), want.size()) != 0) ++fails;
            {   // one warp over the handle's devices
                const float hm[9] = {1.01f, 0.02f, -3.f, -0.015f, 0.99f, 2.5f, 1e-5f, -2e-5f, 1.f};
                float hinv[9];
                oracle_invert3x3(hm, hinv);
                std::vector<uint8_t> wr((size_t)w * h, 0), gr((size_t)w * h, 0);
                oracle_rectify(a.data(), w, h, w, wr.data(), w, h, w, hinv);
                if (stx_rectify_multi_u8(m, a.data(), w, h, w, gr.data(), w, h, w, hinv, STX_RECT_DST_IS_ZERO) != STX_OK) ++fails;
                if (std::memcmp(wr.data(), gr.data(), wr.size()) != 0) ++fails;
            }
            stx_multi_destroy(m);
        }
        std::printf("multi-device fails so far: %d (%d device(s) visible)\n", fails, nd);
    }
    std::printf(fails ? "DROPIN FAIL\n" : "DROPIN OK\n");
    return fails ? 1 : 0;
}
