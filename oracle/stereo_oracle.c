/*
 * stereo_oracle.c -- CPU restatement of dmckinnon/stereo's three hot-path
 * functions.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see stereo_oracle.h).
 *
 * Every function cites the reference lines it follows.  Nothing here is a copy
 * of reference text: the reference operates on cv::Mat / Eigen / std::vector,
 * this file on plain pointers, and the arithmetic is restated statement by
 * statement so that the rounding sequence is the same.
 *
 * Must be compiled with -ffp-contract=off (no FMA) and without -ffast-math.
 */
#include "stereo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ M --- */

/* Features.cpp:1007-1018 then :806-815.  aVec[i] -= bVec[i] is one rounded
 * fp32 subtraction; norm += v[i]*v[i] is a rounded product followed by a
 * rounded sum into a single float accumulator, i ascending; sqrt is the
 * correctly rounded float square root. */
float oracle_descriptor_distance(const float *a, const float *b)
{
    float norm = 0.0f; /* strict IEEE build: gcc may not reassociate this sum */
    for (int i = 0; i < ORACLE_DESC_LEN; ++i) {
        float diff = a[i] - b[i];
        float sq = diff * diff;
        norm = norm + sq;
    }
    return sqrtf(norm);
}

/* Features.cpp:1020-1079 */
int oracle_match(const float *desc1, const float *score1, int n1,
                 const float *desc2, const float *score2, int n2,
                 float dist_limit, int32_t *closest_out, int32_t *second_out,
                 float *min_dist_out, float *second_dist_out, uint8_t *status_out)
{
    int accepted = 0;
    for (int i = 0; i < n1; ++i) {
        const float *f = desc1 + (size_t)i * ORACLE_DESC_LEN;
        /* :1033-1035 */
        int closest = -1;
        int second = -1;
        float minDist = -1.0f;
        /* :1036-1047 */
        for (int j = 0; j < n2; ++j) {
            float dist = oracle_descriptor_distance(f, desc2 + (size_t)j * ORACLE_DESC_LEN);
            if (minDist == -1.0f || dist < minDist) {
                second = closest;
                closest = j;
                minDist = dist;
            }
        }
        float secondDist = -1.0f;
        if (second >= 0)
            secondDist = oracle_descriptor_distance(f, desc2 + (size_t)second * ORACLE_DESC_LEN);

        uint8_t status;
        if (closest == -1 || second == -1 || closest == second) {
            status = ORACLE_MATCH_FAILED; /* :1049-1054 */
        } else if (fabsf(score1[i] - score2[closest]) > dist_limit) {
            status = ORACLE_MATCH_SCORE; /* :1057-1060, float abs */
        } else {
            /* :1063-1067.  NN_RATIO is the double literal 0.8 (Features.h:33):
             * the float ratio is promoted to double for the comparison. */
            float ratio = minDist / secondDist;
            if ((double)ratio < 0.8) {
                status = ORACLE_MATCH_ACCEPTED;
                ++accepted;
            } else {
                status = ORACLE_MATCH_RATIO;
            }
        }
        if (closest_out) closest_out[i] = closest;
        if (second_out) second_out[i] = second;
        if (min_dist_out) min_dist_out[i] = minDist;
        if (second_dist_out) second_dist_out[i] = secondDist;
        if (status_out) status_out[i] = status;
    }
    return accepted;
}

/* ------------------------------------------------------------------ R --- */

/* 2x2 minor used by the cofactor expansion: m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1)
 * with i1=(i+1)%3, i2=(i+2)%3 (Eigen's cofactor_3x3<i,j>). */
static float cof3(const float m[9], int i, int j)
{
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    float p = m[i1 * 3 + j1] * m[i2 * 3 + j2];
    float q = m[i1 * 3 + j2] * m[i2 * 3 + j1];
    return p - q;
}

/* Eigen 3.3 compute_inverse<Matrix3f, 3>: first cofactor column, determinant
 * as its dot product with column 0 reduced as a0 + (a1 + a2), one reciprocal,
 * then every cofactor times that reciprocal.  Called at Stereography.cpp:617. */
void oracle_invert3x3(const float H[9], float Hinv[9])
{
    float c00 = cof3(H, 0, 0), c10 = cof3(H, 1, 0), c20 = cof3(H, 2, 0);
    float a0 = c00 * H[0], a1 = c10 * H[3], a2 = c20 * H[6];
    float det = a0 + (a1 + a2);
    float invdet = 1.0f / det;
    /* result(j,i) = cofactor<i,j> * invdet */
    Hinv[0] = c00 * invdet;
    Hinv[1] = c10 * invdet;
    Hinv[2] = c20 * invdet;
    Hinv[3] = cof3(H, 0, 1) * invdet;
    Hinv[4] = cof3(H, 1, 1) * invdet;
    Hinv[5] = cof3(H, 2, 1) * invdet;
    Hinv[6] = cof3(H, 0, 2) * invdet;
    Hinv[7] = cof3(H, 1, 2) * invdet;
    Hinv[8] = cof3(H, 2, 2) * invdet;
}

/* float -> uchar as the reference's x86-64 build performs it: truncate toward
 * zero to a 32-bit int, keep the low byte; NaN gives 0 (cvttss2si returns
 * 0x80000000).  Stated explicitly instead of relying on undefined behaviour. */
static uint8_t to_uchar(float v)
{
    if (isnan(v)) return 0;
    if (v >= 2147483648.0f || v < -2147483648.0f) return 0;
    return (uint8_t)((int32_t)v & 0xFF);
}

/* Stereography.cpp:577-594 */
uint8_t oracle_bilinear(const uint8_t *img, size_t stride, float x, float y)
{
    float x1 = floorf(x); /* :584 */
    float x2 = ceilf(x);  /* :585 */
    float y1 = floorf(y); /* :586 */
    float y2 = ceilf(y);  /* :587 */
    int ix1 = (int)x1, ix2 = (int)x2, iy1 = (int)y1, iy2 = (int)y2;
    float wx1 = (x2 - x) / (x2 - x1); /* 0/0 = NaN when x is an integer */
    float wx2 = (x - x1) / (x2 - x1);
    /* :589-590, products and sum in float, then truncation to uchar */
    uint8_t y1Val = to_uchar(wx1 * (float)img[(size_t)iy1 * stride + ix1] +
                             wx2 * (float)img[(size_t)iy1 * stride + ix2]);
    uint8_t y2Val = to_uchar(wx1 * (float)img[(size_t)iy2 * stride + ix1] +
                             wx2 * (float)img[(size_t)iy2 * stride + ix2]);
    float wy1 = (y2 - y) / (y2 - y1);
    float wy2 = (y - y1) / (y2 - y1);
    return to_uchar(wy1 * (float)y1Val + wy2 * (float)y2Val); /* :592 */
}

/* Stereography.cpp:612-630 */
void oracle_rectify(const uint8_t *src, int src_w, int src_h, size_t src_stride,
                    uint8_t *dst, int dst_w, int dst_h, size_t dst_stride,
                    const float Hinv[9])
{
    float xmax = (float)(src_w - 1); /* original.cols - 1 promoted, :621 */
    float ymax = (float)(src_h - 1); /* :623 */
    for (int y = 0; y < dst_h; ++y) {
        for (int x = 0; x < dst_w; ++x) {
            float px = (float)x, py = (float)y; /* Vector3f pixel(x, y, 1), :616 */
            /* :617 3x3 * 3x1: row dot product reduced as a0 + (a1 + a2) */
            float q[3];
            for (int i = 0; i < 3; ++i) {
                float a0 = Hinv[i * 3 + 0] * px;
                float a1 = Hinv[i * 3 + 1] * py;
                float a2 = Hinv[i * 3 + 2] * 1.0f;
                q[i] = a0 + (a1 + a2);
            }
            /* :618 true division by the saved third coordinate (Eigen >= 3.3) */
            float z = q[2];
            float tx = q[0] / z;
            float ty = q[1] / z;
            if (0.0f < tx && tx < xmax) {     /* :621 */
                if (0.0f < ty && ty < ymax) { /* :623 */
                    dst[(size_t)y * dst_stride + x] = oracle_bilinear(src, src_stride, tx, ty);
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ D --- */

/* Stereography.cpp:664-688 */
void oracle_disparity_ref(const uint8_t *img0, const uint8_t *img1, int w, int h,
                          size_t stride, uint8_t *out, size_t out_stride)
{
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = img0 + (size_t)y * stride;
        const uint8_t *r1 = img1 + (size_t)y * stride;
        uint8_t *o = out + (size_t)y * out_stride;
        for (int x = 0; x < w; ++x) {
            int pixel0 = r0[x];
            if (pixel0 == 0) { /* :670, depth stays Mat::zeros */
                o[x] = 0;
                continue;
            }
            int xBest = 0;      /* :671 */
            int distance = 255; /* :672 */
            for (int x1 = 0; x1 < w; ++x1) {
                int pixel1 = r1[x1];
                if (pixel1 == 0) continue; /* :676 */
                int ad = abs(pixel1 - pixel0);
                if (ad < distance) { /* :677 */
                    distance = ad;
                    xBest = x1;
                }
            }
            o[x] = (uint8_t)(abs(xBest - x) & 0xFF); /* :686 int -> uchar */
        }
    }
}

static void store_out(void *out, size_t out_stride_bytes, int out_bytes, int y, int x, int v)
{
    uint8_t *row = (uint8_t *)out + (size_t)y * out_stride_bytes;
    if (out_bytes == 1)
        row[x] = (uint8_t)(v & 0xFF);
    else
        ((uint16_t *)row)[x] = (uint16_t)(v & 0xFFFF);
}

static int ext_args_ok(int w, int h, int win, int max_disp, int cost_mode, int out_bytes)
{
    if (w < 0 || h < 0) return 0;
    if (win < 1 || (win & 1) == 0) return 0;
    if (max_disp < 0) return 0;
    if (cost_mode != 0 && cost_mode != 1) return 0;
    if (out_bytes != 1 && out_bytes != 2) return 0;
    return 1;
}

/* D1-ext, brute force.  Not in the reference. */
int oracle_disparity_ext(const uint8_t *img0, const uint8_t *img1, int w, int h,
                         size_t stride, int win, int max_disp, int cost_mode,
                         void *out, size_t out_stride_bytes, int out_bytes)
{
    if (!ext_args_ok(w, h, win, max_disp, cost_mode, out_bytes)) return -1;
    const int r = win / 2;
    const long long init = (cost_mode ? 65025LL : 255LL) * win * win;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int p0 = img0[(size_t)y * stride + x];
            if (p0 == 0) {
                store_out(out, out_stride_bytes, out_bytes, y, x, 0);
                continue;
            }
            int lo = 0, hi = w - 1;
            if (max_disp > 0) {
                lo = x - max_disp + 1;
                if (lo < 0) lo = 0;
                hi = x;
            }
            int xBest = 0;
            long long best = init;
            for (int x1 = lo; x1 <= hi; ++x1) {
                if (img1[(size_t)y * stride + x1] == 0) continue;
                long long cost = 0;
                for (int dy = -r; dy <= r; ++dy) {
                    int yy = y + dy;
                    if (yy < 0 || yy >= h) continue; /* both taps read 0 */
                    for (int dx = -r; dx <= r; ++dx) {
                        int xa = x + dx, xb = x1 + dx;
                        /* both images are zero-extended outside their borders */
                        int a = (xa >= 0 && xa < w) ? img0[(size_t)yy * stride + xa] : 0;
                        int b = (xb >= 0 && xb < w) ? img1[(size_t)yy * stride + xb] : 0;
                        int ad = abs(b - a);
                        cost += cost_mode ? (long long)ad * ad : ad;
                    }
                }
                if (cost < best) {
                    best = cost;
                    xBest = x1;
                }
            }
            store_out(out, out_stride_bytes, out_bytes, y, x, abs(xBest - x));
        }
    }
    return 0;
}

/* D1-ext with separable sliding sums: for each disparity d the per-pixel term
 * T_d(y,x') = f(|img1z(y,x'-d) - img0z(y,x')|) (z = zero-extended) is summed
 * down columns, then along rows.  Same result as the brute-force form; bounded
 * mode only (or win == 1). */
int oracle_disparity_ext_sliding(const uint8_t *img0, const uint8_t *img1, int w,
                                 int h, size_t stride, int win, int max_disp,
                                 int cost_mode, void *out, size_t out_stride_bytes,
                                 int out_bytes)
{
    if (!ext_args_ok(w, h, win, max_disp, cost_mode, out_bytes)) return -1;
    if (max_disp == 0) {
        if (win != 1) return -1;
        return oracle_disparity_ext(img0, img1, w, h, stride, win, max_disp, cost_mode,
                                    out, out_stride_bytes, out_bytes);
    }
    if (w == 0 || h == 0) return 0;
    const int r = win / 2;
    const uint32_t init = (uint32_t)((cost_mode ? 65025u : 255u) * (uint32_t)win * (uint32_t)win);
    /* extended column index e = x' + r + dmaxpad covers x' in [-r, w + r) */
    const int ew = w + 2 * r;
    uint32_t *best = (uint32_t *)malloc((size_t)w * h * sizeof(uint32_t));
    int32_t *bestx = (int32_t *)malloc((size_t)w * h * sizeof(int32_t));
    uint32_t *col = (uint32_t *)calloc((size_t)ew, sizeof(uint32_t));
    uint32_t *pre = (uint32_t *)malloc(((size_t)ew + 1) * sizeof(uint32_t));
    if (!best || !bestx || !col || !pre) {
        free(best); free(bestx); free(col); free(pre);
        return -1;
    }
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        best[i] = init;
        bestx[i] = 0;
    }
    int dmax = max_disp < w ? max_disp : w;
#define TERM(yy, xp)                                                                        \
    do {                                                                                    \
        int a_ = ((xp) >= 0 && (xp) < w) ? img0[(size_t)(yy) * stride + (xp)] : 0;          \
        int b_ = ((xp) - d >= 0 && (xp) - d < w) ? img1[(size_t)(yy) * stride + (xp) - d] : 0; \
        int ad_ = abs(b_ - a_);                                                             \
        term = cost_mode ? (uint32_t)(ad_ * ad_) : (uint32_t)ad_;                           \
    } while (0)
    /* ascending x1 == descending d; strict < keeps the earlier (lower x1) */
    for (int d = dmax - 1; d >= 0; --d) {
        uint32_t term;
        memset(col, 0, (size_t)ew * sizeof(uint32_t));
        for (int yy = 0; yy < r && yy < h; ++yy)
            for (int e = 0; e < ew; ++e) {
                TERM(yy, e - r);
                col[e] += term;
            }
        for (int y = 0; y < h; ++y) {
            int ya = y + r, ys = y - r - 1;
            if (ya < h)
                for (int e = 0; e < ew; ++e) {
                    TERM(ya, e - r);
                    col[e] += term;
                }
            if (ys >= 0)
                for (int e = 0; e < ew; ++e) {
                    TERM(ys, e - r);
                    col[e] -= term;
                }
            pre[0] = 0;
            for (int e = 0; e < ew; ++e) pre[e + 1] = pre[e] + col[e];
            const uint8_t *r0 = img0 + (size_t)y * stride;
            const uint8_t *r1 = img1 + (size_t)y * stride;
            for (int x = d; x < w; ++x) {
                if (r0[x] == 0 || r1[x - d] == 0) continue;
                /* window columns x-r..x+r -> e = x .. x+2r */
                uint32_t cost = pre[x + 2 * r + 1] - pre[x];
                size_t idx = (size_t)y * w + x;
                if (cost < best[idx]) {
                    best[idx] = cost;
                    bestx[idx] = x - d;
                }
            }
        }
    }
#undef TERM
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v = 0;
            if (img0[(size_t)y * stride + x] != 0) v = abs(bestx[(size_t)y * w + x] - x);
            store_out(out, out_stride_bytes, out_bytes, y, x, v);
        }
    free(best); free(bestx); free(col); free(pre);
    return 0;
}

/* ---------------------------------------------------------------- FAST --- */

/* Features.cpp:534-602.  For each point that is brighter than pb (or darker than p_b) the
 * reference walks forward around the ring while the same comparison holds, steps back one,
 * then walks backward until it fails or meets the forward end, and accepts a total length
 * of at least 12. */
int oracle_check_sequential12(const int points[16], int p_b, int pb)
{
    for (int i = 0; i < 16; ++i) {
        int bright;
        if (points[i] > pb)
            bright = 1;
        else if (points[i] < p_b)
            bright = 0;
        else
            continue;
#define SAME(v) (bright ? (v) > pb : (v) < p_b)
        int fLen = 0, fJ;
        for (fJ = i + 1; fJ != i; ++fJ) { /* :566-577 */
            if (fJ == 16) fJ = 0;
            if (fJ == i) break;
            if (SAME(points[fJ]))
                fLen++;
            else
                break;
        }
        fJ--; /* :578 */
        int bLen = 0;
        for (int bJ = i - 1; bJ != i && bJ != fJ; --bJ) { /* :581-592 */
            if (bJ == -1) bJ = 15;
            if (bJ == i || bJ == fJ) break;
            if (SAME(points[bJ]))
                bLen++;
            else
                break;
        }
#undef SAME
        if (fLen + bLen + 1 >= 12) return 1; /* :593-598 */
    }
    return 0;
}

static unsigned rot16(unsigned m, int k) { return ((m << k) | (m >> (16 - k))) & 0xFFFFu; }

static int mask_has_run12(unsigned m)
{
    unsigned x2 = m & rot16(m, 1);
    unsigned x4 = x2 & rot16(x2, 2);
    unsigned x8 = x4 & rot16(x4, 4);
    return (x8 & rot16(x8, 4)) != 0;
}

int oracle_ring_run12(const int points[16], int p_b, int pb)
{
    unsigned bright = 0, dark = 0;
    for (int i = 0; i < 16; ++i) {
        if (points[i] > pb) bright |= 1u << i;
        if (points[i] < p_b) dark |= 1u << i;
    }
    return mask_has_run12(bright) || mask_has_run12(dark);
}

/* Features.cpp:477-517 */
int oracle_three_of_four(int i1, int i5, int i9, int i13, int pb, int p_b)
{
    if ((p_b < i1 && i1 < pb) && (p_b < i9 && i9 < pb)) return 0;   /* :481-484 */
    if ((p_b < i5 && i5 < pb) && (p_b < i13 && i13 < pb)) return 0; /* :485-488 */
    int above = (i1 > pb) + (i5 > pb) + (i9 > pb) + (i13 > pb);     /* :494-497 */
    if (above >= 3) return 1;
    int below = (i1 < p_b) + (i5 < p_b) + (i9 < p_b) + (i13 < p_b); /* :504-507 */
    return below >= 3;
}

/* Features.cpp:407-471.  Ring order exactly as the reference lists it at :429-453
 * (its points 8 and 10 are (h+3, w-1) and (h+3, w+1), i.e. swapped with respect to a
 * clockwise walk; kept). */
int oracle_fast_features(const uint8_t *img, int w, int h, size_t stride, int threshold,
                         float *xy_out, int max_out)
{
    static const int dy[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
    static const int dx[16] = {0, 1, 2, 3, 3, 3, 2, -1, 0, 1, -2, -3, -3, -3, -2, -1};
    int count = 0;
    for (int y = 3; y < h - 3; ++y) {
        for (int x = 3; x < w - 3; ++x) {
            int p = img[(size_t)y * stride + x];
            int pb = p + threshold, p_b = p - threshold; /* :421-423 */
            int pts[16];
            for (int k = 0; k < 16; ++k) pts[k] = img[(size_t)(y + dy[k]) * stride + (x + dx[k])];
            if (!oracle_three_of_four(pts[0], pts[4], pts[8], pts[12], pb, p_b)) continue; /* :433 */
            if (!oracle_check_sequential12(pts, p_b, pb)) continue;                         /* :456 */
            if (count < max_out && xy_out) {
                xy_out[2 * count] = (float)x;     /* feature.p.x = w, :463 */
                xy_out[2 * count + 1] = (float)y; /* feature.p.y = h, :464 */
            }
            ++count;
        }
    }
    return count;
}

/* Exhaustive agreement of the literal scan with the run-length statement over all 3^16
 * bright / dark / neither ring patterns, and of the three-of-four pretest being a pure
 * filter (it never rejects a ring that has a run of 12).  Returns the number of
 * disagreements (0 expected).  Used by tests only. */
long oracle_fast_selfcheck(void)
{
    long bad = 0;
    int pts[16];
    const int p_b = 30, pb = 90; /* 0 = dark, 60 = neither, 91 = bright */
    static const int val[3] = {0, 60, 91};
    long total = 43046721; /* 3^16 */
    for (long code = 0; code < total; ++code) {
        long c = code;
        for (int i = 0; i < 16; ++i) {
            pts[i] = val[c % 3];
            c /= 3;
        }
        int a = oracle_check_sequential12(pts, p_b, pb);
        int b = oracle_ring_run12(pts, p_b, pb);
        if (a != b) ++bad;
        if (b && !oracle_three_of_four(pts[0], pts[4], pts[8], pts[12], pb, p_b)) ++bad;
    }
    return bad;
}

/* ---------------------------------------------------- scoring + NMS (N3) --- */

/* Features.cpp:20-34 */
void oracle_gaussian_kernel(int size, float sigma, float *out)
{
    for (int y = 0; y < size; ++y) {
        for (int x = 0; x < size; ++x) {
            float x_u = x - size / 2.f; /* :26 */
            float y_u = y - size / 2.f; /* :27 */
            /* :28; exp of a float argument is the float overload in the reference's C++ */
            float g = (1.f / (2.f * sigma * sigma * (float)3.14159265358979323846)) *
                      expf(-1 * (x_u * x_u + y_u * y_u) / (2 * sigma * sigma));
            out[x * size + y] = g; /* img.at<float>(x, y), :29 */
        }
    }
}

/* Features.cpp:704-740 */
float oracle_shi_tomasi_score(const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                              float px, float py, const float kernel3x3[9])
{
    int cy = (int)py, cx = (int)px; /* :718-719 */
    if (cx < 1 || cy < 1 || cx > w - 2 || cy > h - 2) return NAN;
    float m00 = 0.f, m01 = 0.f, m10 = 0.f, m11 = 0.f; /* Mat::zeros(2, 2, CV_32F), :708 */
    for (int n = -1; n <= 1; ++n) {         /* ST_WINDOW / 2 = 1, :714 */
        for (int m = -1; m <= 1; ++m) {     /* :716 */
            int i = n + cy, j = m + cx;
            float wgt = kernel3x3[(n + 1) * 3 + (m + 1)]; /* :720 */
            int vx = gx[(size_t)i * stride + j], vy = gy[(size_t)i * stride + j];
            m00 += wgt * (float)(vx * vx); /* :721 */
            m01 += wgt * (float)(vx * vy); /* :722 */
            m10 += wgt * (float)(vx * vy); /* :723 */
            m11 += wgt * (float)(vy * vy); /* :724 */
        }
    }
    float a = 1.f;                                /* :732 */
    float b = -1 * (m00 + m11);                   /* :733 */
    float c = m00 * m11 - m10 * m01;              /* :734 */
    float root = sqrtf(b * b - 4 * a * c);
    float eigen1 = (-b + root) / 2 * a;           /* :735 */
    float eigen2 = (-b - root) / 2 * a;           /* :736 */
    return eigen2 < eigen1 ? eigen2 : eigen1;     /* std::min(eigen1, eigen2), :738 */
}

/* Features.cpp:704-785 */
int oracle_score_and_cluster(const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                             const float *xy, int n, const float kernel3x3[9],
                             float score_threshold, float cluster_dist, float *xy_out,
                             float *score_out, int32_t *index_out)
{
    if (n <= 0) return 0;
    float *score = (float *)malloc(sizeof(float) * (size_t)n);
    int32_t *good = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *kept = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!score || !good || !kept) { free(score); free(good); free(kept); return -1; }
    int ngood = 0;
    for (int i = 0; i < n; ++i) {
        score[i] = oracle_shi_tomasi_score(gx, gy, w, h, stride, xy[2 * i], xy[2 * i + 1], kernel3x3);
        if (score[i] > score_threshold) good[ngood++] = i; /* :742 */
    }
    int nkept = 0;
    for (int a = 0; a < ngood; ++a) { /* :752-778 */
        int f = good[a];
        int is_max = 1;
        for (int b = 0; b < ngood; ++b) {
            if (b == a) continue;
            int f2 = good[b];
            int xmargin = (int)fabsf(xy[2 * f] - xy[2 * f2]);         /* :762 */
            int ymargin = (int)fabsf(xy[2 * f + 1] - xy[2 * f2 + 1]); /* :763 */
            if (xmargin <= cluster_dist && ymargin <= cluster_dist) { /* :764 */
                if (score[f] < score[f2]) {                           /* :766 */
                    is_max = 0;
                    break;
                }
            }
        }
        if (is_max) kept[nkept++] = f;
    }
    /* :783 sort by score descending; insertion by rank keeps equal scores in input order */
    for (int a = 0; a < nkept; ++a) {
        int f = kept[a];
        int rank = 0;
        for (int b = 0; b < nkept; ++b) {
            int f2 = kept[b];
            if (score[f2] > score[f] || (score[f2] == score[f] && b < a)) ++rank;
        }
        xy_out[2 * rank] = xy[2 * f];
        xy_out[2 * rank + 1] = xy[2 * f + 1];
        score_out[rank] = score[f];
        index_out[rank] = f;
    }
    free(score);
    free(good);
    free(kept);
    return nkept;
}

/* Features.cpp:48-80 */
int oracle_cluster_features(const float *xy, const float *score, int n, int window, int32_t *index_out)
{
    int kept = 0;
    for (int a = 0; a < n; ++a) {
        int is_max = 1;
        for (int b = 0; b < n; ++b) {
            if (b == a) continue;                                     /* :57 */
            int xmargin = (int)fabsf(xy[2 * a] - xy[2 * b]);         /* :61 */
            int ymargin = (int)fabsf(xy[2 * a + 1] - xy[2 * b + 1]); /* :62 */
            if (xmargin <= window && ymargin <= window) {            /* :63 */
                if (score[a] < score[b]) {                           /* :65 */
                    is_max = 0;
                    break;
                }
            }
        }
        if (is_max) index_out[kept++] = a;
    }
    return kept;
}

/* ------------------------------------------------ SIFT-like descriptors --- */

#define ORACLE_PI 3.14159f                                 /* Features.h:42 */
#define ORACLE_RAD2DEG(A) ((A) * 180.f / ORACLE_PI)        /* :43 */
#define ORACLE_DEG2RAD(A) ((A) * ORACLE_PI / 180.f)        /* :44 */

/* Features.cpp:971-978 and :881-888: gradient magnitude and angle in degrees at one pixel */
static void oracle_grad_polar(int vx, int vy, float *mag, float *angle)
{
    float gX = (float)vx, gY = (float)vy;
    *mag = sqrtf(gX * gX + gY * gY);
    if (gX != 0)
        *angle = ORACLE_RAD2DEG(atanf(gY / gX)); /* atan of a float: the float overload */
    else
        *angle = 90;
}

/* Features.cpp:945-995 */
void oracle_feature_orientation(const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                                float px, float py, const float k9[81], float *angle)
{
    float hist[36] = {0.0f}; /* ORIENTATION_HIST_BINS, :953 */
    for (int n = -4; n <= 4; ++n) {     /* ANGLE_WINDOW / 2 = 4, :955 */
        for (int m = -4; m <= 4; ++m) { /* :957 */
            int i = n + (int)py;        /* :960 */
            int j = m + (int)px;        /* :961 */
            if (i < 0 || i >= h) continue; /* :964 */
            if (j < 0 || j >= w) continue; /* :966 */
            float mag, ang;
            oracle_grad_polar(gx[(size_t)i * stride + j], gy[(size_t)i * stride + j], &mag, &ang);
            hist[(int)(ang / 10)] += mag * k9[(n + 4) * 9 + (m + 4)]; /* :979 */
        }
    }
    float dominant = 0; /* :985 */
    for (int i = 0; i < 36; ++i) {
        if (hist[i] > dominant) { /* :988 */
            dominant = hist[i];
            *angle = ORACLE_DEG2RAD(i * 10.f); /* :992 */
        }
    }
}

/* L2_norm<float>, Features.cpp:806-815, and NormaliseVector, :816-824 */
static void oracle_normalise128(float *v)
{
    float norm = 0;
    for (int i = 0; i < 128; ++i) norm += v[i] * v[i];
    float s = sqrtf(norm);
    for (int i = 0; i < 128; ++i) v[i] /= s;
}

/* Features.cpp:844-934 */
void oracle_sift_descriptors(const uint8_t *gx, const uint8_t *gy, int w, int h, size_t stride,
                             const float *xy, int n, const float k9[81], const float k4[16],
                             float *angle, float *desc)
{
    for (int f = 0; f < n; ++f) {
        float px = xy[2 * f], py = xy[2 * f + 1];
        oracle_feature_orientation(gx, gy, w, h, stride, px, py, k9, &angle[f]); /* :849 */
        float *vec = desc + (size_t)f * 128;
        int entry = 0; /* :855 */
        for (int j = 0; j < 16; j += 4) {     /* DESC_WINDOW, DESC_SUB_WINDOW, :860 */
            for (int k = 0; k < 16; k += 4) { /* :862 */
                float hist[8] = {0.0f};       /* DESC_BINS, :864 */
                for (int nn = j; nn < j + 4; ++nn) {
                    for (int m = k; m < k + 4; ++m) {
                        int imgX = (int)px - 8 + m;  /* :870 */
                        int imgY = (int)py - 8 + nn; /* :871 */
                        if (imgY < 0 || imgY >= h) continue; /* :874 */
                        if (imgX < 0 || imgX >= w) continue; /* :876 */
                        float mag, ang;
                        oracle_grad_polar(gx[(size_t)imgY * stride + imgX], gy[(size_t)imgY * stride + imgX],
                                          &mag, &ang);
                        ang -= angle[f];                         /* :890 degrees minus radians, as written */
                        if (ang > 360) ang = (float)((int)ang % 360); /* :891-892 */
                        if (ang < 0) ang = 0;                    /* :893-894 */
                        int bin = (int)ang / 45;                 /* DESC_BIN_SIZE, :895 */
                        if (bin > 7) bin = 7; /* ang == 360: the reference writes past hist[7] */
                        hist[bin] += mag * k4[(j / 4) * 4 + (k / 4)];
                    }
                }
                for (int index = 0; index < 8; ++index) vec[entry++] = hist[index]; /* :900-904 */
            }
        }
        oracle_normalise128(vec); /* :910 */
        for (int j = 0; j < 128; ++j)
            if (vec[j] > 0.2f) vec[j] = 0.2f; /* ILLUMINANCE_BOUND, :920-926 */
        oracle_normalise128(vec); /* :929 */
    }
}

/* ---- RANSAC hypothesis scoring --------------------------------- */

/* Three-term sums of fixed-size Eigen expressions (Matrix3f * Vector3f, dot(), norm()) go
 * through Eigen's unrolled reduction, which splits 3 terms as a0 + (a1 + a2) -- the same
 * order oracle_rectify uses for H^-1 * x.  Row r of a row-major 3x3 times a vector: */
static float row_dot3(const float *m, int r, const float v[3])
{
    float a0 = m[r * 3] * v[0], a1 = m[r * 3 + 1] * v[1], a2 = m[r * 3 + 2] * v[2];
    return a0 + (a1 + a2);
}

/* column c of a row-major 3x3 times a vector (a row vector times the matrix) */
static float col_dot3(const float *m, int c, const float v[3])
{
    float a0 = m[c] * v[0], a1 = m[3 + c] * v[1], a2 = m[6 + c] * v[2];
    return a0 + (a1 + a2);
}

/* The products through the dynamic-size selector S (MatrixXf, :356-363) take Eigen's run-time
 * paths instead; there the three terms are accumulated in index order. */
static float row_dot3_seq(const float *m, int r, const float v[3])
{
    float s = m[r * 3] * v[0];
    s = s + m[r * 3 + 1] * v[1];
    s = s + m[r * 3 + 2] * v[2];
    return s;
}

static float col_dot3_seq(const float *m, int c, const float v[3])
{
    float s = m[c] * v[0];
    s = s + m[3 + c] * v[1];
    s = s + m[6 + c] * v[2];
    return s;
}

void oracle_lindstrom(float x[3], float xp[3], const float E[9])
{
    /* Etilde = S E S^T is E's upper-left 2x2 (:360); S only selects rows */
    const float e00 = E[0], e01 = E[1], e10 = E[3], e11 = E[4];
    float n0 = row_dot3_seq(E, 0, xp), n1 = row_dot3_seq(E, 1, xp);  /* n = S E x' (:362) */
    float m0 = col_dot3_seq(E, 0, x), m1 = col_dot3_seq(E, 1, x);    /* n' = S E^T x (:363) */
    /* a = n^T Etilde n' (:366): the row vector n^T Etilde first */
    float r0 = n0 * e00 + n1 * e10, r1 = n0 * e01 + n1 * e11;
    float a = r0 * m0 + r1 * m1;
    float b = 0.5f * ((n0 * n0 + n1 * n1) + (m0 * m0 + m1 * m1)); /* :367 */
    /* c = x^T E x' (:368): the row vector x^T E first */
    float q[3] = {col_dot3(E, 0, x), col_dot3(E, 1, x), col_dot3(E, 2, x)};
    float c = q[0] * xp[0] + (q[1] * xp[1] + q[2] * xp[2]);
    float d = sqrtf(b * b - a * c);
    float lambda = c / (b + d);
    float xd0 = lambda * n0, xd1 = lambda * n1;    /* x_delta (:371) */
    float pd0 = lambda * m0, pd1 = lambda * m1;    /* xprime_delta (:372) */
    n0 = n0 - (e00 * pd0 + e01 * pd1);             /* :373 */
    n1 = n1 - (e10 * pd0 + e11 * pd1);
    m0 = m0 - (e00 * xd0 + e01 * xd1);             /* :375, Etilde again (not its transpose) */
    m1 = m1 - (e10 * xd0 + e11 * xd1);
    float sx = (xd0 * n0 + xd1 * n1) / (n0 * n0 + n1 * n1);   /* :377 */
    xd0 = sx * n0;
    xd1 = sx * n1;
    float sp = (pd0 * m0 + pd1 * m1) / (m0 * m0 + m1 * m1);   /* :378 */
    pd0 = sp * m0;
    pd1 = sp * m1;
    x[0] = x[0] - xd0;  /* :379, S^T pads the third row with 0 */
    x[1] = x[1] - xd1;
    x[2] = x[2] - 0.0f;
    xp[0] = xp[0] - pd0;
    xp[1] = xp[1] - pd1;
    xp[2] = xp[2] - 0.0f;
}

static float dot3(const float a[3], const float b[3])
{
    float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2];
    return p0 + (p1 + p2);
}

int oracle_triangulate(float *depth0, float *depth1, float x[3], float xp[3], const float E[9],
                       const float R[9], const float t[3], int decomp_ok)
{
    oracle_lindstrom(x, xp, E);
    if (!decomp_ok) return 0;  /* :398-401 */
    float nx[3] = {x[0] / x[2], x[1] / x[2], x[2] / x[2]};        /* :403 */
    float v[3] = {xp[0] / xp[2], xp[1] / xp[2], xp[2] / xp[2]};   /* :404, :406 */
    float u[3] = {row_dot3(R, 0, nx), row_dot3(R, 1, nx), row_dot3(R, 2, nx)};  /* :405 */
    float un = sqrtf(dot3(u, u)), vn = sqrtf(dot3(v, v));
    for (int i = 0; i < 3; ++i) {
        u[i] = u[i] / un;  /* :408 */
        v[i] = v[i] / vn;  /* :409 */
    }
    float a = dot3(u, t), b = dot3(u, u), c = dot3(u, v), d = dot3(v, t), e = dot3(v, v);
    float g = c * c - b * e;
    if ((double)fabsf(g) < 1e-9) return 0;  /* :418 */
    float d0, d1;
    if ((double)fabsf(c) < 1e-9) {  /* :422 */
        d1 = (a * c - b * d) / (c * c - b * e);
        d0 = (c * d1 - a) / b;
    } else {
        d0 = (a * e - c * d) / (c * c - b * e);
        d1 = (a + b * d0) / c;
    }
    /* :433-437 (the 3-D point) has no effect on the outputs */
    *depth0 = d0;
    *depth1 = d1;
    return 1;
}

float oracle_reprojection_error(const float hyp[ORACLE_HYP_FLOATS], const float K1inv[9],
                                const float K2inv[9], const float K2[9], const float p1[2],
                                const float p2[2], float *d0_out, float *d1_out)
{
    const float *E = hyp, *R = hyp + 9, *t = hyp + 18, *Rt = hyp + 21, *tt = hyp + 30;
    const float P1[3] = {p1[0], p1[1], 1.0f}, P2[3] = {p2[0], p2[1], 1.0f};  /* :228-229 */
    float point1[3] = {row_dot3(K1inv, 0, P1), row_dot3(K1inv, 1, P1), row_dot3(K1inv, 2, P1)};  /* :167 */
    float point2[3] = {row_dot3(K2inv, 0, P2), row_dot3(K2inv, 1, P2), row_dot3(K2inv, 2, P2)};  /* :168 */
    float d0 = NAN, d1 = NAN;
    int ok = oracle_triangulate(&d0, &d1, point1, point2, E, Rt, tt, hyp[33] != 0.0f);
    if (d0_out) *d0_out = ok ? d0 : NAN;
    if (d1_out) *d1_out = ok ? d1 : NAN;
    if (!ok) return 1000.0f;  /* :172 */
    float Rinv[9];
    oracle_invert3x3(R, Rinv);  /* R.inverse(), :175 */
    float s[3] = {point1[0] * d1, point1[1] * d1, point1[2] * d1};
    float tp[3];
    for (int i = 0; i < 3; ++i) tp[i] = row_dot3(Rinv, i, s) - row_dot3(Rinv, i, t);  /* :175 */
    float z = tp[2];
    for (int i = 0; i < 3; ++i) tp[i] = tp[i] / z;  /* :176 */
    float w[3] = {row_dot3(K2, 0, tp), row_dot3(K2, 1, tp), row_dot3(K2, 2, tp)};  /* :177 */
    float df[3] = {w[0] - P1[0], w[1] - P1[1], w[2] - P1[2]};
    return sqrtf(dot3(df, df));  /* :179 */
}

void oracle_ransac_score(const float *hyp, int H, const float K1[9], const float K2[9], const float *p1,
                         const float *p2, int n, const int32_t *exclude, float threshold,
                         int32_t *inliers, float *err_sum, float *errors, float *depths)
{
    float K1inv[9], K2inv[9];
    oracle_invert3x3(K1, K1inv);
    oracle_invert3x3(K2, K2inv);
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        const float *hy = hyp + (size_t)h * ORACLE_HYP_FLOATS;
        int count = 0;
        float sum = 0.0f;
        for (int i = 0; i < n; ++i) {
            int skip = 0;
            if (exclude)
                for (int k = 0; k < 8; ++k) skip |= exclude[(size_t)h * 8 + k] == i;
            float d0 = NAN, d1 = NAN, err = -1.0f;
            if (!skip) {
                err = oracle_reprojection_error(hy, K1inv, K2inv, K2, p1 + 2 * (size_t)i, p2 + 2 * (size_t)i, &d0, &d1);
                if (err < threshold) {  /* :232-236 */
                    count++;
                    sum += err;
                }
            }
            if (errors) errors[(size_t)h * n + i] = err;
            if (depths) {
                depths[((size_t)h * n + i) * 2] = d0;
                depths[((size_t)h * n + i) * 2 + 1] = d1;
            }
        }
        inliers[h] = count;
        err_sum[h] = sum;
    }
}
