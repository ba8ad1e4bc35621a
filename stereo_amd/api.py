"""Host-side mirror of the reference's three hot-path functions.

The reference is C++ (OpenCV Mat / Eigen in and out); its drop-in adapter is
stereo_amd/adapter/stereo_dropin.hpp.  This module offers the same three
operations, with the same names, argument meaning and error behaviour, on numpy
arrays (and torch device tensors for the *_dev forms), so that tests and
benchmarks can drive libstereo_hip.so through its C ABI.

    MatchDescriptors(list1, list2, distLimitBetweenMatches)   Features.h:195-196
    RectifyImage(original, rectified, H)                      Stereography.h:48-51
    ComputeDepthImage(img0, img1)                             Stereography.h:53-55
"""
import copy
import ctypes as C
import os
import sys
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import check

# stereo/Features.h:33-34
NN_RATIO = 0.8
MAX_DIST_BETWEEN_MATCHES = 1000.0
DESC_LENGTH = 128
FAST_THRESHOLD = 30  # stereo/Features.h:13
ST_THRESH = 700.0    # stereo/Features.h:14
NMS_WINDOW = 2       # stereo/Features.h:16
ST_WINDOW = 3        # stereo/Features.h:24
FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD = 70  # stereo/Stereography.h:11
MIN_NUM_INLIERS = 20                           # stereo/Stereography.h:12
FUNDAMENTAL_RANSAC_ITERATIONS = 200            # stereo/Stereography.h:13
HYP_FLOATS = 36      # STX_HYP_FLOATS, include/stereo_hip.h

MATCH_FAILED, MATCH_SCORE, MATCH_RATIO, MATCH_ACCEPTED = 0, 1, 2, 3
COST_SAD, COST_SSD = 0, 1
RECT_PRESERVE_DST, RECT_DST_IS_ZERO = 0, 1
MATCH_AUTO, MATCH_EXACT, MATCH_MFMA, MATCH_MFMA_BF16, MATCH_MFMA_BF16X1, MATCH_SMALL = 0, 1, 2, 3, 4, 5
DISP_AUTO, DISP_BRUTEFORCE, DISP_COLSUM = 0, 1, 2
NOALLOC_MAX_PIXELS = 1 << 23  # STX_NOALLOC_MAX_PIXELS, include/stereo_hip.h


@dataclass
class Feature:
    """struct Feature, stereo/Features.h:52-84 (p is cv::Point2f as (x, y))."""
    scale: int = 0
    p: tuple = (0.0, 0.0)
    score: float = 0.0
    angle: float = 0.0
    desc: np.ndarray = field(default_factory=lambda: np.zeros(DESC_LENGTH, np.float32))
    distFromBestMatch: float = 0.0
    depth: float = 0.0


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class Context:
    """One stx_ctx: bound to one HIP device and one stream."""

    def __init__(self, device=None):
        self._lib = _lib.load_library()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        h = C.c_void_p()
        check(self._lib.stx_create(C.byref(h), int(device)), "stx_create")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.stx_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ---------------------------------------------------------
    def set_stream(self, hip_stream):
        """hip_stream: integer handle, e.g. torch.cuda.current_stream().cuda_stream."""
        check(self._lib.stx_set_stream(self._h, C.c_void_p(int(hip_stream))), "stx_set_stream")

    def synchronize(self):
        check(self._lib.stx_synchronize(self._h), "stx_synchronize")

    def set_match_algo(self, algo):
        check(self._lib.stx_set_match_algo(self._h, int(algo)), "stx_set_match_algo")

    def workspace_bytes(self):
        """Device bytes held in the context's grow-only workspace slots."""
        return int(self._lib.stx_workspace_bytes(self._h))

    def staging_bytes(self):
        """Page-locked host bytes held as staging for the host-pointer calls."""
        return int(self._lib.stx_staging_bytes(self._h))

    def host_buffer_kind(self, arr):
        """0 pageable / 1 page-locked (DMA direct) / 2 page-locked and kernel-visible, for a numpy array's memory."""
        a = np.asarray(arr)
        rc = int(self._lib.stx_host_buffer_kind(self._h, C.c_void_p(a.ctypes.data), a.nbytes))
        check(min(rc, 0), "stx_host_buffer_kind")
        return rc

    def match_stats(self):
        """Candidates the exact replay looked at in the last prefilter call of this context:
        {"queries", "candidates_total", "candidates_per_query", "candidates_max", "overflow_queries"}
        (queries = 0 when the last matching call did not use the prefilter).  Waits for the stream."""
        tot, mx, ov, nq = C.c_longlong(), C.c_int(), C.c_int(), C.c_int()
        check(self._lib.stx_match_stats(self._h, C.byref(tot), C.byref(mx), C.byref(ov), C.byref(nq)),
              "stx_match_stats")
        return {"queries": nq.value, "candidates_total": tot.value,
                "candidates_per_query": tot.value / nq.value if nq.value else 0.0,
                "candidates_max": mx.value, "overflow_queries": ov.value}

    def set_disparity_algo(self, algo):
        check(self._lib.stx_set_disparity_algo(self._h, int(algo)), "stx_set_disparity_algo")

    def set_host_pipeline(self, on=True, threads=0, chunks=0):
        """How host-pointer calls move their buffers: pinned staging + chunked overlap (on), or
        plain hipMemcpyAsync of the caller's pageable arrays (off).  Same results either way."""
        check(self._lib.stx_set_host_pipeline(self._h, int(bool(on)), int(threads), int(chunks)),
              "stx_set_host_pipeline")

    # -- M ------------------------------------------------------------------
    def match(self, desc1, score1, desc2, score2, dist_limit=MAX_DIST_BETWEEN_MATCHES):
        """Arrays in, per-query arrays out: (closest, second, min_dist, second_dist, status)."""
        desc1 = np.ascontiguousarray(desc1, np.float32).reshape(-1, DESC_LENGTH)
        desc2 = np.ascontiguousarray(desc2, np.float32).reshape(-1, DESC_LENGTH)
        score1 = np.ascontiguousarray(score1, np.float32).reshape(-1)
        score2 = np.ascontiguousarray(score2, np.float32).reshape(-1)
        n1, n2 = desc1.shape[0], desc2.shape[0]
        if score1.shape[0] != n1 or score2.shape[0] != n2:
            raise ValueError("score arrays must have one entry per descriptor")
        closest = np.full(n1, -1, np.int32)
        second = np.full(n1, -1, np.int32)
        min_dist = np.full(n1, -1.0, np.float32)
        second_dist = np.full(n1, -1.0, np.float32)
        status = np.zeros(n1, np.uint8)
        check(self._lib.stx_match(self._h, _ptr(desc1), _ptr(score1), n1, _ptr(desc2),
                                  _ptr(score2), n2, C.c_float(dist_limit), _ptr(closest),
                                  _ptr(second), _ptr(min_dist), _ptr(second_dist),
                                  _ptr(status)), "stx_match")
        return closest, second, min_dist, second_dist, status

    def match_features(self, rec1, rec2, desc_offset, score_offset, dist_limit=MAX_DIST_BETWEEN_MATCHES):
        """match() on arrays of feature RECORDS (the reference's std::vector<Feature> layout):
        rec1 / rec2 are [n, stride_bytes] uint8 arrays (or structured arrays) holding the 128-float
        descriptor at desc_offset and the score at score_offset of every record."""
        def rows(rec):
            a = np.ascontiguousarray(rec)
            return a.view(np.uint8).reshape(len(a), -1) if len(a) else None   # (-1 cannot be inferred for 0 rows)

        r1, r2 = rows(rec1), rows(rec2)
        if r1 is None and r2 is None:
            stride0 = np.asarray(rec1).dtype.itemsize * int(np.prod(np.asarray(rec1).shape[1:], dtype=np.int64))
            r1 = r2 = np.zeros((0, max(stride0, 1)), np.uint8)
        elif r1 is None:
            r1 = np.zeros((0, r2.shape[1]), np.uint8)
        elif r2 is None:
            r2 = np.zeros((0, r1.shape[1]), np.uint8)
        if r1.shape[1] != r2.shape[1]:
            raise ValueError("both lists must use the same record size")
        n1, n2, stride = r1.shape[0], r2.shape[0], r1.shape[1]
        closest = np.full(n1, -1, np.int32)
        second = np.full(n1, -1, np.int32)
        min_dist = np.full(n1, -1.0, np.float32)
        second_dist = np.full(n1, -1.0, np.float32)
        status = np.zeros(n1, np.uint8)
        check(self._lib.stx_match_features(self._h, _ptr(r1) if n1 else None, n1, _ptr(r2) if n2 else None, n2, stride,
                                           int(desc_offset), int(score_offset), C.c_float(dist_limit), _ptr(closest),
                                           _ptr(second), _ptr(min_dist), _ptr(second_dist), _ptr(status)),
              "stx_match_features")
        return closest, second, min_dist, second_dist, status

    def match_dev(self, desc1, score1, desc2, score2, dist_limit, closest, second, min_dist,
                  second_dist, status):
        """torch device tensors in and out; asynchronous on the context's stream."""
        n1, n2 = int(desc1.shape[0]), int(desc2.shape[0])
        check(self._lib.stx_match_dev(self._h, desc1.data_ptr(), score1.data_ptr(), n1,
                                      desc2.data_ptr(), score2.data_ptr(), n2,
                                      C.c_float(dist_limit), closest.data_ptr(),
                                      second.data_ptr(), min_dist.data_ptr(),
                                      second_dist.data_ptr(), status.data_ptr()), "stx_match_dev")

    # -- R ------------------------------------------------------------------
    def invert_homography(self, H):
        H = np.ascontiguousarray(H, np.float32).reshape(9)
        out = np.empty(9, np.float32)
        fp = C.POINTER(C.c_float)
        check(self._lib.stx_invert_homography(H.ctypes.data_as(fp), out.ctypes.data_as(fp)),
              "stx_invert_homography")
        return out.reshape(3, 3)

    def rectify(self, original, rectified, Hinv, flags=RECT_PRESERVE_DST):
        """original: HxW uint8; rectified: H'xW' uint8, modified in place."""
        _check_image(original, "original")
        _check_image(rectified, "rectified", writable=True)
        Hinv = np.ascontiguousarray(Hinv, np.float32).reshape(9)
        check(self._lib.stx_rectify_u8(
            self._h, _ptr(original), original.shape[1], original.shape[0],
            original.strides[0] if original.shape[0] > 1 else max(original.shape[1], 1),
            _ptr(rectified), rectified.shape[1], rectified.shape[0],
            rectified.strides[0] if rectified.shape[0] > 1 else max(rectified.shape[1], 1),
            Hinv.ctypes.data_as(C.POINTER(C.c_float)), int(flags)), "stx_rectify_u8")
        return rectified

    def rectify_keep(self, slot, original, rectified, Hinv, flags=RECT_PRESERVE_DST):
        """rectify() that also keeps the warped image on the GPU in resident slot 0 or 1."""
        _check_image(original, "original")
        _check_image(rectified, "rectified", writable=True)
        Hinv = np.ascontiguousarray(Hinv, np.float32).reshape(9)
        check(self._lib.stx_rectify_u8_keep(
            self._h, int(slot), _ptr(original), original.shape[1], original.shape[0],
            original.strides[0] if original.shape[0] > 1 else max(original.shape[1], 1),
            _ptr(rectified), rectified.shape[1], rectified.shape[0],
            rectified.strides[0] if rectified.shape[0] > 1 else max(rectified.shape[1], 1),
            Hinv.ctypes.data_as(C.POINTER(C.c_float)), int(flags)), "stx_rectify_u8_keep")
        return rectified

    def disparity_resident(self, slot_img0, slot_img1, shape, win=1, max_disp=0, cost_mode=COST_SAD, out=None):
        """disparity() over the two resident images left by rectify_keep (shape = (H, W))."""
        h, w = int(shape[0]), int(shape[1])
        if out is None:
            out = np.zeros((h, w), np.uint8)
        check(self._lib.stx_disparity_resident_u8(self._h, int(slot_img0), int(slot_img1), w, h, int(win),
                                                  int(max_disp), int(cost_mode), _ptr(out), max(w, 1)),
              "stx_disparity_resident_u8")
        return out

    def rectify_dev(self, src, dst, Hinv, flags=RECT_PRESERVE_DST):
        """src: [n,H,W] or [H,W] uint8 device tensor; dst likewise; Hinv: n x 9 host floats."""
        s3 = src if src.dim() == 3 else src.unsqueeze(0)
        d3 = dst if dst.dim() == 3 else dst.unsqueeze(0)
        n = int(s3.shape[0])
        Hinv = np.ascontiguousarray(Hinv, np.float32).reshape(n, 9)
        check(self._lib.stx_rectify_u8_dev(
            self._h, s3.data_ptr(), int(s3.shape[2]), int(s3.shape[1]), int(s3.stride(1)),
            int(s3.stride(0)), d3.data_ptr(), int(d3.shape[2]), int(d3.shape[1]),
            int(d3.stride(1)), int(d3.stride(0)), Hinv.ctypes.data_as(C.POINTER(C.c_float)), n,
            int(flags)), "stx_rectify_u8_dev")

    def rectify_rowband_dev(self, src, dst_band, dst_row0, Hinv, flags=RECT_PRESERVE_DST):
        """src: [H,W] uint8 device tensor (whole source); dst_band: [rows,W'] device tensor that
        receives destination rows [dst_row0, dst_row0+rows)."""
        Hinv = np.ascontiguousarray(Hinv, np.float32).reshape(9)
        check(self._lib.stx_rectify_rowband_u8_dev(
            self._h, src.data_ptr(), int(src.shape[1]), int(src.shape[0]), int(src.stride(0)),
            dst_band.data_ptr(), int(dst_band.shape[1]), int(dst_row0), int(dst_band.shape[0]),
            int(dst_band.stride(0)), Hinv.ctypes.data_as(C.POINTER(C.c_float)), int(flags)),
            "stx_rectify_rowband_u8_dev")

    # -- FAST -----------------------------------------------------------------
    def fast_detect(self, img, threshold=FAST_THRESHOLD, max_out=None):
        """(xy float32 [n,2] in raster order, total count)."""
        _check_image(img, "img")
        h, w = img.shape
        if max_out is None:
            max_out = max(1, w * h // 4)
        xy = np.zeros((max_out, 2), np.float32)
        cnt = C.c_int(0)
        check(self._lib.stx_fast_detect_u8(self._h, _ptr(img), w, h,
                                           img.strides[0] if h > 1 else max(w, 1), int(threshold),
                                           _ptr(xy), int(max_out), C.byref(cnt)), "stx_fast_detect_u8")
        return xy[:min(cnt.value, max_out)], cnt.value

    def fast_detect_dev(self, img, xy, count, threshold=FAST_THRESHOLD):
        """img: [H,W] uint8 device tensor; xy: [max,2] float32; count: int32[1] device tensor."""
        check(self._lib.stx_fast_detect_u8_dev(self._h, img.data_ptr(), int(img.shape[1]), int(img.shape[0]),
                                               int(img.stride(0)), int(threshold), xy.data_ptr(),
                                               int(xy.shape[0]), count.data_ptr()), "stx_fast_detect_u8_dev")

    # -- scoring + NMS ---------------------------------------------------------
    def gaussian_kernel(self, size=ST_WINDOW, sigma=1.0):
        """CreateGaussianKernel, stereo/Features.cpp:20-34 (host)."""
        out = np.zeros(size * size, np.float32)
        check(self._lib.stx_gaussian_kernel(int(size), float(sigma), out.ctypes.data_as(C.POINTER(C.c_float))),
              "stx_gaussian_kernel")
        return out.reshape(size, size)

    def score_cluster(self, grad_x, grad_y, xy, kernel=None, score_threshold=ST_THRESH,
                      cluster_dist=NMS_WINDOW):
        """(xy [m,2], score [m], index [m]) of the surviving features, best score first."""
        _check_image(grad_x, "grad_x")
        _check_image(grad_y, "grad_y")
        if grad_x.shape != grad_y.shape or grad_x.strides != grad_y.strides:
            raise ValueError("grad_x and grad_y must have the same shape and pitch")
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        n = xy.shape[0]
        k = np.ascontiguousarray(self.gaussian_kernel() if kernel is None else kernel, np.float32).reshape(9)
        h, w = grad_x.shape
        oxy = np.zeros((max(n, 1), 2), np.float32)
        osc = np.zeros(max(n, 1), np.float32)
        oidx = np.zeros(max(n, 1), np.int32)
        cnt = C.c_int(0)
        check(self._lib.stx_score_cluster(self._h, _ptr(grad_x), _ptr(grad_y), w, h,
                                          grad_x.strides[0] if h > 1 else max(w, 1), _ptr(xy), n,
                                          k.ctypes.data_as(C.POINTER(C.c_float)), float(score_threshold),
                                          float(cluster_dist), _ptr(oxy), _ptr(osc), _ptr(oidx), C.byref(cnt)),
              "stx_score_cluster")
        m = cnt.value
        return oxy[:m], osc[:m], oidx[:m]

    def sift_descriptors(self, grad_x, grad_y, xy, angle=None):
        """(angle [n], desc [n,128]) for the features at xy; `angle` gives the prior
        Feature::angle values (kept where the orientation window holds no gradient)."""
        _check_image(grad_x, "grad_x")
        _check_image(grad_y, "grad_y")
        if grad_x.shape != grad_y.shape or grad_x.strides != grad_y.strides:
            raise ValueError("grad_x and grad_y must have the same shape and pitch")
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        n = xy.shape[0]
        ang = np.zeros(max(n, 1), np.float32)
        if angle is not None:
            ang[:n] = np.asarray(angle, np.float32)
        desc = np.zeros((max(n, 1), DESC_LENGTH), np.float32)
        h, w = grad_x.shape
        check(self._lib.stx_sift_descriptors(self._h, _ptr(grad_x), _ptr(grad_y), w, h,
                                             grad_x.strides[0] if h > 1 else max(w, 1), _ptr(xy), n,
                                             _ptr(ang), _ptr(desc)), "stx_sift_descriptors")
        return ang[:n], desc[:n]

    def sift_descriptors_dev(self, grad_x, grad_y, xy, angle, desc):
        """Device tensors: grad_* [H,W] uint8, xy [n,2] float32, angle [n] (in/out), desc [n,128]."""
        check(self._lib.stx_sift_descriptors_dev(self._h, grad_x.data_ptr(), grad_y.data_ptr(),
                                                 int(grad_x.shape[1]), int(grad_x.shape[0]),
                                                 int(grad_x.stride(0)), xy.data_ptr(), int(xy.shape[0]),
                                                 angle.data_ptr(), desc.data_ptr()), "stx_sift_descriptors_dev")

    def cluster_features(self, xy, score, window):
        """Input positions of the features that survive ClusterFeatures, in input order."""
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        score = np.ascontiguousarray(score, np.float32).reshape(-1)
        n = xy.shape[0]
        if score.shape[0] != n:
            raise ValueError("xy and score must have the same length")
        idx = np.zeros(max(n, 1), np.int32)
        cnt = C.c_int(0)
        check(self._lib.stx_cluster_features(self._h, _ptr(xy), _ptr(score), n, int(window), _ptr(idx),
                                             C.byref(cnt)), "stx_cluster_features")
        return idx[:cnt.value]

    def score_cluster_dev(self, grad_x, grad_y, xy, kernel, score_threshold, cluster_dist, xy_out, score_out,
                          index_out, count):
        """Device tensors: grad_* [H,W] uint8, xy [n,2] float32, outputs sized n, count int32[1]."""
        k = np.ascontiguousarray(kernel, np.float32).reshape(9)
        check(self._lib.stx_score_cluster_dev(self._h, grad_x.data_ptr(), grad_y.data_ptr(), int(grad_x.shape[1]),
                                              int(grad_x.shape[0]), int(grad_x.stride(0)), xy.data_ptr(),
                                              int(xy.shape[0]), k.ctypes.data_as(C.POINTER(C.c_float)),
                                              float(score_threshold), float(cluster_dist), xy_out.data_ptr(),
                                              score_out.data_ptr(), index_out.data_ptr(), count.data_ptr()),
              "stx_score_cluster_dev")

    # -- hypothesis scoring ---------------------------------------------------
    def ransac_score(self, hyp, K1, K2, p1, p2, exclude=None, threshold=FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD,
                     want_errors=False, want_depths=False):
        """The pair loop of FindFundamentalMatrixWithRANSAC (stereo/Stereography.cpp:226-238) for
        every hypothesis record in `hyp` ([H, 36], layout in include/stereo_hip.h).  Returns
        (inliers [H] int32, err_sum [H] float32[, errors [H,n]][, depths [H,n,2]])."""
        hyp = np.ascontiguousarray(hyp, np.float32).reshape(-1, HYP_FLOATS)
        k1 = np.ascontiguousarray(K1, np.float32).reshape(9)
        k2 = np.ascontiguousarray(K2, np.float32).reshape(9)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        if p1.shape != p2.shape:
            raise ValueError("p1 and p2 must have the same length")
        nh, n = hyp.shape[0], p1.shape[0]
        ex = None
        if exclude is not None:
            ex = np.ascontiguousarray(exclude, np.int32).reshape(-1, 8)
            if ex.shape[0] != nh:
                raise ValueError("exclude must hold eight positions per hypothesis")
        inl = np.zeros(max(nh, 1), np.int32)
        esum = np.zeros(max(nh, 1), np.float32)
        errors = np.zeros((nh, n), np.float32) if want_errors else None
        depths = np.zeros((nh, n, 2), np.float32) if want_depths else None
        check(self._lib.stx_ransac_score(self._h, _ptr(hyp), nh, _ptr(k1), _ptr(k2), _ptr(p1), _ptr(p2), n,
                                         _ptr(ex) if ex is not None else None, float(threshold), _ptr(inl),
                                         _ptr(esum), _ptr(errors) if want_errors else None,
                                         _ptr(depths) if want_depths else None), "stx_ransac_score")
        out = (inl[:nh], esum[:nh])
        if want_errors:
            out += (errors,)
        if want_depths:
            out += (depths,)
        return out

    def triangulate(self, E, R, t, K_x, K_xprime, x, xprime, decomposed=True):
        """Triangulate (stereo/Stereography.cpp:383-443) for every pair at once, as the depth loop of
        main.cpp:289-316 calls it: pointX = K_x^-1 (x, 1), pointXPrime = K_xprime^-1 (xprime, 1),
        LindstromOptimisation with E, then the two ray depths with R, t = the outputs of
        DecomposeEssentialMatrix(E, R_other, R, t) (main.cpp:285 / Stereography.cpp:398), which
        stay with the caller.  Returns (depth0 [n], depth1 [n]) float32, NaN where Triangulate
        returns false.  (The reference's Triangulate also rescales the shared E in place on every
        call, :318, which moves E by an ulp or two from pair to pair; here every pair sees the E
        passed in.)"""
        rec = np.zeros((1, HYP_FLOATS), np.float32)
        rec[0, 0:9] = np.asarray(E, np.float32).reshape(9)
        rec[0, 9:18] = np.eye(3, dtype=np.float32).reshape(9)   # ReprojectionError's R, t: unused here
        rec[0, 21:30] = np.asarray(R, np.float32).reshape(9)
        rec[0, 30:33] = np.asarray(t, np.float32).reshape(3)
        rec[0, 33] = 1.0 if decomposed else 0.0
        _, _, depths = self.ransac_score(rec, K_x, K_xprime, x, xprime, want_depths=True)
        return depths[0, :, 0].copy(), depths[0, :, 1].copy()

    def ransac_score_dev(self, hyp, K1, K2, p1, p2, exclude, threshold, inliers, err_sum, errors=None, depths=None):
        """Device tensors: p1, p2 [n,2] float32, inliers [H] int32, err_sum [H] float32, errors
        [H,n] / depths [H,n,2] float32 or None; hyp [H,36], K1, K2, exclude [H,8] host arrays."""
        hyp = np.ascontiguousarray(hyp, np.float32).reshape(-1, HYP_FLOATS)
        k1 = np.ascontiguousarray(K1, np.float32).reshape(9)
        k2 = np.ascontiguousarray(K2, np.float32).reshape(9)
        ex = None if exclude is None else np.ascontiguousarray(exclude, np.int32).reshape(-1, 8)
        if ex is not None and ex.shape[0] != hyp.shape[0]:
            raise ValueError("exclude must hold eight positions per hypothesis")
        n = int(p1.shape[0])
        for t, shape, name in ((p1, (n, 2), "p1"), (p2, (n, 2), "p2"), (inliers, (hyp.shape[0],), "inliers"),
                               (err_sum, (hyp.shape[0],), "err_sum"), (errors, (hyp.shape[0], n), "errors"),
                               (depths, (hyp.shape[0], n, 2), "depths")):
            if t is not None and (tuple(t.shape) != shape or not t.is_contiguous() or t.element_size() != 4):
                raise ValueError("%s must be a contiguous 4-byte tensor of shape %s" % (name, shape))
        check(self._lib.stx_ransac_score_dev(self._h, _ptr(hyp), hyp.shape[0], _ptr(k1), _ptr(k2), p1.data_ptr(),
                                             p2.data_ptr(), int(p1.shape[0]), _ptr(ex) if ex is not None else None,
                                             float(threshold), inliers.data_ptr(), err_sum.data_ptr(),
                                             errors.data_ptr() if errors is not None else None,
                                             depths.data_ptr() if depths is not None else None),
              "stx_ransac_score_dev")

    def inlier_sum_dev(self, errors, threshold, inliers, err_sum):
        """Stereography.cpp:232-236 over rows of given errors.  Device tensors: errors [H,n] float32,
        inliers [H] int32, err_sum [H] float32."""
        nh, n = int(errors.shape[0]), int(errors.shape[1])
        for t, shape, name in ((errors, (nh, n), "errors"), (inliers, (nh,), "inliers"), (err_sum, (nh,), "err_sum")):
            if tuple(t.shape) != shape or not t.is_contiguous() or t.element_size() != 4:
                raise ValueError("%s must be a contiguous 4-byte tensor of shape %s" % (name, shape))
        check(self._lib.stx_inlier_sum_dev(self._h, errors.data_ptr(), nh, n, float(threshold), inliers.data_ptr(),
                                           err_sum.data_ptr()), "stx_inlier_sum_dev")

    # -- D ------------------------------------------------------------------
    def disparity(self, img0, img1, win=1, max_disp=0, cost_mode=COST_SAD, out_dtype=np.uint8, out=None):
        """out: optional C-contiguous [H,W] array of out_dtype to write into (a caller in a loop
        reuses it; a fresh numpy array costs a page fault per 4 KiB on first touch)."""
        _check_image(img0, "img0")
        _check_image(img1, "img1")
        if img0.shape != img1.shape:
            raise ValueError("img0 and img1 must have the same shape")
        if img0.strides[0] != img1.strides[0] and img0.shape[0] > 1:
            img1 = np.ascontiguousarray(img1)
            img0 = np.ascontiguousarray(img0)
        h, w = img0.shape
        out_dtype = np.dtype(out_dtype)
        if out is None:
            out = np.zeros((h, w), out_dtype)
        elif out.shape != (h, w) or out.dtype != out_dtype or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous %s array of shape %s" % (out_dtype, (h, w)))
        stride = img0.strides[0] if h > 1 else max(w, 1)
        fn = {1: self._lib.stx_disparity_u8, 2: self._lib.stx_disparity_u16}[out_dtype.itemsize]
        check(fn(self._h, _ptr(img0), _ptr(img1), w, h, stride, int(win), int(max_disp),
                 int(cost_mode), _ptr(out), max(w, 1) * out_dtype.itemsize), "stx_disparity")
        return out

    def rectify_disparity(self, img0, img1, Hinv0, Hinv1, win=1, max_disp=0, cost_mode=COST_SAD,
                          want_rectified=False):
        """main.cpp:362-380 on the device: rectify both images into zeroed destinations,
        then scan.  Returns depth, or (depth, rect0, rect1)."""
        _check_image(img0, "img0")
        _check_image(img1, "img1")
        if img0.shape != img1.shape:
            raise ValueError("img0 and img1 must have the same shape")
        img0, img1 = np.ascontiguousarray(img0), np.ascontiguousarray(img1)
        h, w = img0.shape
        out = np.zeros((h, w), np.uint8)
        r0 = np.zeros((h, w), np.uint8) if want_rectified else None
        r1 = np.zeros((h, w), np.uint8) if want_rectified else None
        fp = C.POINTER(C.c_float)
        h0 = np.ascontiguousarray(Hinv0, np.float32).reshape(9)
        h1 = np.ascontiguousarray(Hinv1, np.float32).reshape(9)
        check(self._lib.stx_rectify_disparity_u8(
            self._h, _ptr(img0), _ptr(img1), w, h, max(w, 1), h0.ctypes.data_as(fp),
            h1.ctypes.data_as(fp), int(win), int(max_disp), int(cost_mode), _ptr(out), max(w, 1),
            _ptr(r0) if want_rectified else None, _ptr(r1) if want_rectified else None, max(w, 1)),
            "stx_rectify_disparity_u8")
        return (out, r0, r1) if want_rectified else out

    def disparity_dev(self, img0, img1, out, win=1, max_disp=0, cost_mode=COST_SAD):
        """img0/img1: [n,H,W] or [H,W] uint8 device tensors; out: same shape, uint8 or int16."""
        a = img0 if img0.dim() == 3 else img0.unsqueeze(0)
        b = img1 if img1.dim() == 3 else img1.unsqueeze(0)
        o = out if out.dim() == 3 else out.unsqueeze(0)
        if a.shape != b.shape or a.stride() != b.stride() or o.shape != a.shape:
            raise ValueError("img0, img1 and out must agree in shape and layout")
        n, h, w = (int(v) for v in a.shape)
        ob = o.element_size()
        check(self._lib.stx_disparity_dev(
            self._h, a.data_ptr(), b.data_ptr(), n, w, h, int(a.stride(1)), int(a.stride(0)),
            int(win), int(max_disp), int(cost_mode), o.data_ptr(), int(o.stride(1)) * ob,
            int(o.stride(0)) * ob, ob), "stx_disparity_dev")

    def disparity_dev_noalloc(self, img0, img1, out, win=1, max_disp=0, cost_mode=COST_SAD):
        """disparity_dev through the entry point that allocates nothing (static device scratch;
        scans of more than 256 disparities are limited to NOALLOC_MAX_PIXELS pixels)."""
        a = img0 if img0.dim() == 3 else img0.unsqueeze(0)
        b = img1 if img1.dim() == 3 else img1.unsqueeze(0)
        o = out if out.dim() == 3 else out.unsqueeze(0)
        if a.shape != b.shape or a.stride() != b.stride() or o.shape != a.shape:
            raise ValueError("img0, img1 and out must agree in shape and layout")
        n, h, w = (int(v) for v in a.shape)
        ob = o.element_size()
        check(self._lib.stx_disparity_dev_noalloc(
            self._h, a.data_ptr(), b.data_ptr(), n, w, h, int(a.stride(1)), int(a.stride(0)),
            int(win), int(max_disp), int(cost_mode), o.data_ptr(), int(o.stride(1)) * ob,
            int(o.stride(0)) * ob, ob), "stx_disparity_dev_noalloc")

    def disparity_rowband_dev(self, img0, img1, band_row0, full_h, row0, rows, out, win=1,
                              max_disp=0, cost_mode=COST_SAD):
        """img0/img1: [h_band, W] device tensors holding rows [band_row0, band_row0+h_band)."""
        h, w = int(img0.shape[0]), int(img0.shape[1])
        ob = out.element_size()
        check(self._lib.stx_disparity_rowband_dev(
            self._h, img0.data_ptr(), img1.data_ptr(), w, h, int(band_row0), int(full_h),
            int(img0.stride(0)), int(row0), int(rows), int(win), int(max_disp), int(cost_mode),
            out.data_ptr(), int(out.stride(0)) * ob, ob), "stx_disparity_rowband_dev")


class MultiContext:
    """stx_multi: several GPUs behind one handle in ONE process (the C++ drop-in's way to more
    than one GPU; one context, stream and host thread per device, no collective)."""

    def __init__(self, device_ids=None):
        self._lib = _lib.load_library()
        h = C.c_void_p()
        if device_ids is None:
            check(self._lib.stx_multi_create(C.byref(h), None, 0), "stx_multi_create")
        else:
            ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
            check(self._lib.stx_multi_create(C.byref(h), ids, len(device_ids)), "stx_multi_create")
        self._h = h
        self.size = int(self._lib.stx_multi_size(h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.stx_multi_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    def set_match_algo(self, algo):
        for k in range(self.size):
            check(self._lib.stx_set_match_algo(self._lib.stx_multi_context(self._h, k), int(algo)),
                  "stx_set_match_algo")

    def disparity_batch(self, img0, img1, win=1, max_disp=0, cost_mode=COST_SAD, out=None):
        """img0/img1: [n,H,W] uint8 arrays (a batch of pairs, dealt to the devices in contiguous
        shares).  Returns the [n,H,W] uint8 maps (written into `out` when given: a C-contiguous
        [n,H,W] uint8 array, e.g. page-locked memory)."""
        a, b = np.ascontiguousarray(img0, np.uint8), np.ascontiguousarray(img1, np.uint8)
        if a.ndim != 3 or a.shape != b.shape:
            raise ValueError("img0 and img1 must be [n,H,W] arrays of one shape")
        n, h, w = a.shape
        if out is None:
            out = np.zeros((n, h, w), np.uint8)
        elif not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.shape == a.shape
                  and out.flags.c_contiguous and out.flags.writeable):
            raise ValueError("out must be a writable C-contiguous [n,H,W] uint8 array")
        check(self._lib.stx_disparity_batch_u8(self._h, _ptr(a), _ptr(b), n, w, h, max(w, 1), max(w, 1) * h,
                                               int(win), int(max_disp), int(cost_mode), _ptr(out), max(w, 1),
                                               max(w, 1) * h), "stx_disparity_batch_u8")
        return out

    def disparity_rowbands(self, img0, img1, win=1, max_disp=0, cost_mode=COST_SAD, out_dtype=np.uint8):
        """One pair, sharded over the devices by row bands with win//2 halo rows."""
        _check_image(img0, "img0")
        _check_image(img1, "img1")
        if img0.shape != img1.shape:
            raise ValueError("img0 and img1 must have the same shape")
        a, b = np.ascontiguousarray(img0), np.ascontiguousarray(img1)
        h, w = a.shape
        out_dtype = np.dtype(out_dtype)
        out = np.zeros((h, w), out_dtype)
        fn = {1: self._lib.stx_disparity_rowband_u8, 2: self._lib.stx_disparity_rowband_u16}[out_dtype.itemsize]
        check(fn(self._h, _ptr(a), _ptr(b), w, h, max(w, 1), int(win), int(max_disp), int(cost_mode), _ptr(out),
                 max(w, 1) * out_dtype.itemsize), "stx_disparity_rowband")
        return out

    def rectify(self, original, rectified, Hinv, flags=RECT_PRESERVE_DST):
        """As Context.rectify, destination row bands over the devices (every device holds the source)."""
        _check_image(original, "original")
        _check_image(rectified, "rectified", writable=True)
        Hinv = np.ascontiguousarray(Hinv, np.float32).reshape(9)
        check(self._lib.stx_rectify_multi_u8(
            self._h, _ptr(original), original.shape[1], original.shape[0],
            original.strides[0] if original.shape[0] > 1 else max(original.shape[1], 1),
            _ptr(rectified), rectified.shape[1], rectified.shape[0],
            rectified.strides[0] if rectified.shape[0] > 1 else max(rectified.shape[1], 1),
            Hinv.ctypes.data_as(C.POINTER(C.c_float)), int(flags)), "stx_rectify_multi_u8")
        return rectified

    def match(self, desc1, score1, desc2, score2, dist_limit=MAX_DIST_BETWEEN_MATCHES):
        """As Context.match, list1's queries sharded over the devices."""
        desc1 = np.ascontiguousarray(desc1, np.float32).reshape(-1, DESC_LENGTH)
        desc2 = np.ascontiguousarray(desc2, np.float32).reshape(-1, DESC_LENGTH)
        score1 = np.ascontiguousarray(score1, np.float32).reshape(-1)
        score2 = np.ascontiguousarray(score2, np.float32).reshape(-1)
        n1, n2 = desc1.shape[0], desc2.shape[0]
        if score1.shape[0] != n1 or score2.shape[0] != n2:
            raise ValueError("score arrays must have one entry per descriptor")
        closest = np.full(n1, -1, np.int32)
        second = np.full(n1, -1, np.int32)
        min_dist = np.full(n1, -1.0, np.float32)
        second_dist = np.full(n1, -1.0, np.float32)
        status = np.zeros(n1, np.uint8)
        check(self._lib.stx_match_multi(self._h, _ptr(desc1), _ptr(score1), n1, _ptr(desc2), _ptr(score2), n2,
                                        C.c_float(dist_limit), _ptr(closest), _ptr(second), _ptr(min_dist),
                                        _ptr(second_dist), _ptr(status)), "stx_match_multi")
        return closest, second, min_dist, second_dist, status


def _check_image(a, name, writable=False):
    if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.ndim != 2:
        raise TypeError("%s must be a 2-D uint8 numpy array (CV_8U, one channel)" % name)
    if a.size == 0:
        return
    if a.shape[1] > 1 and a.strides[1] != 1:
        raise ValueError("%s must have contiguous rows" % name)
    if a.shape[0] > 1 and a.strides[0] < a.shape[1]:
        raise ValueError("%s has overlapping rows" % name)
    if writable and not a.flags.writeable:
        raise ValueError("%s must be writable" % name)


_DEFAULT = None


def default_context():
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Context()
    return _DEFAULT


# --------------------------------------------------------------------------
# The three reference signatures
# --------------------------------------------------------------------------

def MatchDescriptors(list1, list2, distLimitBetweenMatches, ctx=None, out=None):
    """MatchDescriptors, stereo/Features.cpp:1020-1079.

    list1/list2: sequences of Feature.  Returns a list of (Feature, Feature)
    pairs -- copies, with distFromBestMatch set on both -- in ascending order of
    the list1 index.  A query whose closest or second-closest is unset prints
    the reference's line "Error: failed to match feature <i>" (:1052).
    """
    ctx = ctx or default_context()
    out = out or sys.stdout
    n1, n2 = len(list1), len(list2)
    d1 = np.stack([np.asarray(f.desc, np.float32) for f in list1]) if n1 else np.zeros((0, 128), np.float32)
    d2 = np.stack([np.asarray(f.desc, np.float32) for f in list2]) if n2 else np.zeros((0, 128), np.float32)
    s1 = np.array([f.score for f in list1], np.float32)
    s2 = np.array([f.score for f in list2], np.float32)
    closest, second, min_dist, second_dist, status = ctx.match(d1, s1, d2, s2, distLimitBetweenMatches)
    matches = []
    for i in range(n1):
        st = status[i]
        if st == MATCH_FAILED:
            out.write("Error: failed to match feature %d\n" % i)
            continue
        if st != MATCH_ACCEPTED:
            continue
        a = copy.deepcopy(list1[i])
        b = copy.deepcopy(list2[int(closest[i])])
        a.distFromBestMatch = float(min_dist[i])
        b.distFromBestMatch = float(min_dist[i])
        matches.append((a, b))
    return matches


def FindFASTFeatures(img, features, ctx=None):
    """FindFASTFeatures, stereo/Features.cpp:407-471: appends one Feature per FAST-12 corner
    (p = (x, y), raster order) to `features`; returns True like the reference."""
    ctx = ctx or default_context()
    xy, _ = ctx.fast_detect(img, FAST_THRESHOLD)
    for x, y in xy:
        features.append(Feature(p=(float(x), float(y))))
    return True


def CreateSIFTDescriptors(grad_x, grad_y, features, descriptors, ctx=None):
    """CreateSIFTDescriptors, stereo/Features.cpp:827-937, on the CV_8U gradient images of
    :830-837: sets angle and desc of every feature in place and appends the descriptors."""
    ctx = ctx or default_context()
    xy = np.array([f.p for f in features], np.float32).reshape(-1, 2)
    ang, desc = ctx.sift_descriptors(grad_x, grad_y, xy, [f.angle for f in features])
    for f, a, d in zip(features, ang, desc):
        f.angle = float(a)
        f.desc = d.copy()
        descriptors.append(f.desc)
    return True


def ClusterFeatures(features, windowSize, ctx=None):
    """ClusterFeatures, stereo/Features.cpp:48-80: the features that are the strongest inside
    their window, in input order."""
    ctx = ctx or default_context()
    xy = np.array([f.p for f in features], np.float32).reshape(-1, 2)
    score = np.array([f.score for f in features], np.float32)
    return [features[int(i)] for i in ctx.cluster_features(xy, score, windowSize)]


def ScoreAndClusterFeatures(grad_x, grad_y, features, scoreThreshold=ST_THRESH,
                            distanceForWithinCluster=NMS_WINDOW, ctx=None):
    """ScoreAndClusterFeatures, stereo/Features.cpp:672-786, on the CV_8U gradient images the
    reference makes with OpenCV's GaussianBlur + Sobel (:684-690; they stay with the caller).
    Returns the surviving features, best score first, as copies with `score` set."""
    ctx = ctx or default_context()
    xy = np.array([f.p for f in features], np.float32).reshape(-1, 2)
    _, score, index = ctx.score_cluster(grad_x, grad_y, xy, None, scoreThreshold, distanceForWithinCluster)
    out = []
    for s, i in zip(score, index):
        f = copy.deepcopy(features[int(i)])
        f.score = float(s)
        out.append(f)
    return out


def RectifyImage(original, rectified, H, ctx=None):
    """RectifyImage, stereo/Stereography.cpp:596-631.

    H is the forward homography (original -> rectified); it is inverted here
    as the reference does at :617.  `rectified` is modified in place; pixels
    whose source falls outside the strict bounds keep their prior value.
    """
    ctx = ctx or default_context()
    Hinv = ctx.invert_homography(H)
    ctx.rectify(original, rectified, Hinv, RECT_PRESERVE_DST)


def ComputeDepthImage(img0, img1, ctx=None, out=None, win=1, max_disp=0, cost_mode=COST_SAD):
    """ComputeDepthImage, stereo/Stereography.cpp:649-691.

    With the default win=1, max_disp=0 this is the reference's scan.  A size
    mismatch prints "Cannot compute depth!" and returns an empty 0x0 image
    (:655-659).  win/max_disp/cost_mode select the D1-ext generalisation.
    """
    out = out or sys.stdout
    if img0.shape[1] != img1.shape[1] or img0.shape[0] != img1.shape[0]:
        out.write("Cannot compute depth!\n")
        return np.zeros((0, 0), np.uint8)
    ctx = ctx or default_context()
    return ctx.disparity(img0, img1, win=win, max_disp=max_disp, cost_mode=cost_mode)
