"""ctypes wrapper of oracle/liboracle.so (the CPU restatement).

TEST INFRASTRUCTURE.  May be imported only by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the stereo_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# STEREO_ORACLE_LIB: another build of the same source (the sanitizer build of `make asan-check`)
_SO = os.environ.get("STEREO_ORACLE_LIB", os.path.join(_HERE, "liboracle.so"))
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "stereo_oracle.c")
    if "STEREO_ORACLE_LIB" in os.environ:
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_descriptor_distance.restype = C.c_float
        _lib.oracle_bilinear.restype = C.c_uint8
        _lib.oracle_fast_selfcheck.restype = C.c_long
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(lib().oracle_descriptor_distance(_p(a), _p(b)))


def match(desc1, score1, desc2, score2, dist_limit=1000.0):
    desc1 = np.ascontiguousarray(desc1, np.float32).reshape(-1, 128)
    desc2 = np.ascontiguousarray(desc2, np.float32).reshape(-1, 128)
    score1 = np.ascontiguousarray(score1, np.float32)
    score2 = np.ascontiguousarray(score2, np.float32)
    n1, n2 = len(desc1), len(desc2)
    closest = np.full(n1, -7, np.int32)
    second = np.full(n1, -7, np.int32)
    md = np.zeros(n1, np.float32)
    sd = np.zeros(n1, np.float32)
    st = np.zeros(n1, np.uint8)
    lib().oracle_match(_p(desc1), _p(score1), n1, _p(desc2), _p(score2), n2,
                       C.c_float(dist_limit), _p(closest), _p(second), _p(md), _p(sd), _p(st))
    return closest, second, md, sd, st


def invert3x3(H):
    H = np.ascontiguousarray(H, np.float32).reshape(9)
    out = np.empty(9, np.float32)
    lib().oracle_invert3x3(_p(H), _p(out))
    return out.reshape(3, 3)


def bilinear(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return int(lib().oracle_bilinear(_p(img), C.c_size_t(img.strides[0]), C.c_float(x), C.c_float(y)))


def rectify(src, dst, Hinv):
    src = np.ascontiguousarray(src, np.uint8)
    assert dst.flags.c_contiguous and dst.dtype == np.uint8
    Hinv = np.ascontiguousarray(Hinv, np.float32).reshape(9)
    lib().oracle_rectify(_p(src), src.shape[1], src.shape[0], C.c_size_t(max(src.shape[1], 1)),
                         _p(dst), dst.shape[1], dst.shape[0], C.c_size_t(max(dst.shape[1], 1)),
                         _p(Hinv))
    return dst


def disparity_ref(img0, img1):
    img0 = np.ascontiguousarray(img0, np.uint8)
    img1 = np.ascontiguousarray(img1, np.uint8)
    h, w = img0.shape
    out = np.full((h, w), 0xAA, np.uint8)
    lib().oracle_disparity_ref(_p(img0), _p(img1), w, h, C.c_size_t(max(w, 1)), _p(out),
                               C.c_size_t(max(w, 1)))
    return out


def disparity_ext(img0, img1, win, max_disp, cost_mode=0, out_dtype=np.uint8, sliding=False):
    img0 = np.ascontiguousarray(img0, np.uint8)
    img1 = np.ascontiguousarray(img1, np.uint8)
    h, w = img0.shape
    dt = np.dtype(out_dtype)
    out = np.zeros((h, w), dt)
    fn = lib().oracle_disparity_ext_sliding if sliding else lib().oracle_disparity_ext
    rc = fn(_p(img0), _p(img1), w, h, C.c_size_t(max(w, 1)), int(win), int(max_disp),
            int(cost_mode), _p(out), C.c_size_t(max(w, 1) * dt.itemsize), dt.itemsize)
    if rc != 0:
        raise ValueError("oracle_disparity_ext: invalid arguments")
    return out


def check_sequential12(points, p_b, pb):
    return bool(lib().oracle_check_sequential12((C.c_int * 16)(*[int(v) for v in points]), int(p_b), int(pb)))


def ring_run12(points, p_b, pb):
    return bool(lib().oracle_ring_run12((C.c_int * 16)(*[int(v) for v in points]), int(p_b), int(pb)))


def three_of_four(i1, i5, i9, i13, pb, p_b):
    return bool(lib().oracle_three_of_four(int(i1), int(i5), int(i9), int(i13), int(pb), int(p_b)))


def fast_selfcheck():
    return int(lib().oracle_fast_selfcheck())


def fast_features(img, threshold=30):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = max(1, w * h)
    xy = np.zeros((cap, 2), np.float32)
    n = lib().oracle_fast_features(_p(img), w, h, C.c_size_t(max(w, 1)), int(threshold), _p(xy), cap)
    return xy[:n].copy()


def gaussian_kernel(size, sigma):
    out = np.zeros(size * size, np.float32)
    lib().oracle_gaussian_kernel(int(size), C.c_float(sigma), _p(out))
    return out.reshape(size, size)


def shi_tomasi_score(gx, gy, x, y, kernel):
    gx = np.ascontiguousarray(gx, np.uint8)
    gy = np.ascontiguousarray(gy, np.uint8)
    k = np.ascontiguousarray(kernel, np.float32).reshape(9)
    h, w = gx.shape
    f = lib().oracle_shi_tomasi_score
    f.restype = C.c_float
    return float(f(_p(gx), _p(gy), w, h, C.c_size_t(w), C.c_float(x), C.c_float(y), _p(k)))


def score_and_cluster(gx, gy, xy, kernel, score_threshold, cluster_dist):
    gx = np.ascontiguousarray(gx, np.uint8)
    gy = np.ascontiguousarray(gy, np.uint8)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    k = np.ascontiguousarray(kernel, np.float32).reshape(9)
    h, w = gx.shape
    n = xy.shape[0]
    oxy = np.zeros((max(n, 1), 2), np.float32)
    osc = np.zeros(max(n, 1), np.float32)
    oidx = np.zeros(max(n, 1), np.int32)
    m = lib().oracle_score_and_cluster(_p(gx), _p(gy), w, h, C.c_size_t(max(w, 1)), _p(xy), n, _p(k),
                                       C.c_float(score_threshold), C.c_float(cluster_dist), _p(oxy), _p(osc),
                                       _p(oidx))
    assert m >= 0
    return oxy[:m].copy(), osc[:m].copy(), oidx[:m].copy()


def cluster_features(xy, score, window):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    score = np.ascontiguousarray(score, np.float32).reshape(-1)
    n = xy.shape[0]
    idx = np.zeros(max(n, 1), np.int32)
    m = lib().oracle_cluster_features(_p(xy), _p(score), n, int(window), _p(idx))
    return idx[:m].copy()


def sift_descriptors(gx, gy, xy, angle=None):
    gx = np.ascontiguousarray(gx, np.uint8)
    gy = np.ascontiguousarray(gy, np.uint8)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    n = xy.shape[0]
    h, w = gx.shape
    ang = np.zeros(max(n, 1), np.float32) if angle is None else np.array(angle, np.float32)
    desc = np.zeros((max(n, 1), 128), np.float32)
    k9 = gaussian_kernel(9, 1.0).reshape(81)
    k4 = gaussian_kernel(4, 1.0).reshape(16)
    lib().oracle_sift_descriptors(_p(gx), _p(gy), w, h, C.c_size_t(max(w, 1)), _p(xy), n, _p(k9), _p(k4),
                                  _p(ang), _p(desc))
    return ang[:n].copy(), desc[:n].copy()


HYP_FLOATS = 36


def ransac_score(hyp, K1, K2, p1, p2, exclude=None, threshold=70.0):
    """(inliers [H], err_sum [H], errors [H,n], depths [H,n,2]) -- oracle_ransac_score."""
    hyp = np.ascontiguousarray(hyp, np.float32).reshape(-1, HYP_FLOATS)
    k1 = np.ascontiguousarray(K1, np.float32).reshape(9)
    k2 = np.ascontiguousarray(K2, np.float32).reshape(9)
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    nh, n = hyp.shape[0], p1.shape[0]
    ex = None if exclude is None else np.ascontiguousarray(exclude, np.int32).reshape(nh, 8)
    inl = np.zeros(max(nh, 1), np.int32)
    esum = np.zeros(max(nh, 1), np.float32)
    errors = np.zeros((nh, n), np.float32)
    depths = np.zeros((nh, n, 2), np.float32)
    lib().oracle_ransac_score(_p(hyp), nh, _p(k1), _p(k2), _p(p1), _p(p2), n, _p(ex) if ex is not None else None,
                              C.c_float(threshold), _p(inl), _p(esum), _p(errors), _p(depths))
    return inl[:nh].copy(), esum[:nh].copy(), errors, depths
