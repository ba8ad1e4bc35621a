"""End-to-end (host-pointer, PCIe-inclusive) rates of the host C-ABI entry points: the plain-copy
path against the pinned-staging pipeline at several thread / chunk settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_amd
from stereo_amd import synth

ctx = stereo_amd.Context(0)


def best(fn, n=15):
    fn(); fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3


W, H, D = 2880, 1988, 256
l, r = synth.rectified_pair(2024, H, W, D)
src = synth.texture(3, 1080, 1920)
Hinv = ctx.invert_homography(synth.homography(3, 1920, 1080))
dst = np.zeros_like(src)
dout = np.zeros_like(l)
n = 32768
d1, s1, d2, s2 = synth.descriptor_sets(42, n, n)
pairs = [synth.rectified_pair(400 + k, 1080, 1920, 192) for k in range(4)]
ba = np.stack([p[0] for p in pairs] * 8)
bb = np.stack([p[1] for p in pairs] * 8)
mc = stereo_amd.MultiContext([0])

settings = [("plain", None)] + [("pipe t%d c%d" % (t, c), (t, c)) for t, c in
                                ((1, 1), (4, 1), (8, 1), (4, 2), (8, 2), (4, 4), (8, 4), (8, 8), (0, 0))]
print("%-12s %21s %21s %21s %21s %21s %21s" % ("setting", "disp 9x9 D256 cfg2", "disp ref-mode cfg2", "rectify 1080p keep",
                                                "rectify 1080p zero", "R+R+D fused cfg2", "match 32k x 32k"))
for name, st in settings:
    if st is None:
        ctx.set_host_pipeline(False)
    else:
        ctx.set_host_pipeline(True, st[0], st[1])
    row = []
    row.append(best(lambda: ctx.disparity(l, r, win=9, max_disp=D, out=dout)))
    row.append(best(lambda: ctx.disparity(l, r, out=dout), 7))
    row.append(best(lambda: ctx.rectify(src, dst, Hinv)))
    row.append(best(lambda: ctx.rectify(src, dst, Hinv, stereo_amd.api.RECT_DST_IS_ZERO)))
    row.append(best(lambda: ctx.rectify_disparity(l, r, Hinv, Hinv, win=9, max_disp=D), 7))
    row.append(best(lambda: ctx.match(d1, s1, d2, s2, 1000.0), 5))
    print("%-12s " % name + " ".join("%8.3f / %8.3f ms" % v for v in row), flush=True)
ctx.set_host_pipeline(True, 0, 0)
# page-locked caller buffers (torch pin_memory; hipHostMalloc in C++): no staging copies
import torch


def pinned(a):
    t = torch.empty(a.shape, dtype=torch.uint8, pin_memory=True)
    v = t.numpy()
    v[...] = a
    return t, v


keep = [pinned(x) for x in (l, r, dout, src, dst)]
pl, pr, pdout, psrc, pdst = [k[1] for k in keep]
row = [best(lambda: ctx.disparity(pl, pr, win=9, max_disp=D, out=pdout)), best(lambda: ctx.disparity(pl, pr, out=pdout), 7),
       best(lambda: ctx.rectify(psrc, pdst, Hinv)), best(lambda: ctx.rectify(psrc, pdst, Hinv, stereo_amd.api.RECT_DST_IS_ZERO))]
print("%-12s " % "pinned t0 c0" + " ".join("%8.3f / %8.3f ms" % v for v in row), flush=True)
b, m = best(lambda: mc.disparity_batch(ba, bb, win=9, max_disp=192), 5)
print("stx_disparity_batch_u8, 32 x 1080p pairs, D=192, 1 device: best %.2f ms median %.2f ms = %.0f pairs/s" % (b, m, 32 / m * 1e3))
def pinned3(a):
    t = torch.empty(a.shape, dtype=torch.uint8, pin_memory=True)
    v = t.numpy()
    v[...] = a
    return t, v


keep3 = [pinned3(x) for x in (ba, bb, np.zeros_like(ba))]
b, m = best(lambda: mc.disparity_batch(keep3[0][1], keep3[1][1], win=9, max_disp=192, out=keep3[2][1]), 5)
print("the same batch from and into page-locked arrays:               best %.2f ms median %.2f ms = %.0f pairs/s" % (b, m, 32 / m * 1e3))
print("(best / median of the calls; outputs written into reused arrays except for the fused call and matching)")
# matching at config 5's size through host pointers: separate arrays vs the reference's 540-byte records
n = 200000
d1, s1, d2, s2 = synth.descriptor_sets(42, n, n)
feature = np.dtype([("scale", np.int32), ("p", np.float32, 2), ("score", np.float32), ("angle", np.float32),
                    ("desc", np.float32, 128), ("distFromBestMatch", np.float32), ("depth", np.float32)])
r1, r2 = np.zeros(n, feature), np.zeros(n, feature)
r1["desc"], r1["score"], r2["desc"], r2["score"] = d1, s1, d2, s2
t0 = time.perf_counter()
pk = (np.ascontiguousarray(r1["desc"]), np.ascontiguousarray(r1["score"]), np.ascontiguousarray(r2["desc"]), np.ascontiguousarray(r2["score"]))
t_pack = (time.perf_counter() - t0) * 1e3
print("stx_match 200k x 200k, host arrays:    best %.2f ms median %.2f ms  (+ %.1f ms to split the records on the host, numpy)" % (best(lambda: ctx.match(d1, s1, d2, s2, 1000.0), 3) + (t_pack,)))
print("stx_match_features 200k x 200k records: best %.2f ms median %.2f ms  (records uploaded as they lie)" % best(lambda: ctx.match_features(r1, r2, 20, 12, 1000.0), 3))
