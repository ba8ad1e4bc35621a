"""End-to-end (host-pointer, PCIe-inclusive) rates of the host C-ABI entry points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_amd
from stereo_amd import synth

ctx = stereo_amd.Context(0)
def best(fn, n=7):
    fn(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts), float(np.median(ts))

W, H, D = 2880, 1988, 256
l, r = synth.rectified_pair(2024, H, W, D)
units = (D * (W - D) + D * (D + 1) // 2) * H
b, m = best(lambda: ctx.disparity(l, r, win=9, max_disp=D))
print("disparity 9x9 D=256 %dx%d host API: best %.3f ms median %.3f ms -> %.3g Mpix*disp/s (HBM-resident kernel: 0.375 ms)" % (W, H, b * 1e3, m * 1e3, units / m / 1e6))
b, m = best(lambda: ctx.disparity(l, r))
print("disparity reference mode host API: best %.3f ms median %.3f ms -> %.3g Mpix*disp/s" % (b * 1e3, m * 1e3, W * W * H / m / 1e6))
src = synth.texture(3, 1080, 1920); Hinv = ctx.invert_homography(synth.homography(3, 1920, 1080))
dst = np.zeros_like(src)
b, m = best(lambda: ctx.rectify(src, dst, Hinv))
print("rectify 1920x1080 host API (dst uploaded, PRESERVE): best %.3f ms median %.3f ms -> %.3g Mpix/s" % (b * 1e3, m * 1e3, 1920 * 1080 / m / 1e6))
n = 32768
d1, s1, d2, s2 = synth.descriptor_sets(42, n, n)
b, m = best(lambda: ctx.match(d1, s1, d2, s2, 1000.0), 5)
print("match %dx%d host API: best %.3f ms median %.3f ms -> %.3g descriptors/s (H2D of 2 x %.0f MB dominates)" % (n, n, b * 1e3, m * 1e3, n / m, n * 512 / 1e6))
