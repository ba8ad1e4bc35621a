"""stx_rectify_u8 on one 1080p image through host pointers (destination declared zero): time and a
check against the device-resident call.  Environment: STX_HOST_ZERO_COPY=0 switches the kernel's direct stores to host memory off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stereo_amd import api, synth

h, w = 1080, 1920
ctx = api.Context(0)
src = synth.texture(70, h, w)
Hinv = ctx.invert_homography(synth.homography(80, w, h))
dst = np.zeros((h, w), np.uint8)
ref = np.zeros((h, w), np.uint8)
ctx.set_host_pipeline(False)
ctx.rectify(src, ref, Hinv, api.RECT_DST_IS_ZERO)
ctx.set_host_pipeline(True)
for _ in range(5):
    ctx.rectify(src, dst, Hinv, api.RECT_DST_IS_ZERO)
ts = []
for _ in range(40):
    t0 = time.perf_counter(); ctx.rectify(src, dst, Hinv, api.RECT_DST_IS_ZERO); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("STX_HOST_ZERO_COPY=%s: best %.3f median %.3f ms, equal to the plain path: %s" % (
    os.environ.get("STX_HOST_ZERO_COPY", "1"), ts[0], ts[len(ts) // 2], bool((dst == ref).all())))
