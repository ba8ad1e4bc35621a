"""stx_disparity_u8 through host pointers: config 2 (9x9, D = 256), reference mode, 1080p D = 192 and the
741x500 pair of config 1; time and a check against the plain-copy path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stereo_amd import api, synth

ctx = api.Context(0)
for name, (h, w, D, win) in (("config 2, 9x9 D=256", (1988, 2880, 256, 9)), ("config 2, reference mode", (1988, 2880, 0, 1)),
                             ("1080p, 9x9 D=192", (1080, 1920, 192, 9)), ("741x500, 9x9 D=64", (500, 741, 64, 9))):
    l, r = synth.rectified_pair(2024, h, w, max(D, 16))
    out = np.zeros((h, w), np.uint8)
    ctx.set_host_pipeline(False)
    ref = ctx.disparity(l, r, win=win, max_disp=D)
    ctx.set_host_pipeline(True)
    for _ in range(4):
        ctx.disparity(l, r, win=win, max_disp=D, out=out)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); ctx.disparity(l, r, win=win, max_disp=D, out=out); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("%-26s best %.3f median %.3f ms  equal: %s" % (name, ts[0], ts[len(ts) // 2], bool((out == ref).all())))
