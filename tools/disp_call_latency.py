import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
from stereo_amd import api, synth
ctx = api.Context()
l, r = synth.rectified_pair(3, 64, 256, 32)
a, b = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
o = torch.empty_like(a)
for _ in range(100): ctx.disparity_dev(a, b, o, win=9, max_disp=32)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5000): ctx.disparity_dev(a, b, o, win=9, max_disp=32)
t1 = time.perf_counter()
ctx.synchronize()
t2 = time.perf_counter()
print("host us/call %.2f, total us/call %.2f" % ((t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))
