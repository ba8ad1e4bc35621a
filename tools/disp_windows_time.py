"""Times the sliding-sum disparity kernel per window size on the headline pair (2880x1988, D=256)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.init()
from stereo_amd import api, synth

ctx = api.Context()
H, W, D = 1988, 2880, 256
l, r = synth.rectified_pair(2024, H, W, D)
a, b = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for cm, name in ((api.COST_SAD, "SAD"), (api.COST_SSD, "SSD")):
    for win in (9, 11, 13, 15, 17, 21, 31):
        for _ in range(2):
            ctx.disparity_dev(a, b, out, win=win, max_disp=D, cost_mode=cm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            ctx.disparity_dev(a, b, out, win=win, max_disp=D, cost_mode=cm)
        e1.record()
        torch.cuda.synchronize()
        print("%s %2dx%-2d  %.3f ms" % (name, win, win, e0.elapsed_time(e1) / n), flush=True)
