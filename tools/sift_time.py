"""Times stx_sift_descriptors_dev on ~12k features of a 2880x1988 gradient pair."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.init()
from stereo_amd import api

ctx = api.Context()
H, W, n = 1988, 2880, 12238
rng = np.random.RandomState(12)
gx, gy = [np.clip(rng.randn(H, W) * 60, 0, 255).astype(np.uint8) for _ in range(2)]
xy = np.stack([rng.randint(0, W, n), rng.randint(0, H, n)], axis=1).astype(np.float32)
t_gx, t_gy, t_xy = torch.from_numpy(gx).cuda(), torch.from_numpy(gy).cuda(), torch.from_numpy(xy).cuda()
ang = torch.zeros(n, dtype=torch.float32, device="cuda")
desc = torch.empty((n, 128), dtype=torch.float32, device="cuda")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    ctx.sift_descriptors_dev(t_gx, t_gy, t_xy, ang, desc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ctx.sift_descriptors_dev(t_gx, t_gy, t_xy, ang, desc)
e1.record()
torch.cuda.synchronize()
print("%.4f ms per call, %d features" % (e0.elapsed_time(e1) / 20, n))
