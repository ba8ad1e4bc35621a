"""Run the FAST-12 path a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_amd
from stereo_amd import synth, api

W, H = 2880, 1988
left, _ = synth.rectified_pair(2024, H, W, 256)
img = torch.from_numpy(left).cuda()
ctx = stereo_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
xy = torch.empty((W * H // 4, 2), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
for _ in range(20):
    ctx.fast_detect_dev(img, xy, cnt, api.FAST_THRESHOLD)
torch.cuda.synchronize()
print("corners", int(cnt.item()))
