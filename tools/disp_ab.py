"""A/B timing of two builds of libstereo_hip.so on the headline disparity workload, alternating
within one process run each (STEREO_HIP_LIB selects the build)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
torch.cuda.init()
from stereo_amd import api, synth
ctx = api.Context()
H, W, D = 1988, 2880, 256
l, r = synth.rectified_pair(2024, H, W, D)
a, b = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    ctx.disparity_dev(a, b, out, win=9, max_disp=D)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ctx.disparity_dev(a, b, out, win=9, max_disp=D)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50)
print("%%.4f ms" %% best)
''' % ROOT
for rnd in range(3):
    for name in sys.argv[1:]:
        env = dict(os.environ, STEREO_HIP_LIB=os.path.join(ROOT, name))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        print(rnd, name, out.stdout.strip(), out.stderr.strip()[-200:], flush=True)
