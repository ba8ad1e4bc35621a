"""Time the 200k x 200k matching call (whole call, HIP events).  STEREO_HIP_LIB selects the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stereo_amd
from stereo_amd import synth, api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
clustered = len(sys.argv) > 2 and sys.argv[2] == "clustered"
d1, s1, d2, s2 = synth.descriptor_sets_clustered(77, n, n) if clustered else synth.descriptor_sets(42, n, n)
ctx = stereo_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
t = [torch.from_numpy(a).cuda() for a in (d1, s1, d2, s2)]
oc = torch.empty(n, dtype=torch.int32, device="cuda"); os_ = torch.empty_like(oc)
om = torch.empty(n, dtype=torch.float32, device="cuda"); od = torch.empty_like(om)
ot = torch.empty(n, dtype=torch.uint8, device="cuda")
f = lambda: ctx.match_dev(t[0], t[1], t[2], t[3], 1000.0, oc, os_, om, od, ot)
for _ in range(3): f()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get("STEREO_HIP_LIB", "default")), "ms/call %.3f" % (e0.elapsed_time(e1) / 10),
      "accepted", int((ot == 3).sum().item()), ctx.match_stats())
