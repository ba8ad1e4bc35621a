"""Per-GPU matching time when list1 is sharded N ways against the whole list2 (the strong-scaling
regime of config 5): n1 = 200000 / N queries x 200000 rows, device-resident."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stereo_amd import api, synth

ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 200000
d1, s1, d2, s2 = synth.descriptor_sets(42, n, n)
t = [torch.from_numpy(a).cuda() for a in (d1, s1, d2, s2)]
base = None
only = [int(a) for a in sys.argv[1:]]
for N in (only or (1, 2, 4, 8)):
    n1 = n // N
    o = (torch.empty(n1, dtype=torch.int32, device="cuda"), torch.empty(n1, dtype=torch.int32, device="cuda"),
         torch.empty(n1, dtype=torch.float32, device="cuda"), torch.empty(n1, dtype=torch.float32, device="cuda"),
         torch.empty(n1, dtype=torch.uint8, device="cuda"))
    f = lambda: ctx.match_dev(t[0][:n1], t[1][:n1], t[2], t[3], 1000.0, *o)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    base = base or ms
    print("N = %d: %6d queries per GPU  %.3f ms  (ideal %.3f ms, efficiency %.2f)  %s"
          % (N, n1, ms, base / N, base / N / ms, ctx.match_stats()))
