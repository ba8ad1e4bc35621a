"""Matching time for given (n1, n2) pairs, device-resident: python tools/exp/match_shape_time.py n1:n2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stereo_amd import api, synth

ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 200000
d1, s1, d2, s2 = synth.descriptor_sets(42, n, n)
t = [torch.from_numpy(a).cuda() for a in (d1, s1, d2, s2)]
for arg in sys.argv[1:]:
    n1, n2 = (int(x) for x in arg.split(":"))
    o = (torch.empty(n1, dtype=torch.int32, device="cuda"), torch.empty(n1, dtype=torch.int32, device="cuda"),
         torch.empty(n1, dtype=torch.float32, device="cuda"), torch.empty(n1, dtype=torch.float32, device="cuda"),
         torch.empty(n1, dtype=torch.uint8, device="cuda"))
    f = lambda: ctx.match_dev(t[0][:n1], t[1][:n1], t[2][:n2], t[3][:n2], 1000.0, *o)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    st = ctx.match_stats()
    print("%7d x %7d  %.3f ms  cand/query %.1f" % (n1, n2, e0.elapsed_time(e1) / 10, st["candidates_per_query"]))
