"""Device-resident matching time by problem size for the small-problem kernel and the MFMA prefilter:
where STX_MATCH_AUTO should switch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stereo_amd import api, synth

ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for n1, n2 in ((32, 32), (64, 64), (100, 100), (171, 210), (256, 256), (400, 400), (100, 2000), (2000, 100), (64, 5000), (5000, 64), (500, 500), (1000, 1000), (2000, 2000), (3000, 3000), (5000, 5000), (8000, 8000), (12000, 12000), (20000, 20000),
               (200, 20000), (20000, 1100), (50000, 2000)):
    d1, s1, d2, s2 = synth.descriptor_sets(n1 + n2, n1, n2)
    t = [torch.from_numpy(a).cuda() for a in (d1, s1, d2, s2)]
    o = (torch.empty(n1, dtype=torch.int32, device="cuda"), torch.empty(n1, dtype=torch.int32, device="cuda"),
         torch.empty(n1, dtype=torch.float32, device="cuda"), torch.empty(n1, dtype=torch.float32, device="cuda"),
         torch.empty(n1, dtype=torch.uint8, device="cuda"))
    row = "%6d x %6d:" % (n1, n2)
    for name, algo in (("small", api.MATCH_SMALL), ("x1", api.MATCH_MFMA_BF16X1)):
        ctx.set_match_algo(algo)
        f = lambda: ctx.match_dev(t[0], t[1], t[2], t[3], 1000.0, *o)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        row += "  %s %8.3f ms" % (name, e0.elapsed_time(e1) / 10)
    print(row)
ctx.set_match_algo(api.MATCH_AUTO)
