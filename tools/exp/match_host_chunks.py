"""Host-pointer 200k x 200k matching with list1 in one / two (25 % + 75 %) / four query chunks
(STX_MATCH_CHUNKS, read per call), pageable and page-locked arrays.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stereo_amd import api, synth

n = 200_000
d1, s1, d2, s2 = synth.descriptor_sets(5, n, n)
ctx = api.Context(0)
pin = [torch.from_numpy(a).pin_memory().numpy() for a in (d1, s1, d2, s2)]
ref = None
for mode in ("1", "2", "4", "2", "1", "4"):
    os.environ["STX_MATCH_CHUNKS"] = mode
    for name, arrs in (("pageable", (d1, s1, d2, s2)), ("page-locked", pin)):
        for _ in range(2):
            out = ctx.match(*arrs, 1000.0)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); out = ctx.match(*arrs, 1000.0); ts.append((time.perf_counter() - t0) * 1e3)
        if ref is None:
            ref = out
        same = all(a.tobytes() == b.tobytes() for a, b in zip(out, ref))
        print("chunks %s  %-11s best %.2f median %.2f ms  same %s" % (mode, name, min(ts), sorted(ts)[4], same))
