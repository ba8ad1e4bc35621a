"""Where the 200k x 200k host-pointer matching call spends its time: device-only in one launch and
in four query chunks, host arrays pageable and page-locked.  Run on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import stereo_amd
from stereo_amd import api
from stereo_amd import synth

n = 200_000
d1, s1, d2, s2 = synth.descriptor_sets(5, n, n)
dev = torch.device("cuda", 0)
ctx = api.Context(0)
t = [torch.from_numpy(a).to(dev) for a in (d1, s1, d2, s2)]
outs = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
        torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
        torch.empty(n, dtype=torch.uint8, device=dev))

def best(f, reps=12):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]

def one():
    ctx.match_dev(t[0], t[1], t[2], t[3], 1000.0, *outs)

def chunks(qc):
    for q0 in range(0, n, qc):
        q1 = min(n, q0 + qc)
        ctx.match_dev(t[0][q0:q1], t[1][q0:q1], t[2], t[3], 1000.0, *[o[q0:q1] for o in outs])

four = lambda: chunks(50_176)
two = lambda: chunks(100_352)
three = lambda: chunks(67_072)
for _ in range(3): one(); four(); two(); three()
print("device, one launch:        best %.2f median %.2f ms" % best(one))
print("device, two query chunks:  best %.2f median %.2f ms" % best(two))
print("device, three query chunks: best %.2f median %.2f ms" % best(three))
print("device, four query chunks: best %.2f median %.2f ms" % best(four))
for _ in range(3): ctx.match(d1, s1, d2, s2, 1000.0)
print("host, pageable arrays:     best %.2f median %.2f ms" % best(lambda: ctx.match(d1, s1, d2, s2, 1000.0)))
pin = [torch.from_numpy(a).pin_memory().numpy() for a in (d1, s1, d2, s2)]
for _ in range(3): ctx.match(*pin, 1000.0)
print("host, page-locked arrays:  best %.2f median %.2f ms" % best(lambda: ctx.match(*pin, 1000.0)))
h = torch.from_numpy(d1).pin_memory()
def up():
    t[0].copy_(h, non_blocking=True)
print("one list over the bus (page-locked, 102 MB): best %.2f median %.2f ms" % best(up))
