"""Times stx_ransac_score_dev at config 5's scale (200 hypotheses x 200k pairs); run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.init()
from stereo_amd import api, ransac, synth

ctx = api.Context()
n, nh = 200_000, 200
k1, k2, p1, p2, fm = synth.two_view_matches(77, n)
rng = ransac.msvc_rand(1)
recs, excl = [], []
while len(recs) < nh:
    ch = ransac.draw_eight(4000, rng)
    ok, f8 = ransac.find_fundamental_matrix(p1[ch], p2[ch])
    rec = ransac.hypothesis_record(f8, k1, k2) if ok else None
    if rec is not None:
        recs.append(rec)
        excl.append(ch)
recs, excl = np.stack(recs), np.array(excl, np.int32)
d1, d2 = torch.from_numpy(p1).cuda(), torch.from_numpy(p2).cuda()
inl = torch.zeros(nh, dtype=torch.int32, device="cuda")
tot = torch.zeros(nh, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.ransac_score_dev(recs, k1, k2, d1, d2, excl, float(os.environ.get("RS_THRESHOLD", "70")), inl, tot)
    ctx.synchronize()
    print("ms per call: %.3f" % ((time.perf_counter() - t0) * 100), "inliers", np.sort(inl.cpu().numpy())[-3:])
