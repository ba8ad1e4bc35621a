"""Runs the kernels bench.py times, one workload after the other in a fixed order, so that a
rocprofv3 --pmc pass over THIS script can be cut into per-workload counters by
tools/pmc_collect.py (same inputs, seeds and sizes as bench.py).  Writes the order it ran in to
gpurun_out/pmc_r03/workloads.json (PMC_OUT overrides the directory).

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -o p -- python3 tools/pmc_workloads.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import stereo_amd
from stereo_amd import api, synth

CALLS = 3
torch.cuda.set_device(0)
ctx = stereo_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
order = []   # (key, kernel name substring, kernel launches per call, calls)


def run(key, kernel, per_call, fn, calls=CALLS):
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    order.append({"key": key, "kernel": kernel, "launches_per_call": per_call, "calls": calls})


W, H, D, WIN = 2880, 1988, 256, 9
left, right = synth.rectified_pair(2024, H, W, D)
d_l, d_r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
d_o = torch.empty_like(d_l)
run("disp_band_9x9_sad_d256_2880x1988", "disp_band_kernel<4, false>", 1,
    lambda: ctx.disparity_dev(d_l, d_r, d_o, win=WIN, max_disp=D))
run("disp_ref_2880x1988", "disp_ref_kernel", 1, lambda: ctx.disparity_dev(d_l, d_r, d_o, win=1, max_disp=0))
run("disp_band_1x1_d256_2880x1988", "disp_band_kernel<0, false>", 1,
    lambda: ctx.disparity_dev(d_l, d_r, d_o, win=1, max_disp=D))
run("disp_band_9x9_ssd_d256_2880x1988", "disp_band_kernel<4, true>", 1,
    lambda: ctx.disparity_dev(d_l, d_r, d_o, win=WIN, max_disp=D, cost_mode=api.COST_SSD))
run("disp_band_21x21_sad_d256_2880x1988", "disp_band_kernel<10, false>", 1,
    lambda: ctx.disparity_dev(d_l, d_r, d_o, win=21, max_disp=D))
run("disp_band_31x31_ssd_d256_2880x1988", "disp_band_kernel<15, true>", 1,
    lambda: ctx.disparity_dev(d_l, d_r, d_o, win=31, max_disp=D, cost_mode=api.COST_SSD))
w3, h3, D3 = 3840, 2160, 512
l3, r3 = synth.rectified_pair(303, h3, w3, D3)
t_l3, t_r3 = torch.from_numpy(l3).cuda(), torch.from_numpy(r3).cuda()
t_o3 = torch.empty((h3, w3), dtype=torch.int16, device="cuda")
run("disp_band_9x9_sad_d512_3840x2160", "disp_band_kernel<4, false>", 2,
    lambda: ctx.disparity_dev(t_l3, t_r3, t_o3, win=WIN, max_disp=D3))
del t_l3, t_r3, t_o3
# BASELINE config 4: 1024 pairs of 1080p, 192 disparities, one batched launch (four distinct pairs repeated)
w4, h4, D4, n4, distinct = 1920, 1080, 192, 1024, 4
p4 = [synth.rectified_pair(400 + k, h4, w4, D4) for k in range(distinct)]
d_l4 = torch.from_numpy(np.stack([p[0] for p in p4])).cuda().repeat(n4 // distinct, 1, 1)
d_r4 = torch.from_numpy(np.stack([p[1] for p in p4])).cuda().repeat(n4 // distinct, 1, 1)
t_o4 = torch.empty_like(d_l4)
run("disp_band_9x9_sad_d192_1024x1080p", "disp_band_kernel<4, false>", 1,
    lambda: ctx.disparity_dev(d_l4, d_r4, t_o4, win=WIN, max_disp=D4), calls=2)
del d_l4, d_r4, t_o4
torch.cuda.empty_cache()
nimg, rh, rw = 8, 1080, 1920
src = torch.from_numpy(np.stack([synth.texture(70 + k, rh, rw) for k in range(2)])).cuda().repeat(nimg // 2, 1, 1).contiguous()
dstt = torch.zeros_like(src)
Hinv = np.stack([ctx.invert_homography(synth.homography(80 + k, rw, rh)).reshape(9) for k in range(nimg)])
run("rectify_8x1080p", "rectify_kernel<true>", 1, lambda: ctx.rectify_dev(src, dstt, Hinv, api.RECT_DST_IS_ZERO))
xy = torch.empty((W * H // 4, 2), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
run("fast12_2880x1988", "fast_flags_kernel", 1, lambda: ctx.fast_detect_dev(d_l, xy, cnt, api.FAST_THRESHOLD))
order.append({"key": "fast12_scatter_2880x1988", "kernel": "fast_scatter_kernel", "launches_per_call": 1, "calls": CALLS})
from stereo_amd import ransac as rsc
rn, rh_ = 200_000, api.FUNDAMENTAL_RANSAC_ITERATIONS
rk1, rk2, rp1, rp2, rfm = synth.two_view_matches(77, rn)
rrng = rsc.msvc_rand(1)
recs, excl = [], []
while len(recs) < rh_:
    ch = rsc.draw_eight(4000, rrng)
    okf, f8 = rsc.find_fundamental_matrix(rp1[ch], rp2[ch])
    rec = rsc.hypothesis_record(f8, rk1, rk2) if okf else None
    if rec is not None:
        recs.append(rec)
        excl.append(ch)
recs, excl = np.stack(recs), np.array(excl, np.int32)
t_p1, t_p2 = torch.from_numpy(rp1).cuda(), torch.from_numpy(rp2).cuda()
r_inl = torch.zeros(rh_, dtype=torch.int32, device="cuda")
r_sum = torch.zeros(rh_, dtype=torch.float32, device="cuda")
run("ransac_error_200x200k", "ransac_error_kernel", 1,
    lambda: ctx.ransac_score_dev(recs, rk1, rk2, t_p1, t_p2, excl, 70.0, r_inl, r_sum))
order.append({"key": "ransac_reduce_200x200k", "kernel": "ransac_reduce_kernel", "launches_per_call": 1, "calls": CALLS})
n = 200000
d1, s1, d2, s2 = synth.descriptor_sets(42, n, n)
t = [torch.from_numpy(a).cuda() for a in (d1, s1, d2, s2)]
oc = torch.empty(n, dtype=torch.int32, device="cuda"); os_ = torch.empty_like(oc)
om = torch.empty(n, dtype=torch.float32, device="cuda"); od = torch.empty_like(om)
ot = torch.empty(n, dtype=torch.uint8, device="cuda")
run("match_x1_prefilter_200k", "match_mfma_bf16x1_kernel", 1,
    lambda: ctx.match_dev(t[0], t[1], t[2], t[3], 1000.0, oc, os_, om, od, ot))
order.append({"key": "match_replay_200k", "kernel": "match_replay_kernel", "launches_per_call": 1, "calls": CALLS})
order.append({"key": "match_prep_200k", "kernel": "match_prep_x1_kernel", "launches_per_call": 1, "calls": CALLS})
order.append({"key": "match_warm_200k", "kernel": "match_warm_kernel", "launches_per_call": 1, "calls": CALLS})
order.append({"key": "match_overflow_scan_200k", "kernel": "match_exact_seg_kernel", "launches_per_call": 1, "calls": CALLS})
# the same size on descriptors with repeated structure (near-ties, overflowing queries)
c1, cs1, c2, cs2 = synth.descriptor_sets_clustered(77, n, n)
tc = [torch.from_numpy(a).cuda() for a in (c1, cs1, c2, cs2)]
run("match_x1_prefilter_200k_clustered", "match_mfma_bf16x1_kernel", 1,
    lambda: ctx.match_dev(tc[0], tc[1], tc[2], tc[3], 1000.0, oc, os_, om, od, ot))
order.append({"key": "match_replay_200k_clustered", "kernel": "match_replay_kernel", "launches_per_call": 1, "calls": CALLS})
order.append({"key": "match_overflow_scan_200k_clustered_none_overflows", "kernel": "match_exact_seg_kernel", "launches_per_call": 1, "calls": CALLS})
# ... and with the candidate sub-lists held to 128 slots, so that the two large clusters overflow
# and the overflow pass has its 1 223 queries to scan
os.environ["STX_MATCH_SUBCAP"] = "128"
run("match_x1_prefilter_200k_clustered_128slots", "match_mfma_bf16x1_kernel", 1,
    lambda: ctx.match_dev(tc[0], tc[1], tc[2], tc[3], 1000.0, oc, os_, om, od, ot))
order.append({"key": "match_replay_200k_clustered_128slots", "kernel": "match_replay_kernel", "launches_per_call": 1, "calls": CALLS})
order.append({"key": "match_overflow_scan_200k_clustered", "kernel": "match_exact_seg_kernel", "launches_per_call": 1, "calls": CALLS})
del os.environ["STX_MATCH_SUBCAP"]
out = os.environ.get("PMC_OUT") or os.path.join(ROOT, "gpurun_out", "pmc_r03")
os.makedirs(out, exist_ok=True)
json.dump(order, open(os.path.join(out, "workloads.json"), "w"), indent=1)
print("ran", len(order), "workloads")
