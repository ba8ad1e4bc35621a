"""Times of the two kernels of hypothesis scoring at 200 x 200 000 (HIP events, best of 5 rounds of 10
calls): the whole stx_ransac_score_dev call, and stx_inlier_sum_dev alone on the errors it produced."""
import sys
import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereo_amd
from stereo_amd import synth, ransac


def hypotheses(p1, p2, k1, k2, fm, nh, sample):
    rng = np.random.RandomState(1)
    recs, excl = [], []
    while len(recs) < nh:
        ch = rng.choice(sample, 8, replace=False).astype(np.int32)
        ok, f8 = ransac.find_fundamental_matrix(p1[ch], p2[ch])
        rec = ransac.hypothesis_record(f8, k1, k2) if ok else None
        if rec is not None:
            recs.append(rec)
            excl.append(ch)
    return np.stack(recs), np.array(excl, np.int32)


def timed(fn, rounds=5, calls=10):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(calls):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / calls)
    return best


def main():
    n, nh = 200_000, 200
    k1, k2, p1, p2, fm = synth.two_view_matches(55, n)
    recs, excl = hypotheses(p1[:4000], p2[:4000], k1, k2, fm, nh, 4000)
    dev = torch.device("cuda", 0)
    ctx = stereo_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    inl = torch.zeros(nh, dtype=torch.int32, device=dev)
    tot = torch.zeros(nh, dtype=torch.float32, device=dev)
    err = torch.zeros((nh, n), dtype=torch.float32, device=dev)
    for _ in range(30):
        ctx.ransac_score_dev(recs, k1, k2, d1, d2, excl, 70.0, inl, tot, err)
    torch.cuda.synchronize()
    print("stx_ransac_score_dev 200 x 200k: %.3f ms" % timed(lambda: ctx.ransac_score_dev(recs, k1, k2, d1, d2, excl, 70.0, inl, tot, err)))
    print("stx_inlier_sum_dev alone:        %.3f ms  (inliers %d..%d of %d)" % (
        timed(lambda: ctx.inlier_sum_dev(err, 70.0, inl, tot)), int(inl.min()), int(inl.max()), n))
    for thr in (0.0, 1.0, 1e9):
        print("  threshold %g: %.3f ms" % (thr, timed(lambda: ctx.inlier_sum_dev(err, thr, inl, tot))))


if __name__ == "__main__":
    main()
