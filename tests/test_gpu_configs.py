"""BASELINE.json's configurations at their own sizes, the host-pointer data path, and the
multi-device handle -- all through the C ABI on a real MI355X.

config 1  features -> F -> rectify -> disparity as ONE chain (741x500 synthetic stand-in for the
          Middlebury Motorcycle quarter-res pair, which is not in the image)
config 4  the real launch: 1024 1080p pairs, 192 disparities, one batched call
config 5  200 000 x 200 000 descriptors at full size
(config 2 and config 3 at full size are in test_gpu_parity.py.)
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from stereo_amd import api, synth

pytestmark = pytest.mark.gpu


def _match_equal(got, want):
    names = ["closest", "second", "min_dist", "second_dist", "status"]
    for n, g, w in zip(names, got, want):
        g, w = np.asarray(g), np.asarray(w)
        if g.dtype == np.float32:
            ok = (g.view(np.uint32) == w.view(np.uint32)) | (np.isnan(g) & np.isnan(w))
            assert ok.all(), "%s differs at %s" % (n, np.nonzero(~ok)[0][:8])
        else:
            assert g.tobytes() == w.tobytes(), "%s differs at %s" % (n, np.nonzero(g != w)[0][:8])


# ------------------------------------------------------------------ config 5 ----

def _config5_run(ctx, oracle, d1, s1, d2, s2, sample):
    """The 200k x 200k x1 launch itself: all five outputs of `sample` queries against the exact GPU
    scan of those queries over the whole list2, 64 of them against the CPU oracle.  Returns the
    outputs (numpy), the device tensors, the candidate statistics and the time of a second call."""
    import torch
    n = len(d1)
    dev = torch.device("cuda", 0)
    td1, ts1 = torch.from_numpy(d1).to(dev), torch.from_numpy(s1).to(dev)
    td2, ts2 = torch.from_numpy(d2).to(dev), torch.from_numpy(s2).to(dev)

    def outs(m):
        return (torch.empty(m, dtype=torch.int32, device=dev), torch.empty(m, dtype=torch.int32, device=dev),
                torch.empty(m, dtype=torch.float32, device=dev), torch.empty(m, dtype=torch.float32, device=dev),
                torch.empty(m, dtype=torch.uint8, device=dev))

    full = outs(n)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_match_algo(api.MATCH_MFMA_BF16X1)
        ctx.match_dev(td1, ts1, td2, ts2, 1000.0, *full)
        torch.cuda.synchronize()
        stats = ctx.match_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.match_dev(td1, ts1, td2, ts2, 1000.0, *full)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tidx = torch.from_numpy(sample).to(dev)
        sd1, ss1 = td1[tidx].contiguous(), ts1[tidx].contiguous()
        ex = outs(len(sample))
        ctx.set_match_algo(api.MATCH_EXACT)
        ctx.match_dev(sd1, ss1, td2, ts2, 1000.0, *ex)
        torch.cuda.synchronize()
    finally:
        ctx.set_match_algo(api.MATCH_AUTO)
        ctx.set_stream(0)
    got = [t.cpu().numpy() for t in full]
    want = [t.cpu().numpy() for t in ex]
    _match_equal([g[sample] for g in got], want)
    sub = sample[:: max(1, len(sample) // 64)][:64]
    _match_equal([g[sub] for g in got], oracle.match(d1[sub], s1[sub], d2, s2, 1000.0))
    return got, (td1, ts1, td2, ts2, full, outs), stats, ms


def test_config5_full_size_200k_x_200k(ctx, oracle):
    """BASELINE config 5 at its own size.  The x1 prefilter's schedule depends on the size (pieces
    per query block, padding to 32 tiles, XCD-aligned ranges), so the 200k x 200k launch itself is
    checked: all five outputs of a strided sample of 2048 queries against the exact GPU scan of
    those queries over the whole list2, and 64 of them against the CPU oracle."""
    import torch
    n = 200_000
    d1, s1, d2, s2 = synth.descriptor_sets(42, n, n)
    idx = np.unique(np.concatenate([np.arange(0, n, 97)[:2040], [0, 1, 511, 512, 513, n - 513, n - 2, n - 1]]))
    assert len(idx) >= 2000
    got, (td1, ts1, td2, ts2, full, outs), stats, ms = _config5_run(ctx, oracle, d1, s1, d2, s2, idx)
    # The workload (synth.descriptor_sets): half of list2's rows are noisy copies of
    # list1 rows drawn WITH replacement, scores are independent U[700, 5000] draws, so the score gate
    # (|s1 - s2| > 1000, Features.cpp:1057) comes first and rejects 59 % of the queries; 27 % fail
    # the ratio test, 14 % are accepted, a few have their nearest row at j = 0 and fail.
    st = got[4]
    acc = int((st == api.MATCH_ACCEPTED).sum())
    assert 0.10 * n < acc < 0.20 * n, acc
    assert 0.5 * n < int((st == api.MATCH_SCORE).sum()) < 0.7 * n
    assert 0.2 * n < int((st == api.MATCH_RATIO).sum()) < 0.35 * n
    # the replay's share: a few tens of candidates per query, nobody overflows on this set
    assert stats["queries"] == n and 5 < stats["candidates_per_query"] < 80 and stats["overflow_queries"] == 0, stats
    # AUTO picks the same kernel at this size: same bytes
    auto = outs(n)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.match_dev(td1, ts1, td2, ts2, 1000.0, *auto)
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(0)
    for a, b in zip(auto, full):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    _config5_times["synthetic"] = ms


_config5_times = {}


def test_config5_clustered_200k_x_200k(ctx, oracle, monkeypatch):
    """Config 5's size on descriptors with repeated structure: 2 000 centres with
    ~100 near-duplicates each at noise 1e-3, and two centres with ~500 each.  Near-ties pile up in the
    running-minimum scan: ~120 candidates per query instead of ~30.  With the sub-lists at their default
    size (512 slots at this shape) even the two large clusters stay in their lists; held to 128 slots
    (round 3's first version) their ~1 000 queries overflow and take the overflow pass, the exact scan
    cut along list2.  Bit-exact both ways on a sample that holds queries of the large clusters and
    ordinary ones; the call stays within 2x (3x with the overflow pass) of the ordinary set's time --
    it was a cliff: one lane per overflowing query."""
    n = 200_000
    d1, s1, d2, s2, w1, _ = synth.descriptor_sets_clustered(77, n, n, with_clusters=True)
    hot = np.nonzero(w1 < 2)[0]                      # queries of the two large clusters (and their small share)
    assert len(hot) > 800
    idx = np.unique(np.concatenate([np.arange(0, n, 131)[:1500], hot[:200], [0, 1, 511, 512, n - 1]]))
    base = _config5_times.get("synthetic")
    got, _, stats, ms = _config5_run(ctx, oracle, d1, s1, d2, s2, idx)
    assert stats["queries"] == n and stats["overflow_queries"] == 0, stats
    assert stats["candidates_per_query"] > 60 and stats["candidates_max"] > 400, stats
    print("clustered 200k x 200k: %.2f ms, %s" % (ms, stats))
    if base is not None:
        assert ms < 2.0 * base, (ms, base)
    assert ms < 30.0, ms
    monkeypatch.setenv("STX_MATCH_SUBCAP", "128")
    got2, _, stats2, ms2 = _config5_run(ctx, oracle, d1, s1, d2, s2, idx)
    assert stats2["queries"] == n and stats2["overflow_queries"] > 100, stats2
    print("clustered 200k x 200k, 128-slot sub-lists: %.2f ms, %s" % (ms2, stats2))
    for a_, b_ in zip(got, got2):
        assert np.array_equal(a_, b_, equal_nan=True) if a_.dtype.kind == "f" else np.array_equal(a_, b_)
    if base is not None:
        assert ms2 < 3.0 * base, (ms2, base)
    assert ms2 < 40.0, ms2


# ------------------------------------------------------------------ config 4 ----

def test_config4_real_launch_1024_pairs(ctx, oracle):
    """BASELINE config 4's actual launch: 1024 pairs in one batched call (grid z = 1024, 2 GB
    per input array), four distinct pairs repeated as in bench.py.  Every one of the 1024 maps
    must equal the single-call map of its pair (compared on the device); the four single-call maps are
    checked against the oracle in full."""
    import torch
    h, w, D, win, n, distinct = 1080, 1920, 192, 9, 1024, 4
    pairs = [synth.rectified_pair(400 + k, h, w, D) for k in range(distinct)]
    dev = torch.device("cuda", 0)
    a4 = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    b4 = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    a = a4.repeat(n // distinct, 1, 1)
    b = b4.repeat(n // distinct, 1, 1)
    out = torch.empty_like(a)
    single = torch.empty_like(a4)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.disparity_dev(a, b, out, win=win, max_disp=D)
        for k in range(distinct):
            ctx.disparity_dev(a4[k], b4[k], single[k], win=win, max_disp=D)
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(0)
    same = (out.view(n // distinct, distinct, h, w) == single.unsqueeze(0)).all()
    assert bool(same.item())
    sg = single.cpu().numpy()
    from tests.test_gpu_parity import _oracle_whole_map
    for k in range(distinct):          # every distinct pair's WHOLE map against the oracle (~1 s each)
        l, r = pairs[k]
        want = _oracle_whole_map(oracle, l, r, win, D)
        assert (sg[k] == want).all(), (k, np.argwhere(sg[k] != want)[:5])


@pytest.mark.parametrize("win,cm", [(21, api.COST_SAD), (31, api.COST_SSD)])
def test_config2_size_wide_windows_two_gpu_kernels_agree(ctx, oracle, win, cm):
    """BASELINE config 2's pair (2880x1988, 256 disparities) with a wide window: the register kernel's
    wide form and the LDS column-sum kernel -- two independent implementations -- give the same map
    over the whole image, and the oracle agrees on three row bands."""
    import torch
    h, w, D = 1988, 2880, 256
    l, r = synth.rectified_pair(2024, h, w, D)
    a, b = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    wide, colsum = torch.empty_like(a), torch.empty_like(a)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.disparity_dev(a, b, wide, win=win, max_disp=D, cost_mode=cm)
        ctx.set_disparity_algo(api.DISP_COLSUM)
        ctx.disparity_dev(a, b, colsum, win=win, max_disp=D, cost_mode=cm)
        torch.cuda.synchronize()
    finally:
        ctx.set_disparity_algo(api.DISP_AUTO)
        ctx.set_stream(0)
    assert bool((wide == colsum).all().item())
    g = wide.cpu().numpy()
    rr = win // 2
    for y0 in (0, 997, h - 3):
        b0, b1 = max(0, y0 - rr), min(h, y0 + 3 + rr)
        want = oracle.disparity_ext(l[b0:b1], r[b0:b1], win, D, cm, sliding=True)
        assert (g[y0:y0 + 3] == want[y0 - b0:y0 - b0 + 3]).all()


# ------------------------------------------------------------------ config 1 ----

_cv8u_gradients = synth.gradient_stand_ins


def test_config1_whole_chain_on_the_741x500_stand_in(ctx, oracle):
    """BASELINE config 1 as a chain (main.cpp:188-380): FAST -> score + NMS -> descriptors for both
    images -> MatchDescriptors -> FindFundamentalMatrixWithRANSAC -> RectifyImage x2 ->
    ComputeDepthImage, every stage on the GPU and every stage compared with the same chain through
    the oracle.  Stand-ins, stated: the image pair is synthetic (741x500, the Motorcycle
    quarter-res size; the dataset is not in the image); the gradient images come from
    _cv8u_gradients instead of OpenCV; ComputeRectificationRotations (Eigen BDCSVD + SO(3) log/exp,
    host code that stays the reference's) is replaced by a fixed seeded pair of K R K^-1."""
    from stereo_amd import ransac
    h, w = 500, 741
    left, right = synth.rectified_pair(1001, h, w, 48, zero_frac=0.0)
    stages = []
    for img in (left, right):
        gx, gy = _cv8u_gradients(img)
        feats = []
        assert api.FindFASTFeatures(img, feats, ctx=ctx)
        oxy = oracle.fast_features(img)
        assert [f.p for f in feats] == [tuple(map(float, p)) for p in oxy]
        feats = api.ScoreAndClusterFeatures(gx, gy, feats, api.ST_THRESH, api.NMS_WINDOW, ctx=ctx)
        sxy, ssc, _ = oracle.score_and_cluster(gx, gy, oxy, oracle.gaussian_kernel(3, 1.0), 700.0, 2.0)
        assert [f.p for f in feats] == [tuple(map(float, p)) for p in sxy]
        assert np.array([f.score for f in feats], np.float32).tobytes() == ssc.tobytes()
        descs = []
        assert api.CreateSIFTDescriptors(gx, gy, feats, descs, ctx=ctx)
        oang, odesc = oracle.sift_descriptors(gx, gy, sxy)
        gd = np.stack([f.desc for f in feats])
        assert ((gd.view(np.uint32) == odesc.view(np.uint32)) | (np.isnan(gd) & np.isnan(odesc))).all()
        stages.append((feats, sxy, ssc, odesc))
    n1, n2 = len(stages[0][0]), len(stages[1][0])
    assert n1 > 100 and n2 > 100

    # MatchDescriptors (main.cpp:207)
    matches = api.MatchDescriptors(stages[0][0], stages[1][0], 1000.0, ctx=ctx, out=open(os.devnull, "w"))
    want = oracle.match(stages[0][3], stages[0][2], stages[1][3], stages[1][2], 1000.0)
    acc = np.nonzero(want[4] == api.MATCH_ACCEPTED)[0]
    assert len(matches) == len(acc) and len(acc) >= 30
    p1 = np.array([m[0].p for m in matches], np.float32)
    p2 = np.array([m[1].p for m in matches], np.float32)
    assert p1.tobytes() == stages[0][1][acc].astype(np.float32).tobytes()
    assert p2.tobytes() == stages[1][1][want[0][acc]].astype(np.float32).tobytes()
    assert np.array([m[0].distFromBestMatch for m in matches], np.float32).tobytes() == want[2][acc].tobytes()

    # FindFundamentalMatrixWithRANSAC (main.cpp:220): GPU-scored loop vs oracle-scored loop
    f = 3979.911 / 4
    K = np.array([[f, 0, 311.0], [0, f, 255.0], [0, 0, 1]], np.float32)   # Middlebury calib / 4

    class OracleScorer:
        def ransac_score(self, recs, a, b, q1, q2, exclude=None):
            return oracle.ransac_score(recs, a, b, q1, q2, exclude)[:2]

    ok_g, f_g = ransac.FindFundamentalMatrixWithRANSAC(p1, p2, K, K, seed=7, ctx=ctx)
    ok_o, f_o = ransac.FindFundamentalMatrixWithRANSAC(p1, p2, K, K, seed=7, ctx=OracleScorer())
    assert ok_g == ok_o
    if ok_g:
        assert np.array_equal(f_g, f_o)

    # RectifyImage x2 + ComputeDepthImage (main.cpp:362-380), reference mode and a windowed scan
    H0 = synth.homography(5, w, h, deg_y=0.8, deg_x=0.2)
    H1 = synth.homography(6, w, h, deg_y=-0.6, deg_x=0.3)
    i0, i1 = oracle.invert3x3(H0), oracle.invert3x3(H1)
    assert ctx.invert_homography(H0).tobytes() == i0.astype(np.float32).tobytes()
    r0 = oracle.rectify(left, np.zeros((h, w), np.uint8), i0)
    r1 = oracle.rectify(right, np.zeros((h, w), np.uint8), i1)
    g0, g1 = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
    api.RectifyImage(left, g0, H0, ctx=ctx)
    api.RectifyImage(right, g1, H1, ctx=ctx)
    assert (g0 == r0).all() and (g1 == r1).all()
    depth = api.ComputeDepthImage(g0, g1, ctx=ctx)
    assert (depth == oracle.disparity_ref(r0, r1)).all()
    for win, D in [(1, 0), (9, 64)]:
        wantd = oracle.disparity_ext(r0, r1, win, D, 0)
        got, f0, f1 = ctx.rectify_disparity(left, right, i0, i1, win=win, max_disp=D, want_rectified=True)
        assert (f0 == r0).all() and (f1 == r1).all() and (got == wantd).all()
    assert (depth != 0).any()


# ------------------------------------------------------------- N4 on the GPU ----

def test_descriptor_file_replayed_through_stx_match(ctx, oracle, tmp_path):
    """a descriptor cache file in the reference's text format (Features.h:62-142,
    Features.cpp:1219-1276) is written, read back and its two feature lists matched through
    stx_match; the oracle matches the same read-back values.  (The text keeps 6 significant
    digits, as the reference's operator<< does, so the read-back descriptors are what is matched.)
    No file written by the reference itself exists to pin the format against."""
    from stereo_amd import featfile
    d1, s1, d2, s2 = synth.descriptor_sets(91, 400, 500)
    imgs = []
    for name, d, s in (("im0.png", d1, s1), ("im1.png", d2, s2)):
        feats = [api.Feature(scale=1, p=(float(i % 37) + 0.5, float(i // 37)), score=float(s[i]), angle=float(i % 90),
                             desc=d[i]) for i in range(len(d))]
        imgs.append(featfile.ImageDescriptor(name, 741, 500, np.eye(3, dtype=np.float32),
                                             np.eye(3, dtype=np.float32), feats))
    path = str(tmp_path / "features.txt")
    assert featfile.SaveImageDescriptorsToFile(path, imgs)
    back = featfile.ReadDescriptorsFromFile(path)
    assert len(back) == 2 and len(back[0].features) == 400 and len(back[1].features) == 500
    rd1 = np.stack([f.desc for f in back[0].features])
    rd2 = np.stack([f.desc for f in back[1].features])
    rs1 = np.array([f.score for f in back[0].features], np.float32)
    rs2 = np.array([f.score for f in back[1].features], np.float32)
    assert np.abs(rd1 - d1).max() < 1e-6 and not np.array_equal(rd1, d1)
    want = oracle.match(rd1, rs1, rd2, rs2, 1000.0)
    _match_equal(ctx.match(rd1, rs1, rd2, rs2, 1000.0), want)
    matches = api.MatchDescriptors(back[0].features, back[1].features, 1000.0, ctx=ctx,
                                   out=open(os.devnull, "w"))
    acc = np.nonzero(want[4] == api.MATCH_ACCEPTED)[0]
    assert len(matches) == len(acc) > 50
    assert [m[1].p for m in matches] == [back[1].features[int(want[0][i])].p for i in acc]


def test_match_on_feature_records(ctx, oracle):
    """stx_match_features: the reference's std::vector<Feature> layout (540-byte records, score at
    byte 12, descriptor at byte 20) uploaded as it lies and split on the GPU -- same outputs as
    stx_match on the separate arrays, for the exact scan and the MFMA path, with both pipelines."""
    feature = np.dtype([("scale", np.int32), ("p", np.float32, 2), ("score", np.float32), ("angle", np.float32),
                        ("desc", np.float32, 128), ("distFromBestMatch", np.float32), ("depth", np.float32)])
    assert feature.itemsize == 540 and feature.fields["desc"][1] == 20 and feature.fields["score"][1] == 12
    for n1, n2 in ((70, 90), (3000, 2500), (1, 1), (5, 0)):
        d1, s1, d2, s2 = synth.descriptor_sets(n1 + n2, max(n1, 1), max(n2, 1))
        d2, s2 = d2[:n2], s2[:n2]
        r1, r2 = np.zeros(n1, feature), np.zeros(n2, feature)
        r1["desc"], r1["score"], r1["angle"] = d1, s1, 123.0      # the other fields are noise to the matcher
        r2["desc"], r2["score"], r2["depth"] = d2, s2, -5.0
        want = oracle.match(d1, s1, d2, s2, 1000.0)
        for on in (True, False):
            ctx.set_host_pipeline(on, 0, 0)
            try:
                _match_equal(ctx.match_features(r1, r2, 20, 12, 1000.0), want)
            finally:
                ctx.set_host_pipeline(True, 0, 0)
    # n1 == 0 is a valid no-op, as for the C entry point and the drop-in, also with both lists empty
    r2 = np.zeros(4, feature)
    r2["desc"], r2["score"] = synth.sift_like(3, 4), 900.0
    for empty2 in (r2, np.zeros(0, feature)):
        got = ctx.match_features(np.zeros(0, feature), empty2, 20, 12, 1000.0)
        assert all(len(g) == 0 for g in got)
    r1 = np.zeros(5, feature)
    from stereo_amd import StereoHipError
    with pytest.raises(StereoHipError):
        ctx.match_features(r1, r1, 18, 12)      # misaligned descriptor offset
    with pytest.raises(StereoHipError):
        ctx.match_features(r1, r1, 40, 12)      # descriptor would leave the record


def test_large_host_matching_runs_in_query_chunks(ctx, oracle, monkeypatch):
    """A host-pointer matching call this large runs list1 in two query chunks, a quarter and three
    quarters (the second uploads behind the matching of the first; STX_MATCH_CHUNKS=4 gives round 2's
    four equal ones, =1 none): same outputs as the one-launch device call whichever way, for the
    separate-array and the feature-record entry points; the oracle on a sample."""
    import torch
    n1, n2 = 66_100, 61_000          # n1 * n2 > 4e9 and n1 >= 65536: chunked; ragged last chunk
    d1, s1, d2, s2 = synth.descriptor_sets(5, n1, n2)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(a).to(dev) for a in (d1, s1, d2, s2)]
    outs = (torch.empty(n1, dtype=torch.int32, device=dev), torch.empty(n1, dtype=torch.int32, device=dev),
            torch.empty(n1, dtype=torch.float32, device=dev), torch.empty(n1, dtype=torch.float32, device=dev),
            torch.empty(n1, dtype=torch.uint8, device=dev))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.match_dev(t[0], t[1], t[2], t[3], 1000.0, *outs)
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(0)
    want = [o.cpu().numpy() for o in outs]
    feature = np.dtype([("scale", np.int32), ("p", np.float32, 2), ("score", np.float32), ("angle", np.float32),
                        ("desc", np.float32, 128), ("distFromBestMatch", np.float32), ("depth", np.float32)])
    r1, r2 = np.zeros(n1, feature), np.zeros(n2, feature)
    r1["desc"], r1["score"], r2["desc"], r2["score"] = d1, s1, d2, s2
    for mode in (None, "4", "1"):
        if mode is None:
            monkeypatch.delenv("STX_MATCH_CHUNKS", raising=False)
        else:
            monkeypatch.setenv("STX_MATCH_CHUNKS", mode)
        _match_equal(ctx.match(d1, s1, d2, s2, 1000.0), want)
        _match_equal(ctx.match_features(r1, r2, 20, 12, 1000.0), want)
    idx = np.arange(0, n1, n1 // 40)[:40]
    _match_equal([w_[idx] for w_ in want], oracle.match(d1[idx], s1[idx], d2, s2, 1000.0))


# ------------------------------------------------- the host-pointer data path ----

@pytest.mark.parametrize("threads,chunks", [(1, 1), (4, 0), (3, 8), (8, 3)])
def test_host_pipeline_equals_plain_copies(ctx, oracle, threads, chunks):
    """The pinned-staging / chunked pipeline and the plain-copy path give identical bytes: pitched
    and dense images, reference mode, windowed scans with one and several passes, u16, the warp
    with kept destination pixels, the fused call, matching."""
    rng = np.random.RandomState(threads * 10 + chunks)
    h, w = 333, 1500
    l, r = synth.rectified_pair(3, h, w, 300)
    lp = np.zeros((h, w + 52), np.uint8)[:, :w]
    rp = np.zeros((h, w + 52), np.uint8)[:, :w]
    lp[:], rp[:] = l, r
    src = synth.texture(5, h, w)
    Hinv = oracle.invert3x3(synth.homography(3, w, h))
    dst0 = rng.randint(0, 255, (h, w)).astype(np.uint8)
    d1, s1, d2, s2 = synth.descriptor_sets(9, 3000, 2500)

    def run():
        res = [ctx.disparity(l, r), ctx.disparity(lp, rp, win=9, max_disp=64),
               ctx.disparity(l, r, win=5, max_disp=300, out_dtype=np.uint16),
               ctx.disparity(l[:1], r[:1], win=3, max_disp=8), ctx.disparity(l[:70], r[:70], win=15, max_disp=40)]
        d = dst0.copy()
        ctx.rectify(src, d, Hinv, api.RECT_PRESERVE_DST)
        res.append(d)
        dz = np.zeros((h + 7, w - 9), np.uint8)
        ctx.rectify(src, dz, Hinv, api.RECT_DST_IS_ZERO)
        res.append(dz)
        dzs = np.full((h, w + 40), 9, np.uint8)     # pitched destination: the pad bytes must survive
        ctx.rectify(src, dzs[:, :w], Hinv, api.RECT_DST_IS_ZERO)
        res.append(dzs)
        res.extend(ctx.rectify_disparity(l, r, Hinv, Hinv, win=7, max_disp=32, want_rectified=True))
        res.append(ctx.rectify_disparity(l, r, Hinv, Hinv))
        res.extend(ctx.match(d1, s1, d2, s2, 1000.0))
        return res

    try:
        ctx.set_host_pipeline(False)
        plain = run()
        ctx.set_host_pipeline(True, threads, chunks)
        piped = run()
        again = run()
    finally:
        ctx.set_host_pipeline(True, 0, 0)
    for a, b, c in zip(plain, piped, again):
        assert a.tobytes() == b.tobytes() == c.tobytes()
    assert (plain[0] == oracle.disparity_ref(l, r)).all()
    assert (plain[1] == oracle.disparity_ext(l, r, 9, 64, 0)).all()
    assert (plain[5] == oracle.rectify(src, dst0.copy(), Hinv)).all()


def test_page_locked_caller_buffers_skip_the_staging(ctx, oracle):
    """Dense, page-locked caller arrays (torch pin_memory here; hipHostMalloc / a pinned
    cv::MatAllocator in C++) are read and written by the DMA engines directly; mixed pinned /
    pageable arguments and pinned-but-pitched views fall back per buffer.  Same bytes either way."""
    import torch
    h, w = 400, 1300
    l, r = synth.rectified_pair(31, h, w, 300)
    src = synth.texture(32, h, w)
    Hinv = oracle.invert3x3(synth.homography(5, w, h))

    def pinned(a):
        t = torch.empty(a.shape, dtype=torch.uint8 if a.dtype == np.uint8 else torch.int16, pin_memory=True)
        v = t.numpy().view(a.dtype)
        v[...] = a
        return t, v

    tl, pl = pinned(l)
    tr, pr = pinned(r)
    ts, ps = pinned(src)
    want9 = oracle.disparity_ext(l, r, 9, 64, 0, sliding=True)
    want16 = oracle.disparity_ext(l, r, 5, 300, 0, np.uint16, sliding=True)
    wantr = oracle.rectify(src, np.zeros_like(src), Hinv)
    for on, chunks in ((True, 0), (True, 3), (False, 0)):
        ctx.set_host_pipeline(on, 0, chunks)
        try:
            to, po = pinned(np.zeros((h, w), np.uint8))
            assert (ctx.disparity(pl, pr, win=9, max_disp=64, out=po) == want9).all()
            assert (ctx.disparity(pl, r, win=9, max_disp=64) == want9).all()          # img1 and out pageable
            to16, po16 = pinned(np.zeros((h, w), np.uint16))
            assert (ctx.disparity(pl, pr, win=5, max_disp=300, out_dtype=np.uint16, out=po16) == want16).all()
            assert (ctx.disparity(pl[:, :900], pr[:, :900], win=9, max_disp=64)
                    == oracle.disparity_ext(l[:, :900], r[:, :900], 9, 64, 0, sliding=True)).all()   # pitched views
            td, pd = pinned(np.zeros_like(src))
            ctx.rectify(ps, pd, Hinv, api.RECT_PRESERVE_DST)
            assert (pd == wantr).all()
            pd[...] = 7
            ctx.rectify(ps, pd, Hinv, api.RECT_DST_IS_ZERO)
            assert (pd == wantr).all()
            # a batch through the multi-device handle: pinned inputs and output, then mixed
            from stereo_amd import MultiContext
            batch_l, batch_r = np.stack([l[:200], l[100:300], l[200:400]]), np.stack([r[:200], r[100:300], r[200:400]])
            tbl, pbl = pinned(batch_l)
            tbr, pbr = pinned(batch_r)
            tbo, pbo = pinned(np.zeros_like(batch_l))
            want_b = np.stack([oracle.disparity_ext(batch_l[k], batch_r[k], 9, 64, 0, sliding=True) for k in range(3)])
            mc = MultiContext([0])
            try:
                assert (mc.disparity_batch(pbl, pbr, win=9, max_disp=64, out=pbo) == want_b).all()
                assert (mc.disparity_batch(pbl, batch_r, win=9, max_disp=64) == want_b).all()
            finally:
                mc.close()
            ctx.rectify_keep(0, ps, pd, Hinv)
            ctx.rectify_keep(1, ps, pd, Hinv)
            assert (ctx.disparity_resident(0, 1, (h, w), out=po) == oracle.disparity_ref(wantr, wantr)).all()
        finally:
            ctx.set_host_pipeline(True, 0, 0)


def test_batch_larger_than_one_group(oracle, monkeypatch):
    """A batch that does not fit one group of the host path (256 MB of images and maps; here
    STX_HOST_GROUP_MB=5 holds two 500 x 1400 pairs) runs group after group, the last one ragged:
    same maps as pair-by-pair calls, pageable and page-locked, with and without the pipeline."""
    import stereo_amd
    import torch
    from stereo_amd import MultiContext
    h, w, n = 500, 1400, 5
    pairs = [synth.rectified_pair(60 + k, h, w, 64) for k in range(n)]
    bl, br = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    want = np.stack([oracle.disparity_ext(bl[k], br[k], 9, 64, 0, sliding=True) for k in range(n)])
    pl, pr, po = (torch.empty((n, h, w), dtype=torch.uint8, pin_memory=True) for _ in range(3))
    pl.numpy()[...] = bl
    pr.numpy()[...] = br
    monkeypatch.setenv("STX_HOST_GROUP_MB", "5")
    for pipeline in ("1", "0"):
        monkeypatch.setenv("STX_HOST_PIPELINE", pipeline)
        mc = MultiContext([0])
        try:
            assert (mc.disparity_batch(bl, br, win=9, max_disp=64) == want).all()
            po.numpy()[...] = 7
            assert (mc.disparity_batch(pl.numpy(), pr.numpy(), win=9, max_disp=64, out=po.numpy()) == want).all()
            assert (mc.disparity_batch(pl.numpy(), br, win=9, max_disp=64) == want).all()
        finally:
            mc.close()


def test_no_page_locked_memory_falls_back_to_plain_copies(oracle, monkeypatch):
    """pinned staging is a convenience.  With no page-locked memory to be had
    (STX_PIN_LIMIT_MB makes the staging request fail the way an exhausted host fails it) every
    host-pointer call goes on with plain copies and gives the same bytes; and a matching call on
    page-locked lists asks for no staging for them at all."""
    import stereo_amd
    import torch
    h, w = 300, 1100
    l, r = synth.rectified_pair(51, h, w, 64)
    src = synth.texture(52, h, w)
    Hinv = oracle.invert3x3(synth.homography(5, w, h))
    want = oracle.disparity_ext(l, r, 9, 64, 0, sliding=True)
    wantr = oracle.rectify(src, np.zeros_like(src), Hinv)
    n1, n2 = 3000, 2500
    d1, s1, d2, s2 = synth.descriptor_sets(9, n1, n2)
    wantm = oracle.match(d1, s1, d2, s2, 1000.0)
    feature = np.dtype([("scale", np.int32), ("p", np.float32, 2), ("score", np.float32), ("angle", np.float32),
                        ("desc", np.float32, 128), ("distFromBestMatch", np.float32), ("depth", np.float32)])
    r1, r2 = np.zeros(n1, feature), np.zeros(n2, feature)
    r1["desc"], r1["score"], r2["desc"], r2["score"] = d1, s1, d2, s2
    wl, wr = oracle.rectify(l, np.zeros_like(l), Hinv), oracle.rectify(r, np.zeros_like(r), Hinv)
    wantf = oracle.disparity_ext(wl, wr, 9, 64, 0)
    monkeypatch.setenv("STX_PIN_LIMIT_MB", "0")
    c = stereo_amd.Context(0)
    try:
        for _ in range(2):      # the second round finds the context as the first left it
            assert (c.disparity(l, r, win=9, max_disp=64) == want).all()
            dst = np.zeros_like(src)
            c.rectify(src, dst, Hinv, api.RECT_DST_IS_ZERO)
            assert (dst == wantr).all()
            got, f0, f1 = c.rectify_disparity(l, r, Hinv, Hinv, win=9, max_disp=64, want_rectified=True)
            assert (f0 == wl).all() and (f1 == wr).all() and (got == wantf).all()
            _match_equal(c.match(d1, s1, d2, s2, 1000.0), wantm)
            _match_equal(c.match_features(r1, r2, 20, 12, 1000.0), wantm)
        assert c.staging_bytes() == 0
        # page-locked lists: 1.5 MB each; a 1 MB allowance is enough for the scores and the results
        monkeypatch.setenv("STX_PIN_LIMIT_MB", "1")
        p1 = torch.from_numpy(d1).pin_memory()
        p2 = torch.from_numpy(d2).pin_memory()
        _match_equal(c.match(p1.numpy(), s1, p2.numpy(), s2, 1000.0), wantm)
        got = c.staging_bytes()
        assert 0 < got <= 2 << 20
        # mixed: one list page-locked, the other staged (its staging starts where the first one's would end)
        monkeypatch.delenv("STX_PIN_LIMIT_MB")
        _match_equal(c.match(p1.numpy(), s1, d2, s2, 1000.0), wantm)
        _match_equal(c.match(d1, s1, p2.numpy(), s2, 1000.0), wantm)
        q1 = torch.from_numpy(r1.view(np.uint8).reshape(n1, -1)).pin_memory()
        q2 = torch.from_numpy(r2.view(np.uint8).reshape(n2, -1)).pin_memory()
        f1, f2 = q1.numpy().view(feature).reshape(n1), q2.numpy().view(feature).reshape(n2)
        _match_equal(c.match_features(f1, f2, 20, 12, 1000.0), wantm)
        _match_equal(c.match_features(f1, r2, 20, 12, 1000.0), wantm)
        _match_equal(c.match_features(r1, f2, 20, 12, 1000.0), wantm)
    finally:
        c.close()


def test_page_locked_ranges_and_their_kernel_addresses(ctx, oracle):
    """a caller buffer counts as page-locked only if the WHOLE range lies inside one
    allocation the runtime knows (both ends being host memory is not enough), and kernels store
    into it through its device-visible address, never the raw host pointer.
    stx_host_buffer_kind says how a buffer would be treated; the calls give the oracle's bytes
    whichever way it goes.  hipHostMalloc blocks through ctypes; hipHostRegister as well where the
    box supports it (on this pool it returns hipErrorInvalidValue for any range, so that part
    only runs elsewhere)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")      # the runtime already in the process
    hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipHostFree.argtypes = [ctypes.c_void_p]
    hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
    h, w = 300, 1024
    n = h * w
    l, r = synth.rectified_pair(41, h, w, 64)
    src = synth.texture(42, h, w)
    Hinv = oracle.invert3x3(synth.homography(5, w, h))
    want = oracle.disparity_ext(l, r, 9, 64, 0, sliding=True)
    want_top = oracle.disparity_ext(l[:h // 2], r[:h // 2], 9, 64, 0, sliding=True)
    wantr = oracle.rectify(src, np.zeros_like(src), Hinv)
    blocks = []

    def host_block(nbytes):
        p = ctypes.c_void_p()
        assert hip.hipHostMalloc(ctypes.byref(p), nbytes, 0) == 0
        blocks.append(p.value)
        return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(p.value))

    def check_calls(pl, pr, po):
        for on in (True, False):
            ctx.set_host_pipeline(on, 0, 0)
            try:
                pl[...] = l
                pr[...] = r
                po[...] = 9
                assert (ctx.disparity(pl, pr, win=9, max_disp=64, out=po) == want).all()
                top = po[:h // 2]
                top[...] = 9
                assert (ctx.disparity(pl[:h // 2], pr[:h // 2], win=9, max_disp=64, out=top) == want_top).all()
                pl[...] = src
                top[...] = 3
                ctx.rectify(pl, top, Hinv, api.RECT_DST_IS_ZERO)       # the kernel stores into the caller's rows
                assert (top == wantr[:h // 2]).all()
            finally:
                ctx.set_host_pipeline(True, 0, 0)

    try:
        a = host_block(3 * n)
        pl, pr, po = (a[k * n:(k + 1) * n].reshape(h, w) for k in range(3))
        assert ctx.host_buffer_kind(a) == 2 and ctx.host_buffer_kind(po) == 2 and ctx.host_buffer_kind(po[7:9]) == 2
        assert ctx.host_buffer_kind(np.zeros(n, np.uint8)) == 0                      # pageable
        # one byte past the block's end: the last byte belongs to no (or another) allocation -> staged
        past = np.ctypeslib.as_array((ctypes.c_uint8 * (n + 1)).from_address(blocks[0] + 2 * n))
        b = host_block(n)                                                            # whatever lies behind block a now
        assert ctx.host_buffer_kind(past) == 0
        check_calls(pl, pr, po)
        # registered memory (hipHostRegister): its device-visible address is what the kernels get
        page = 4096
        raw = np.zeros(4 * n + 2 * page, np.uint8)
        off = (-raw.ctypes.data) % page
        buf = raw[off:off + 4 * n]
        half = (h // 2) * w
        spans = [(0, n), (n, n), (2 * n, half), (2 * n + half, n - half)]            # the output rows: TWO registrations
        regs = []
        try:
            for lo, cnt in spans:
                if hip.hipHostRegister(buf[lo:].ctypes.data, cnt, 3) == 0:            # portable | mapped
                    regs.append(buf[lo:].ctypes.data)
            if len(regs) == len(spans):
                rl, rr, ro = (buf[k * n:(k + 1) * n].reshape(h, w) for k in range(3))
                assert ctx.host_buffer_kind(rl) >= 1 and ctx.host_buffer_kind(ro[:h // 2]) >= 1
                assert ctx.host_buffer_kind(ro) == 0          # spans two registrations: staged
                check_calls(rl, rr, ro)
            else:
                print("hipHostRegister unavailable on this box: registered-memory part not run")
        finally:
            ctx.synchronize()
            for p_ in regs:
                hip.hipHostUnregister(p_)
    finally:
        ctx.synchronize()
        for p_ in blocks:
            hip.hipHostFree(p_)


def test_resident_rectified_images_feed_the_scan(ctx, oracle):
    """stx_rectify_u8_keep x2 + stx_disparity_resident_u8: main.cpp:362-380 with the two rectified
    images never uploaded again; both pipeline settings; stale or mismatched slots are refused."""
    from stereo_amd import StereoHipError
    h, w = 200, 520
    a, b = synth.texture(21, h, w), synth.texture(22, h, w)
    i0 = oracle.invert3x3(synth.homography(1, w, h, deg_y=1.0))
    i1 = oracle.invert3x3(synth.homography(2, w, h, deg_y=-1.5))
    r0 = oracle.rectify(a, np.zeros((h, w), np.uint8), i0)
    r1 = oracle.rectify(b, np.zeros((h, w), np.uint8), i1)
    for on in (True, False):
        ctx.set_host_pipeline(on, 0, 0)
        try:
            g0, g1 = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
            ctx.rectify_keep(0, a, g0, i0)
            ctx.rectify_keep(1, b, g1, i1)
            assert (g0 == r0).all() and (g1 == r1).all()
            # other calls in between leave the resident images alone
            ctx.disparity(a, b, win=3, max_disp=16)
            ctx.rectify(a, np.zeros((h, w), np.uint8), i1)
            for win, D, cm in [(1, 0, 0), (9, 64, 0), (13, 40, 1)]:
                got = ctx.disparity_resident(0, 1, (h, w), win=win, max_disp=D, cost_mode=cm)
                assert (got == oracle.disparity_ext(r0, r1, win, D, cm, sliding=D > 0)).all()
            assert (ctx.disparity_resident(1, 0, (h, w)) == oracle.disparity_ref(r1, r0)).all()
            with pytest.raises(StereoHipError):
                ctx.disparity_resident(0, 1, (h, w - 1))       # not the resident size
            small = np.zeros((10, 12), np.uint8)
            ctx.rectify_keep(1, a, small, i0)                   # slot 1 now holds a 12 x 10 image
            with pytest.raises(StereoHipError):
                ctx.disparity_resident(0, 1, (h, w))
        finally:
            ctx.set_host_pipeline(True, 0, 0)


def test_two_contexts_on_two_threads(oracle):
    """One context per thread (the ABI's threading rule): two host threads, each with its own
    context, staging threads and streams, run different host-pointer calls at the same time."""
    import threading
    import stereo_amd
    l, r = synth.rectified_pair(77, 300, 1200, 64)
    src = synth.texture(78, 300, 1200)
    Hinv = oracle.invert3x3(synth.homography(4, 1200, 300))
    want_d = oracle.disparity_ext(l, r, 9, 64, 0, sliding=True)
    want_r = oracle.rectify(src, np.zeros_like(src), Hinv)
    errors = []

    def work(kind):
        try:
            c = stereo_amd.Context(0)
            try:
                for _ in range(12):
                    if kind == 0:
                        assert (c.disparity(l, r, win=9, max_disp=64) == want_d).all()
                    else:
                        d = np.zeros_like(src)
                        c.rectify(src, d, Hinv, api.RECT_DST_IS_ZERO)
                        assert (d == want_r).all()
            finally:
                c.close()
        except Exception as e:  # noqa
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(k % 2,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_entry_points_leave_the_callers_device_alone(ctx):
    """Every entry point restores the thread's current device (one GPU here: it must stay 0 and
    torch must keep working between calls)."""
    import torch
    l, r = synth.rectified_pair(1, 20, 100, 8)
    before = torch.cuda.current_device()
    ctx.disparity(l, r, win=3, max_disp=8)
    t = torch.ones(4, device="cuda") * 2
    assert torch.cuda.current_device() == before and float(t.sum().item()) == 8.0


def test_library_before_torch():
    """Round 1's load-order trap: libstereo_hip.so loaded and used BEFORE torch initialises its
    HIP runtime.  A fresh process each way; both must see the GPU from both sides, with one
    libamdhip64 mapped."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for order in ("lib-first", "torch-first"):
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "load_order_probe.py"), order],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "libstereo_hip ok" in p.stdout and "torch ok" in p.stdout, p.stdout + p.stderr
        final = p.stdout.split("--- final mappings")[1]
        assert len([l for l in final.splitlines() if "amdhip64" in l]) == 1, p.stdout


# ------------------------------------------------------- several GPUs, one process ----

def stereo_amd_lib():
    import stereo_amd
    return stereo_amd.load_library()


def test_multi_device_handle_matches_single_calls(ctx, oracle, monkeypatch):
    """stx_multi_*: pairs dealt to the devices, one pair cut into row bands with halo, queries
    sharded -- each bit-equal to the single-context call (n = 1 here on a one-GPU box; n = 2 and
    n = all where the box has them)."""
    from stereo_amd import MultiContext
    pairs = [synth.rectified_pair(50 + k, 90, 410, 40) for k in range(5)]
    a = np.stack([p[0] for p in pairs])
    b = np.stack([p[1] for p in pairs])
    big_l, big_r = synth.rectified_pair(60, 301, 900, 300)
    d1, s1, d2, s2 = synth.descriptor_sets(12, 2500, 3000)
    want_batch = np.stack([ctx.disparity(p[0], p[1], win=9, max_disp=40) for p in pairs])
    assert (want_batch[2] == oracle.disparity_ext(pairs[2][0], pairs[2][1], 9, 40, 0)).all()
    want_ref = ctx.disparity(big_l, big_r)
    want_w8 = ctx.disparity(big_l, big_r, win=11, max_disp=64)
    want_w16 = ctx.disparity(big_l, big_r, win=9, max_disp=300, out_dtype=np.uint16)
    want_ssd = ctx.disparity(big_l, big_r, win=5, max_disp=32, cost_mode=api.COST_SSD)
    want_m = ctx.match(d1, s1, d2, s2, 1000.0)
    src = synth.texture(61, 301, 900)
    Hinv = oracle.invert3x3(synth.homography(7, 900, 301))
    seed_dst = np.random.RandomState(4).randint(0, 255, (310, 880)).astype(np.uint8)
    want_r = oracle.rectify(src, seed_dst.copy(), Hinv)
    want_rz = oracle.rectify(src, np.zeros((310, 880), np.uint8), Hinv)
    have = int(stereo_amd_lib().stx_device_count())
    # n = 2 and n = 4 run on a one-GPU box too: STX_MULTI_ALLOW_REPEAT=1 (test-only) lets the handle
    # name device 0 more than once, so the threaded shares / halo uploads / staging pools execute
    monkeypatch.setenv("STX_MULTI_ALLOW_REPEAT", "1")
    cases = [[0], [0, 0], [0, 0, 0, 0], [k % have for k in range(2)], [k % have for k in range(min(have, 8))]]
    for ids in [c for i, c in enumerate(cases) if c not in cases[:i]]:
        n = len(ids)
        m = MultiContext(ids)
        try:
            assert m.size == n
            assert (m.disparity_batch(a, b, win=9, max_disp=40) == want_batch).all()
            assert (m.disparity_batch(a[:1], b[:1], win=9, max_disp=40) == want_batch[:1]).all()
            assert (m.disparity_rowbands(big_l, big_r) == want_ref).all()
            assert (m.disparity_rowbands(big_l, big_r, win=11, max_disp=64) == want_w8).all()
            assert (m.disparity_rowbands(big_l, big_r, win=9, max_disp=300, out_dtype=np.uint16) == want_w16).all()
            assert (m.disparity_rowbands(big_l, big_r, win=5, max_disp=32, cost_mode=api.COST_SSD) == want_ssd).all()
            assert (m.disparity_rowbands(big_l[:3], big_r[:3], win=9, max_disp=16)
                    == ctx.disparity(big_l[:3], big_r[:3], win=9, max_disp=16)).all()
            _match_equal(m.match(d1, s1, d2, s2, 1000.0), want_m)
            _match_equal(m.match(d1[:1], s1[:1], d2, s2, 1000.0), [x[:1] for x in want_m])
            assert (m.rectify(src, seed_dst.copy(), Hinv) == want_r).all()          # misses keep the destination
            padded = np.full((310, 900), 5, np.uint8)
            m.rectify(src, padded[:, :880], Hinv, api.RECT_DST_IS_ZERO)
            assert (padded[:, :880] == want_rz).all() and (padded[:, 880:] == 5).all()
        finally:
            m.close()


def test_multi_device_argument_errors(monkeypatch):
    from stereo_amd import MultiContext, StereoHipError
    monkeypatch.delenv("STX_MULTI_ALLOW_REPEAT", raising=False)
    with pytest.raises(StereoHipError):
        MultiContext([0, 0])          # a device twice
    with pytest.raises(StereoHipError):
        MultiContext([99])            # no such device
    m = MultiContext([0])
    try:
        a = np.ones((2, 4, 4), np.uint8)
        with pytest.raises(StereoHipError):
            m.disparity_batch(a, a, win=4, max_disp=4)
    finally:
        m.close()


def test_real_backend_takes_the_wire_format():
    """nccl (= RCCL) with world size 1 on this box: the byte view gather_rows posts goes through a
    grouped send / receive and views back unchanged (tools/nccl_wire_probe.py; the probe also
    prints whether the backend refuses the int16 tensor round 2 would have posted)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(29700 + os.getpid() % 200))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_wire_probe.py")],
                       capture_output=True, text=True, timeout=170, env=env)
    assert p.returncode == 0 and "WIRE OK" in p.stdout, p.stdout + p.stderr
