"""stx_disparity_dev_noalloc: the entry point that allocates nothing (static device scratch for
the keys a scan of more than 256 disparities carries between passes) against stx_disparity_dev,
bit for bit, on two calls in a row -- the second with a smaller input, so stale keys of the first
call lie in the scratch array beyond and beneath the second call's."""
import numpy as np
import pytest

from stereo_amd import api, synth

pytestmark = pytest.mark.gpu


def test_no_alloc_disparity_matches_the_allocating_entry_point(ctx, oracle):
    import torch
    from stereo_amd import StereoHipError
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        # (h, w, D, win, cost, out dtype): two passes SAD u16, then a smaller three-pass SSD,
        # then single-pass and reference-mode calls (no scratch needed at all)
        cases = [(300, 1400, 512, 9, api.COST_SAD, torch.int16),
                 (120, 900, 600, 5, api.COST_SSD, torch.int16),
                 (120, 900, 300, 3, api.COST_SAD, torch.uint8),
                 (64, 700, 100, 9, api.COST_SAD, torch.uint8),
                 (40, 500, 0, 1, api.COST_SAD, torch.uint8)]
        for k, (h, w, D, win, cm, dt) in enumerate(cases):
            l, r = synth.rectified_pair(900 + k, h, w, max(D, 16))
            a, b = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
            want = torch.zeros((h, w), dtype=dt, device=dev)
            got = torch.full((h, w), 77, dtype=dt, device=dev)
            ctx.disparity_dev(a, b, want, win=win, max_disp=D, cost_mode=cm)
            ctx.disparity_dev_noalloc(a, b, got, win=win, max_disp=D, cost_mode=cm)
            torch.cuda.synchronize()
            assert torch.equal(got, want), "case %d" % k
            if k < 2:  # and the allocating path is the oracle's
                npdt = np.uint16 if dt == torch.int16 else np.uint8
                ref = oracle.disparity_ext(l, r, win, D, cm, npdt, sliding=(cm == api.COST_SAD))
                assert (want.cpu().numpy().view(npdt) == ref).all()
        # a batch: 3 pairs, two passes
        pairs = [synth.rectified_pair(950 + k, 100, 640, 400) for k in range(3)]
        a = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
        b = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
        want = torch.zeros(a.shape, dtype=torch.int16, device=dev)
        got = torch.zeros(a.shape, dtype=torch.int16, device=dev)
        ctx.disparity_dev(a, b, want, win=7, max_disp=400)
        ctx.disparity_dev_noalloc(a, b, got, win=7, max_disp=400)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        # beyond the scratch array: a clear error, nothing launched
        big_h, big_w = 2200, 3900   # 8.58 Mpixel > NOALLOC_MAX_PIXELS
        assert big_h * big_w > api.NOALLOC_MAX_PIXELS
        z = torch.ones((big_h, big_w), dtype=torch.uint8, device=dev)
        o = torch.zeros((big_h, big_w), dtype=torch.int16, device=dev)
        with pytest.raises(StereoHipError) as e:
            ctx.disparity_dev_noalloc(z, z, o, win=3, max_disp=512)
        assert e.value.code == -5
        # ... while 256 disparities need no scratch and pass at any size
        o8 = torch.zeros((big_h, big_w), dtype=torch.uint8, device=dev)
        w8 = torch.zeros((big_h, big_w), dtype=torch.uint8, device=dev)
        ctx.disparity_dev_noalloc(z, z, o8, win=3, max_disp=256)
        ctx.disparity_dev(z, z, w8, win=3, max_disp=256)
        torch.cuda.synchronize()
        assert torch.equal(o8, w8)
    finally:
        ctx.set_stream(0)


def test_no_alloc_entry_point_allocates_nothing():
    """The property the entry point exists for: on a FRESH context, calling only
    stx_disparity_dev_noalloc -- two passes, so the keys carried between passes are in use -- leaves
    the context's workspace at zero bytes and the device's free memory where it was."""
    import torch
    import stereo_amd
    dev = torch.device("cuda", 0)
    l, r = synth.rectified_pair(990, 200, 1100, 400)
    a, b = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    out = torch.zeros((200, 1100), dtype=torch.int16, device=dev)
    want = torch.zeros((200, 1100), dtype=torch.int16, device=dev)
    fresh = stereo_amd.Context(0)
    try:
        fresh.set_stream(torch.cuda.current_stream().cuda_stream)
        assert fresh.workspace_bytes() == 0
        # first call: one-time set-up of the kernel's LDS attribute (hipFuncSetAttribute, no memory)
        fresh.disparity_dev_noalloc(a, b, out, win=9, max_disp=400)
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info(dev)
        for _ in range(3):
            fresh.disparity_dev_noalloc(a, b, out, win=9, max_disp=400)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info(dev)
        assert fresh.workspace_bytes() == 0
        assert free1 == free0, (free0, free1)
        # the allocating entry point, for contrast, takes its key buffer from the workspace
        fresh.disparity_dev(a, b, want, win=9, max_disp=400)
        torch.cuda.synchronize()
        assert fresh.workspace_bytes() >= 200 * 1100 * 4
        assert torch.equal(out, want)
    finally:
        fresh.set_stream(0)
        fresh.close()


def test_no_alloc_entry_point_under_hip_graph_capture():
    """What the entry point is for: a caller that records its work into a HIP graph, where hipMalloc
    (and any synchronisation) is illegal.  A two-pass scan is captured on a side stream and replayed
    twice into a cleared output; the allocating entry point gives the expected bytes outside the graph."""
    import torch
    import stereo_amd
    dev = torch.device("cuda", 0)
    l, r = synth.rectified_pair(991, 160, 900, 300)
    a, b = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    out = torch.zeros((160, 900), dtype=torch.int16, device=dev)
    want = torch.zeros((160, 900), dtype=torch.int16, device=dev)
    fresh = stereo_amd.Context(0)
    side = torch.cuda.Stream()
    try:
        fresh.set_stream(torch.cuda.current_stream().cuda_stream)
        fresh.disparity_dev(a, b, want, win=7, max_disp=300)
        torch.cuda.synchronize()
        fresh.set_stream(side.cuda_stream)
        with torch.cuda.stream(side):
            fresh.disparity_dev_noalloc(a, b, out, win=7, max_disp=300)   # one-time kernel attributes, outside the capture
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                fresh.disparity_dev_noalloc(a, b, out, win=7, max_disp=300)
        for _ in range(2):
            out.zero_()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want)
    finally:
        fresh.set_stream(0)
        fresh.close()
