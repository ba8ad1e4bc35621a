"""world_size-2 gloo tests of the multi-GPU sharding logic (CPU, oracle compute)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_and_row_band():
    from stereo_amd.dist import shard_range, row_band
    for n in (0, 1, 7, 8, 1988):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    row0, rows, b0, bh = row_band(100, 1, 4, 4)
    assert (row0, rows, b0, bh) == (25, 25, 21, 33)
    assert row_band(100, 0, 4, 4) == (0, 25, 0, 29)
    assert row_band(100, 3, 4, 4) == (75, 25, 71, 29)


def _worker(rank, world, port, win, D, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as o
        from stereo_amd import synth
        from stereo_amd.dist import disparity_rowbands, disparity_pairs
        left, right = synth.rectified_pair(11, 37, 53, D)
        full = o.disparity_ext(left, right, win, D, 0)

        def band(i0, i1, band_row0, full_h, row0, rows):
            # the band carries its halo, so clipping the window to the band
            # equals clipping it to the full image
            res = o.disparity_ext(i0.numpy(), i1.numpy(), win, D, 0)
            return torch.from_numpy(res[row0 - band_row0: row0 - band_row0 + rows].copy())

        got = disparity_rowbands(band, torch.from_numpy(left), torch.from_numpy(right), win)
        ok = True
        if rank == 0:
            ok = bool((got.numpy() == full).all())
        else:
            ok = got is None

        lo, hi, mine = disparity_pairs(lambda a, b: torch.arange(a, b), 5)
        counts = torch.tensor([hi - lo])
        dist.all_reduce(counts)
        ok = ok and int(counts) == 5 and mine.tolist() == list(range(lo, hi))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("win,D", [(1, 16), (5, 12)])
def test_rowband_sharding_world2(win, D):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + win
    procs = [ctx.Process(target=_worker, args=(r, 2, port, win, D, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker_rm(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as o
        from stereo_amd import synth
        from stereo_amd.dist import rectify_rowbands, match_sharded
        h, w = 41, 60
        src = synth.texture(5, h, w)
        Hinv = o.invert3x3(synth.homography(5, w, h))
        full = o.rectify(src, np.zeros((h, w), np.uint8), Hinv)
        # the warp is output-pixel-local: a band is a slice of the full result
        got = rectify_rowbands(lambda row0, rows: torch.from_numpy(full[row0:row0 + rows].copy()), h)
        ok = bool((got.numpy() == full).all()) if rank == 0 else got is None

        n1, n2 = 37, 50
        d1, s1, d2, s2 = synth.descriptor_sets(9, n1, n2)
        want = o.match(d1, s1, d2, s2, 1000.0)

        def queries(lo, hi):
            return tuple(torch.from_numpy(a) for a in o.match(d1[lo:hi], s1[lo:hi], d2, s2, 1000.0))

        res = match_sharded(queries, n1)
        if rank == 0:
            for g, wnt in zip(res, want):
                ok = ok and g.numpy().tobytes() == wnt.tobytes()
        else:
            ok = ok and res is None

        # hypothesis-sharded RANSAC scoring (N2): golden records, 8 hypotheses over 2 ranks
        from stereo_amd.dist import ransac_sharded
        g = np.load(os.path.join(ROOT, "tests", "golden", "n2_ransac_8x96.npz"))

        def hyps(lo, hi):
            inl, tot, _, _ = o.ransac_score(g["hyp"][lo:hi], g["K1"], g["K2"], g["p1"], g["p2"],
                                            g["exclude"][lo:hi], 70.0)
            return torch.from_numpy(inl), torch.from_numpy(tot)

        res = ransac_sharded(hyps, 8)
        if rank == 0:
            ok = ok and res[0].numpy().tolist() == g["inliers"].tolist()
            ok = ok and res[1].numpy().tobytes() == g["err_sum"].tobytes()
        else:
            ok = ok and res is None
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_rectify_and_match_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker_rm, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


# ------------------------------------------------ what reaches the backend's type map ----

# torch/include/torch/csrc/distributed/c10d/NCCLUtils.hpp, getNcclDataType: the element types
# torch's NCCL (= RCCL) backend accepts.  There is no 16-bit integer among them; gloo takes
# anything, so only a check on the tensors handed to P2POp sees a u16 map on its way to RCCL.
NCCL_DTYPES = {torch.uint8, torch.int8, torch.int32, torch.int64, torch.float16, torch.float32,
               torch.float64, torch.bfloat16, torch.bool}


def _worker_wire(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stereo_amd import dist as sdist
        seen = []
        real = dist.P2POp

        def checked(op, tensor, peer, group=None, *a, **k):
            seen.append(tensor.dtype)
            assert tensor.dtype in NCCL_DTYPES, "a %s tensor would reach RCCL" % tensor.dtype
            assert tensor.is_contiguous()
            return real(op, tensor, peer, group, *a, **k)

        dist.P2POp = checked
        ok = True
        # config 3's u16 map cut into row bands (3 columns wide, so a row is an odd number of u16s)
        full = (torch.arange(7 * 3, dtype=torch.int32).reshape(7, 3) * 2500 % 65536).to(torch.uint16)
        for dt in (torch.uint16, torch.int16, torch.uint8, torch.int32, torch.float32):
            f = full.view(torch.int16).to(dt) if dt not in (torch.uint16,) else full
            got = sdist.disparity_rowbands(
                lambda i0, i1, b0, fh, row0, rows: f[row0:row0 + rows].clone(), f, f, 3)
            if rank == 0:
                ok = ok and got.dtype == dt and got.shape == f.shape and got.view(torch.uint8).tolist() == f.view(torch.uint8).tolist()
            else:
                ok = ok and got is None
        # the five per-query arrays of matching and the two of RANSAC scoring, ragged shares (n = 5)
        arrs = (torch.arange(5, dtype=torch.int32), -torch.arange(5, dtype=torch.int32),
                torch.arange(5, dtype=torch.float32) / 3, torch.arange(5, dtype=torch.float32) / 7,
                torch.arange(5, dtype=torch.uint8))
        res = sdist.match_sharded(lambda lo, hi: tuple(a[lo:hi].clone() for a in arrs), 5)
        if rank == 0:
            ok = ok and all(g.dtype == a.dtype and g.tolist() == a.tolist() for g, a in zip(res, arrs))
        res = sdist.ransac_sharded(lambda lo, hi: (arrs[0][lo:hi].clone(), arrs[2][lo:hi].clone()), 5)
        if rank == 0:
            ok = ok and res[0].tolist() == arrs[0].tolist() and res[1].tolist() == arrs[2].tolist()
        # an empty share (1 row over 2 ranks) and a multi-dimensional block (pairs of config 4)
        one = torch.full((1, 4), 9, dtype=torch.uint16)
        got = sdist.rectify_rowbands(lambda row0, rows: one[row0:row0 + rows].clone(), 1)
        if rank == 0:
            ok = ok and got.tolist() == one.tolist()
        maps = torch.arange(3 * 2 * 5, dtype=torch.uint8).reshape(3, 2, 5)
        lo, hi = sdist.shard_range(3, rank, world)
        got = sdist.gather_rows(maps[lo:hi].clone(), [2, 1], 0)
        if rank == 0:
            ok = ok and got.tolist() == maps.tolist()
        # a count that does not describe the block is refused before anything is posted
        try:
            sdist.gather_rows(maps[lo:hi].clone(), [1, 2], 0)
            ok = False
        except ValueError:
            pass
        ok = ok and len(seen) > 0 and set(seen) == {torch.uint8}
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_only_bytes_reach_the_backend_world2():
    """the u16 map of config 3 went to dist.P2POp as torch.int16, which RCCL's type
    map rejects.  Every tensor handed to the backend must have a dtype of that map."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30300 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker_wire, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
