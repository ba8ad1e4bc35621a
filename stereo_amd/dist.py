"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Nothing on this path is a reduction.  Batches shard by whole image pairs,
a single large pair shards by epipolar row bands (the disparity search is
row-local; a win x win window needs win//2 halo rows, which each rank reads
from its own copy of the input, so there is no GPU<->GPU halo exchange), and
matching shards list1's queries with list2 replicated.  The only collective is
the optional gather of results to rank 0.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) share of n units for `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def row_band(full_h, rank, world, halo):
    """Rows this rank outputs, and the rows it must hold to do so.

    Returns (row0, rows, band_row0, band_rows): output rows [row0, row0+rows),
    input rows [band_row0, band_row0+band_rows) = the output rows plus `halo`
    rows on each side, clipped to the image."""
    row0, row1 = shard_range(full_h, rank, world)
    b0 = max(0, row0 - halo)
    b1 = min(full_h, row1 + halo)
    if row1 == row0:
        b0, b1 = row0, row0
    return row0, row1 - row0, b0, b1 - b0


def _wire(t):
    """The tensor as rows of bytes: what actually travels.

    torch's NCCL (= RCCL) type map has no 16-bit integer (NCCLUtils.hpp getNcclDataType: uint8,
    int8, int32, int64, half, float, double, bfloat16 ...), so a u16 disparity map (config 3)
    cannot be posted as it is.  Nothing on this path is a reduction -- the element type means
    nothing to the transport -- so every block goes out as uint8 and is viewed back on the root."""
    t = t.contiguous()
    rows = int(t.shape[0])
    return t.view(torch.uint8).reshape(rows, -1) if t.numel() else t.new_empty((rows, 0), dtype=torch.uint8)


def gather_rows(local, counts, dst=0, group=None):
    """Gather per-rank row blocks (dim 0, sizes `counts`) to rank `dst`.

    Point-to-point sends into the root (grouped isend/irecv): on an 8-GPU
    xGMI node that is 7 concurrent links into the root.  Only uint8 tensors are
    handed to the backend (see _wire).  Returns the concatenated tensor on
    `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    if len(counts) != world or int(counts[rank]) != int(local.shape[0]):
        raise ValueError("gather_rows: counts[%d] = %s does not describe this rank's %d rows"
                         % (rank, counts[rank] if len(counts) == world else "?", int(local.shape[0])))
    tail = tuple(local.shape[1:])
    row_bytes = local.element_size()
    for d in tail:
        row_bytes *= int(d)
    if rank == dst:
        # one buffer for the whole result: every receive lands in its final place (no torch.cat copy)
        total = sum(int(c) for c in counts)
        out = torch.empty((total,) + tail, dtype=local.dtype, device=local.device)
        wire = out.view(torch.uint8).reshape(total, row_bytes) if out.numel() else None
        ops = []
        at = 0
        for r in range(world):
            c = int(counts[r])
            if r == dst:
                out[at:at + c] = local
            elif c > 0 and row_bytes > 0:
                ops.append(dist.P2POp(dist.irecv, wire[at:at + c], r, group))
            at += c
        # one group: on RCCL the receives are posted together and run on all links at once
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return out
    if local.shape[0] > 0 and row_bytes > 0:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, _wire(local), dst, group)]):
            q.wait()
    return None


def disparity_rowbands(compute_band, img0, img1, win, group=None, gather=True):
    """Row-band sharded disparity of ONE pair held (replicated) by every rank.

    compute_band(img0_band, img1_band, band_row0, full_h, row0, rows) -> [rows, W]
    tensor for output rows [row0, row0+rows).  Returns the full map on rank 0
    (None elsewhere) when gather=True, else this rank's band."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    full_h = int(img0.shape[0])
    row0, rows, b0, bh = row_band(full_h, rank, world, win // 2)
    out = compute_band(img0[b0:b0 + bh], img1[b0:b0 + bh], b0, full_h, row0, rows)
    if not gather or world == 1:
        return out
    counts = [row_band(full_h, r, world, win // 2)[1] for r in range(world)]
    return gather_rows(out, counts, 0, group)


def disparity_pairs(compute_pairs, n_pairs, group=None):
    """Pair-sharded batch: rank r computes pairs shard_range(n_pairs, r, world).
    compute_pairs(lo, hi) -> [hi-lo, H, W] tensor.  No collective: each rank
    keeps (or copies out) its own results."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_pairs, rank, world)
    return lo, hi, compute_pairs(lo, hi)


def rectify_rowbands(compute_band, dst_h, group=None, gather=True):
    """Row-band sharded warp of ONE image whose source every rank holds.

    compute_band(row0, rows) -> [rows, W] tensor with destination rows [row0, row0+rows).
    The warp is output-pixel-local, so there is no halo and no exchange; with gather=True the
    bands are collected on rank 0 (None elsewhere), else each rank keeps its band (the case
    where the band feeds this rank's disparity band directly)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    row0, row1 = shard_range(dst_h, rank, world)
    out = compute_band(row0, row1 - row0)
    if not gather or world == 1:
        return out
    counts = [shard_range(dst_h, r, world)[1] - shard_range(dst_h, r, world)[0] for r in range(world)]
    return gather_rows(out, counts, 0, group)


def match_sharded(compute_queries, n1, group=None, gather=True):
    """Query-sharded matching: rank r matches list1 rows
    shard_range(n1, r, world) against the whole (replicated) list2.

    compute_queries(lo, hi) -> tuple of per-query tensors of length hi-lo (closest, second,
    min_dist, second_dist, status).  Queries are independent, so the per-rank results are simply
    the rows [lo, hi) of the single-GPU result; gather=True concatenates them on rank 0 in list1
    order (the order the reference emits its pairs in, Features.cpp:1028), None elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n1, rank, world)
    parts = compute_queries(lo, hi)
    if not gather or world == 1:
        return parts
    counts = [shard_range(n1, r, world)[1] - shard_range(n1, r, world)[0] for r in range(world)]
    res = [gather_rows(p, counts, 0, group) for p in parts]
    return tuple(res) if rank == 0 else None


def ransac_sharded(compute_hypotheses, n_hyp, group=None, gather=True):
    """Hypothesis-sharded RANSAC scoring: rank r scores hypotheses
    shard_range(n_hyp, r, world) over the whole (replicated) pair list -- hypotheses are
    independent, so there is no exchange.

    compute_hypotheses(lo, hi) -> (inliers, err_sum) tensors of length hi-lo.  gather=True
    concatenates them on rank 0 in iteration order (the order the selection rule of
    Stereography.cpp:239-246 walks them in), None elsewhere."""
    return match_sharded(compute_hypotheses, n_hyp, group, gather)
