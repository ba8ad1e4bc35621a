"""Host side of FindFundamentalMatrixWithRANSAC (stereo/Stereography.cpp:184-281) around the
batched scoring kernel (Context.ransac_score / stx_ransac_score).

The reference draws eight pairs, solves for F and decomposes E once per iteration, then scores
all remaining pairs -- 200 times in sequence.  Neither the draws nor the solves depend on the
scores, so here all hypotheses are made first (host, a few hundred 8x9 / 3x3 SVDs), scored in ONE
GPU call, and the reference's selection rule (:239-246, :275-280) runs over the results.

Two things differ from the reference by necessity and are stated where they occur:
  * the seed is a parameter (the reference seeds rand() from an uninitialised matrix, :193);
    the draw sequence itself is the MSVC CRT's rand() that the reference's build uses;
  * the SVDs are LAPACK's (numpy), not Eigen's BDCSVD: F is unaffected (the sign of the null
    vector cancels in F /= F(2,2), :158), but the signs of the singular vectors that
    DecomposeEssentialMatrix turns into R and t are a convention of the SVD routine.  A caller
    that has Eigen (the C++ adapter) passes Eigen's matrices instead; the kernel only sees records.
"""
import numpy as np

from .api import (FUNDAMENTAL_RANSAC_ITERATIONS, FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD, HYP_FLOATS,
                  MIN_NUM_INLIERS, default_context)

f32 = np.float32


def msvc_rand(seed):
    """The MSVC CRT's rand() after srand(seed): a 32-bit LCG, bits 16..30 of the state."""
    state = int(seed) & 0xFFFFFFFF
    while True:
        state = (state * 214013 + 2531011) & 0xFFFFFFFF
        yield (state >> 16) & 0x7FFF


def draw_eight(n, rng):
    """:205-211 -- eight positions of the pair list, each drawn from the pairs still left."""
    left = list(range(n))
    chosen = []
    while len(chosen) < 8:
        chosen.append(left.pop(next(rng) % len(left)))
    return chosen


def normalisation_transforms(p1, p2):
    """GetNormalisationTransformAndNormalisePoints, :28-75 (its point list is a by-value copy, so
    only the two transforms come back)."""
    out = []
    for p in (p1, p2):
        n = f32(len(p))
        c = np.zeros(2, f32)
        for q in p:
            c = c + q
        c = c / n
        avg = f32(0)
        for q in p:
            d = q - c
            avg = avg + np.sqrt(f32(d[0] * d[0] + d[1] * d[1]))
        avg = avg / n
        s = f32(np.sqrt(2.0) / np.float64(avg))
        out.append(np.array([[s, 0, f32(-1) * c[0] * s], [0, s, f32(-1) * c[1] * s], [0, 0, 1]], f32))
    return out


def find_fundamental_matrix(p1, p2):
    """FindFundamentalMatrix, :77-161, on the given pairs (float32 [m,2] each).  (ok, F)."""
    p1 = np.asarray(p1, f32).reshape(-1, 2)
    p2 = np.asarray(p2, f32).reshape(-1, 2)
    if len(p1) < 8:
        return False, None
    t1, t2 = normalisation_transforms(p1, p2)
    ones = np.ones((len(p1), 1), f32)
    a = (t1 @ np.hstack([p1, ones]).T).T.astype(f32)   # first = normalise1 * (x, y, 1)
    b = (t2 @ np.hstack([p2, ones]).T).T.astype(f32)
    y = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1],
                  a[:, 0], a[:, 1], np.ones(len(a), f32)], axis=1).astype(f32)
    try:
        _, _, vt = np.linalg.svd(y, full_matrices=True)
    except np.linalg.LinAlgError:
        return False, None
    f = vt[8].astype(f32)
    f = f / f32(np.sqrt(f32(np.dot(f, f))))
    fm = (t2.T @ f.reshape(3, 3) @ t1).astype(f32)
    # :158 divides by F(2,2) whatever it is: a hypothesis with F(2,2) = 0 comes out as inf / NaN,
    # as in the reference, and is dropped by its callers' isfinite checks
    with np.errstate(divide="ignore", invalid="ignore"):
        return True, (fm / fm[2, 2]).astype(f32)


_W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], f32)


def decompose_essential_matrix(e):
    """DecomposeEssentialMatrix, :301-353.  (ok, E scaled by its first singular value, R1, R2, t)."""
    e = np.asarray(e, f32).reshape(3, 3)
    try:
        d = np.linalg.svd(e, compute_uv=False)
        e = (e / f32(d[0])).astype(f32)   # :317-318, in place in the reference
        u, _, vt = np.linalg.svd(e)
    except np.linalg.LinAlgError:
        return False, e, None, None, None
    u = u.astype(f32)
    v = vt.T.astype(f32)
    if f32(np.linalg.det(u @ v.T)) == f32(-1):   # :330, an exact comparison
        v = -v
    r1 = (u @ _W @ v.T).astype(f32)
    r2 = (u @ _W.T @ v.T).astype(f32)
    return True, e, r1, r2, u[:, 2].copy()


def hypothesis_record(fundamental, k1, k2):
    """One STX_HYP_FLOATS record for a fundamental matrix (:220-225 and the decomposition inside
    Triangulate, :398), or None where the outer decomposition fails."""
    k1 = np.asarray(k1, f32).reshape(3, 3)
    k2 = np.asarray(k2, f32).reshape(3, 3)
    e = (k2.T @ np.asarray(fundamental, f32).reshape(3, 3) @ k1).astype(f32)   # :220
    ok, e, _, r, t = decompose_essential_matrix(e)   # :225: R is the function's second rotation
    if not ok or not np.all(np.isfinite(e)):
        return None
    rec = np.zeros(HYP_FLOATS, f32)
    rec[0:9] = e.reshape(9)
    rec[9:18] = r.reshape(9)
    rec[18:21] = t
    ok2, _, _, r_in, t_in = decompose_essential_matrix(e.copy())   # :398, on ReprojectionError's copy
    if ok2:
        rec[21:30] = r_in.reshape(9)
        rec[30:33] = t_in
        rec[33] = 1
    return rec


def select_hypothesis(inliers, err_sum):
    """The selection rule of :239-246 and :275-280 over scored hypotheses, in iteration order.
    Returns the winning position or -1.  minError is an int in the reference (:191): the average
    error is truncated when it is stored (:244)."""
    limit = FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD * MIN_NUM_INLIERS
    min_error = limit
    best = -1
    for h, (cnt, tot) in enumerate(zip(inliers, err_sum)):
        cnt = int(cnt)
        if cnt < 0:
            continue   # hypothesis that FindFundamentalMatrix / the decomposition did not produce
        avg = f32(tot)
        if cnt > 0:
            avg = f32(avg / f32(cnt))
        if cnt > MIN_NUM_INLIERS and avg < f32(min_error):
            min_error = int(avg)
            best = h
    return best if min_error < limit else -1


def FindFundamentalMatrixWithRANSAC(p1, p2, K1, K2, seed=0, iterations=FUNDAMENTAL_RANSAC_ITERATIONS, ctx=None):
    """(ok, F) for matched positions p1[i] <-> p2[i] (float32 [n,2]) and camera matrices K1, K2.
    Stereography.h:29 with the match list split into its two position arrays and
    stereo.img1.K / stereo.img2.K passed directly."""
    p1 = np.ascontiguousarray(p1, f32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, f32).reshape(-1, 2)
    n = len(p1)
    if len(p2) != n:
        raise ValueError("p1 and p2 must have the same length")
    if n < 8:
        raise ValueError("fewer than eight matches (the reference's draw loop does not terminate)")
    ctx = ctx or default_context()
    rng = msvc_rand(seed)
    fs, recs, excl, slot = [], [], [], []
    for it in range(iterations):
        chosen = draw_eight(n, rng)
        ok, fm = find_fundamental_matrix(p1[chosen], p2[chosen])
        rec = hypothesis_record(fm, K1, K2) if ok and np.all(np.isfinite(fm)) else None
        fs.append(fm)
        if rec is None:
            slot.append(-1)
            continue
        slot.append(len(recs))
        recs.append(rec)
        excl.append(chosen)
    inl = np.full(iterations, -1, np.int64)
    tot = np.zeros(iterations, f32)
    if recs:
        gi, gs = ctx.ransac_score(np.stack(recs), K1, K2, p1, p2, exclude=np.array(excl, np.int32))
        for it, s in enumerate(slot):
            if s >= 0:
                inl[it], tot[it] = gi[s], gs[s]
    best = select_hypothesis(inl, tot)
    if best < 0:
        return False, None
    return True, fs[best]
