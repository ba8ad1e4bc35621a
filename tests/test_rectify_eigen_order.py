"""How far can the UNPINNED Eigen arithmetic move a rectified pixel?  

RectifyImage (stereo/Stereography.cpp:616-618) leaves three statements to Eigen, whose version the
reference does not pin (only a local include path, stereo.vcxproj:93):

    H.inverse()        3x3 inverse                 oracle: cofactors x (1/det), det = a0 + (a1 + a2)
    H^-1 * pixel       3x3 by 3x1 product          oracle: every row reduced as a0 + (a1 + a2)
    q /= q(2)          perspective division        oracle: a true fp32 division (Eigen >= 3.3)

The oracle (and the GPU kernel, bit-equal to it) takes one reading of each.  This file evaluates
the plausible other readings -- the sums associated as (a0 + a1) + a2, the cofactors DIVIDED by
the determinant, and the multiplication by the reciprocal that Eigen 3.2 did -- in all 8
combinations, on the 741x500 and 1080p warps of the bench, and counts against the oracle's output:

    * pixels that differ at all, and by more than 1 grey level (north_star's bar is +-1 LSB);
    * how many of the latter sit on a DISCONTINUITY of the warp as written: a source coordinate
      within 1e-3 of an integer (floor/ceil pick other taps; an exact integer gives 0,
      Stereography.cpp:584-593) or of a bound of the strict in-image test (:621, :623).

Outside that set the output is a continuous function of the coordinate up to the two (uchar)
truncations, so an ulp in the coordinate can move a pixel by one grey level (by two only where
both row values AND their blend sit on a truncation edge at once -- no such pixel in either
image).  The test asserts exactly that: outside the counted discontinuity set every reading is
within +-1 LSB of the oracle's (= the GPU's) output.  The counts are in DESIGN.md 2 and in
profiles/r03_rectify_eigen_order_exposure.txt (python tests/test_rectify_eigen_order.py).
"""
import itertools

import numpy as np
import pytest

from stereo_amd import synth

f32 = np.float32

VARIANTS = list(itertools.product(("a0+(a1+a2)", "(a0+a1)+a2"), ("x(1/det)", "/det"), ("divide", "x(1/z)")))
BASE = ("a0+(a1+a2)", "x(1/det)", "divide")          # the oracle's
CASES = {"741x500": (500, 741, 5, 0.8, 0.2), "1080p": (1080, 1920, 80, 2.0, 0.5)}


def _sum3(a0, a1, a2, order):
    return a0 + (a1 + a2) if order == "a0+(a1+a2)" else (a0 + a1) + a2


def invert(H, order, inv_mode):
    m = np.asarray(H, f32)
    c = np.zeros((3, 3), f32)
    for i in range(3):
        for j in range(3):
            i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
            c[i, j] = f32(m[i1, j1] * m[i2, j2]) - f32(m[i1, j2] * m[i2, j1])
    det = _sum3(f32(c[0, 0] * m[0, 0]), f32(c[1, 0] * m[1, 0]), f32(c[2, 0] * m[2, 0]), order)
    if inv_mode == "x(1/det)":
        return (c.T * (f32(1) / det)).astype(f32)
    return (c.T / det).astype(f32)


def warp(src, dst, hinv, order, div_mode):
    """RectifyImage + BilinearInterpolatePixel for all pixels at once.  Returns the image, the
    source coordinates and the in-image mask."""
    h, w = dst.shape
    sh, sw = src.shape
    X, Y = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
    hm = np.asarray(hinv, f32)
    one = f32(1)

    def row(k):
        return _sum3(hm[k, 0] * X, hm[k, 1] * Y, hm[k, 2] * one, order)

    with np.errstate(all="ignore"):
        z = row(2)
        if div_mode == "divide":
            qx, qy = row(0) / z, row(1) / z
        else:
            rz = one / z
            qx, qy = row(0) * rz, row(1) * rz
        inb = (0 < qx) & (qx < f32(sw - 1)) & (0 < qy) & (qy < f32(sh - 1))
        cx, cy = np.where(inb, qx, f32(0.5)), np.where(inb, qy, f32(0.5))
        x1, x2, y1, y2 = np.floor(cx), np.ceil(cx), np.floor(cy), np.ceil(cy)
        wx1, wx2 = (x2 - cx) / (x2 - x1), (cx - x1) / (x2 - x1)
        wy1, wy2 = (y2 - cy) / (y2 - y1), (cy - y1) / (y2 - y1)
        ix1, ix2 = x1.astype(np.int64), np.minimum(x2.astype(np.int64), sw - 1)
        iy1, iy2 = y1.astype(np.int64), np.minimum(y2.astype(np.int64), sh - 1)
        s = src.astype(f32)
        r1 = wx1 * s[iy1, ix1] + wx2 * s[iy1, ix2]
        r2 = wx1 * s[iy2, ix1] + wx2 * s[iy2, ix2]
        nan_x = np.isnan(r1) | np.isnan(r2)
        r1 = np.trunc(np.where(nan_x, f32(0), r1))
        r2 = np.trunc(np.where(nan_x, f32(0), r2))
        v = wy1 * r1 + wy2 * r2
        v = np.trunc(np.where(np.isnan(v), f32(0), v))
    out = dst.copy()
    out[inb] = v[inb].astype(np.uint8)
    return out, qx, qy, inb


def discontinuity_set(qx, qy, sw, sh, eps=1e-3):
    """Pixels whose source coordinate (as the oracle computes it) is within eps of an integer on
    either axis or of a bound of the strict in-image test."""
    with np.errstate(invalid="ignore"):
        near_int = (np.abs(qx - np.rint(qx)) < eps) | (np.abs(qy - np.rint(qy)) < eps)
        near_bound = ((np.abs(qx) < eps) | (np.abs(qx - (sw - 1)) < eps) |
                      (np.abs(qy) < eps) | (np.abs(qy - (sh - 1)) < eps))
    return near_int | near_bound | ~np.isfinite(qx) | ~np.isfinite(qy)


def exposure(src, H, base_out=None):
    """Per variant: counts against the oracle's reading (or against base_out when given)."""
    sh, sw = src.shape
    zero = np.zeros_like(src)
    ref, qx, qy, inb = warp(src, zero, invert(H, BASE[0], BASE[1]), BASE[0], BASE[2])
    if base_out is not None:
        ref = base_out
    disc = discontinuity_set(qx, qy, sw, sh)
    rows = {}
    for order, inv_mode, div_mode in VARIANTS:
        out, _, _, _ = warp(src, zero, invert(H, order, inv_mode), order, div_mode)
        d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
        rows[(order, inv_mode, div_mode)] = {
            "differ": int((d > 0).sum()),
            "by_more_than_1": int((d > 1).sum()),
            "of_those_on_discontinuity": int(((d > 1) & disc).sum()),
            "off_discontinuity_by_2": int(((d == 2) & ~disc).sum()),
            "off_discontinuity_by_more_than_2": int(((d > 2) & ~disc).sum()),
            "max_off_discontinuity": int(d[~disc].max()),
        }
    return rows, int(disc.sum()), int(inb.sum()), ref


def _case(name):
    h, w, seed, dy, dx = CASES[name]
    return synth.texture(70 if name == "1080p" else 1001, h, w), synth.homography(seed, w, h, deg_y=dy, deg_x=dx)


def report(name):
    src, H = _case(name)
    rows, ndisc, ninb, _ = exposure(src, H)
    lines = ["%s: %d pixels, %d in the source image, %d on the discontinuity set" % (name, src.size, ninb, ndisc)]
    for k, r in rows.items():
        lines.append("  %-11s %-9s %-7s differ %7d   >1 LSB %5d (%d on discontinuities)   off them: by 2: %d, by >2: %d"
                     % (k + (r["differ"], r["by_more_than_1"], r["of_those_on_discontinuity"],
                             r["off_discontinuity_by_2"], r["off_discontinuity_by_more_than_2"])))
    return "\n".join(lines)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reading_is_the_base_variant(oracle, name):
    """The numpy warp with the oracle's three readings IS the oracle, so the counts below are
    counts against the product's output (GPU == oracle bit for bit, tests/test_gpu_parity.py)."""
    src, H = _case(name)
    hinv = invert(H, BASE[0], BASE[1])
    assert hinv.tobytes() == oracle.invert3x3(H).tobytes()
    got, _, _, _ = warp(src, np.zeros_like(src), hinv, BASE[0], BASE[2])
    assert (got == oracle.rectify(src, np.zeros_like(src), hinv)).all()


@pytest.mark.parametrize("name", list(CASES))
def test_other_eigen_readings_stay_within_one_grey_level_off_the_discontinuities(name):
    src, H = _case(name)
    rows, ndisc, ninb, _ = exposure(src, H)
    assert ninb > 0.8 * src.size
    assert ndisc < 0.01 * src.size                 # the set the statement excludes is small: < 1 % of the image
    r0 = rows[BASE]
    assert r0["differ"] == 0
    moved = 0
    for k, r in rows.items():
        # every other reading changes some pixels (the exposure is real) ...
        if k != BASE:
            moved += r["differ"]
            assert r["differ"] < 0.05 * src.size, (k, r)
        # ... but off the discontinuity set never by more than one grey level: everything that
        # exceeds north_star's +-1 LSB is on a counted discontinuity of the warp as written
        assert r["max_off_discontinuity"] <= 1, (k, r)
        assert r["by_more_than_1"] == r["of_those_on_discontinuity"], (k, r)
    assert moved > 0


@pytest.mark.gpu
def test_gpu_output_against_the_other_eigen_readings(ctx):
    """The same counts against what the GPU actually wrote (741x500 case)."""
    import stereo_amd
    src, H = _case("741x500")
    dst = np.zeros_like(src)
    stereo_amd.RectifyImage(src, dst, H, ctx=ctx)
    rows, _, _, ref = exposure(src, H)
    assert (dst == ref).all()
    rows, _, _, _ = exposure(src, H, base_out=dst)
    for k, r in rows.items():
        assert r["max_off_discontinuity"] <= 1 and r["by_more_than_1"] == r["of_those_on_discontinuity"], (k, r)


if __name__ == "__main__":
    for n in CASES:
        print(report(n))
