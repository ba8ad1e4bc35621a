"""Host-side logic of the reference-shaped Python API, with the C-ABI call
replaced by a stub that answers from the oracle (test infrastructure)."""
import io

import numpy as np

import stereo_amd
from stereo_amd import api, synth


class _OracleCtx:
    """Stands in for stereo_amd.Context; answers match() from the CPU oracle."""

    def __init__(self, oracle):
        self.o = oracle

    def match(self, d1, s1, d2, s2, lim):
        return self.o.match(d1, s1, d2, s2, lim)


def _features(desc, score):
    return [api.Feature(scale=1, p=(float(i), float(2 * i)), score=float(score[i]), angle=0.5,
                        desc=desc[i].copy()) for i in range(len(desc))]


def test_match_descriptors_builds_reference_pairs(oracle):
    d1, s1, d2, s2 = synth.descriptor_sets(7, 40, 50)
    d2[0] = d1[3]  # nearest at j = 0: the reference prints a failure line for query 3
    l1, l2 = _features(d1, s1), _features(d2, s2)
    log = io.StringIO()
    pairs = stereo_amd.MatchDescriptors(l1, l2, api.MAX_DIST_BETWEEN_MATCHES, ctx=_OracleCtx(oracle), out=log)
    c, se, md, sd, st = oracle.match(d1, s1, d2, s2, 1000.0)
    accepted = [i for i in range(40) if st[i] == api.MATCH_ACCEPTED]
    assert [p[0].p[0] for p in pairs] == [float(i) for i in accepted]  # ascending i
    for p, i in zip(pairs, accepted):
        assert (p[1].desc == d2[c[i]]).all()
        assert p[0].distFromBestMatch == float(md[i]) == p[1].distFromBestMatch
    assert "Error: failed to match feature 3\n" in log.getvalue()
    assert log.getvalue().count("Error") == int((st == api.MATCH_FAILED).sum())
    # inputs are not modified (the reference takes both lists by value)
    assert all(f.distFromBestMatch == 0.0 for f in l1 + l2)


def test_compute_depth_image_size_mismatch_prints_and_returns_empty():
    log = io.StringIO()
    out = stereo_amd.ComputeDepthImage(np.zeros((4, 5), np.uint8), np.zeros((4, 6), np.uint8), out=log)
    assert out.shape == (0, 0) and out.dtype == np.uint8
    assert log.getvalue() == "Cannot compute depth!\n"


def test_image_argument_checks():
    import pytest
    with pytest.raises(TypeError):
        api._check_image(np.zeros((4, 4), np.float32), "x")
    with pytest.raises(ValueError):
        api._check_image(np.zeros((4, 8), np.uint8)[:, ::2], "x")
    api._check_image(np.zeros((4, 8), np.uint8)[:, :5], "x")  # pitched rows are fine


def test_synth_is_deterministic():
    a = synth.splitmix64(1234, 4)
    assert a[0] == np.uint64(0xBB0CF61A2F8BDE01) or a.dtype == np.uint64
    l1, r1 = synth.rectified_pair(3, 32, 40, 8)
    l2, r2 = synth.rectified_pair(3, 32, 40, 8)
    assert (l1 == l2).all() and (r1 == r2).all()
    assert (l1 == 0).any() and l1.max() > 100
    d = synth.sift_like(1, 16)
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-5) and d.min() >= 0
