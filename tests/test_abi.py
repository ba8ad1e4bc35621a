"""CPU-only checks of the C-ABI library: it loads, exports every symbol that
include/stereo_hip.h declares, and its host-only helpers agree with the oracle.
No compute entry point is called here (there is no GPU in this container)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "stereo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from stereo_amd import _lib
    lib = _lib.load_library()
    names = _declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "libstereo_hip.so does not export %s" % n
        assert n in _lib.PROTOTYPES, "stereo_amd._lib has no prototype for %s" % n
    assert sorted(_lib.PROTOTYPES) == names


def test_version_and_error_strings():
    from stereo_amd import _lib
    lib = _lib.load_library()
    assert lib.stx_version() == 200
    assert lib.stx_error_string(0) == b"ok"
    assert b"invalid argument" in lib.stx_error_string(-1)
    assert b"unsupported" in lib.stx_error_string(-2)


def test_null_context_is_rejected_without_touching_a_gpu():
    from stereo_amd import _lib
    lib = _lib.load_library()
    assert lib.stx_set_stream(None, None) == -1
    assert lib.stx_synchronize(None) == -1
    assert lib.stx_set_match_algo(None, 0) == -1
    assert lib.stx_disparity_u8(None, None, None, 4, 4, 4, 1, 0, 0, None, 4) == -1
    assert lib.stx_match(None, None, None, 0, None, None, 0, C.c_float(1.0), None, None, None, None, None) == -1
    # the widened rows: every entry point checks its context before any HIP call
    assert lib.stx_fast_detect_u8(None, None, 4, 4, 4, 30, None, 0, None) == -1
    assert lib.stx_score_cluster(None, None, None, 4, 4, 4, None, 0, None, C.c_float(700), C.c_float(2), None, None,
                                 None, None) == -1
    assert lib.stx_cluster_features(None, None, None, 0, 2, None, None) == -1
    assert lib.stx_sift_descriptors(None, None, None, 4, 4, 4, None, 0, None, None) == -1
    assert lib.stx_ransac_score(None, None, 0, None, None, None, None, 0, None, C.c_float(70), None, None, None,
                                None) == -1
    assert lib.stx_ransac_score_dev(None, None, 0, None, None, None, None, 0, None, C.c_float(70), None, None, None,
                                    None) == -1
    assert lib.stx_rectify_disparity_u8(None, None, None, 4, 4, 4, None, None, 1, 0, 0, None, 4, None, None, 4) == -1
    # the multi-device handle and the host pipeline switch
    assert lib.stx_rectify_u8_keep(None, 0, None, 4, 4, 4, None, 4, 4, 4, None, 0) == -1
    assert lib.stx_disparity_resident_u8(None, 0, 1, 4, 4, 1, 0, 0, None, 4) == -1
    assert lib.stx_disparity_dev_noalloc(None, None, None, 1, 4, 4, 4, 16, 1, 0, 0, None, 4, 16, 1) == -1
    assert lib.stx_match_features(None, None, 0, None, 0, 540, 20, 12, C.c_float(1.0), None, None, None, None, None) == -1
    assert lib.stx_set_host_pipeline(None, 1, 0, 0) == -1
    assert lib.stx_multi_size(None) == -1
    assert lib.stx_multi_context(None, 0) is None
    assert lib.stx_multi_create(None, None, 0) == -1
    assert lib.stx_disparity_batch_u8(None, None, None, 1, 4, 4, 4, 16, 1, 0, 0, None, 4, 16) == -1
    assert lib.stx_disparity_rowband_u8(None, None, None, 4, 4, 4, 1, 0, 0, None, 4) == -1
    assert lib.stx_disparity_rowband_u16(None, None, None, 4, 4, 4, 1, 0, 0, None, 8) == -1
    assert lib.stx_match_multi(None, None, None, 0, None, None, 0, C.c_float(1.0), None, None, None, None, None) == -1
    lib.stx_multi_destroy(None)  # a no-op


def test_invert_homography_matches_oracle_bit_for_bit(oracle):
    from stereo_amd import _lib, synth
    lib = _lib.load_library()
    fp = C.POINTER(C.c_float)
    mats = [np.eye(3, dtype=np.float32), np.diag([2, 4, 8]).astype(np.float32)]
    for seed in range(20):
        mats.append(synth.homography(seed, 741, 500, deg_y=1 + seed % 3, deg_x=0.5))
        mats.append((synth.uniform01(seed, 9).reshape(3, 3) * 4 - 2).astype(np.float32))
    for H in mats:
        H = np.ascontiguousarray(H, np.float32)
        out = np.empty(9, np.float32)
        assert lib.stx_invert_homography(H.ctypes.data_as(fp), out.ctypes.data_as(fp)) == 0
        ref = oracle.invert3x3(H).reshape(9)
        assert out.tobytes() == ref.tobytes()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from stereo_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setenv("STEREO_HIP_LIB", str(tmp_path / "nope.so"))
    try:
        _lib.load_library()
    except ImportError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load_library() must raise when the .so is missing")
    monkeypatch.setattr(_lib, "_LIB", None)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under stereo_amd/ may import,
    include, link or dlopen it."""
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#\s*include\s*[\"<][^\">]*oracle)|liboracle|dlopen",
                     re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "stereo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), "%s uses the oracle" % os.path.join(dirpath, f)
