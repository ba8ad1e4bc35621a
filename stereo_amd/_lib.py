"""ctypes binding of libstereo_hip.so (C ABI: include/stereo_hip.h).

The library is the product; this module only declares its prototypes.  If the
shared object is missing the loader raises -- there is no fallback path.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class StereoHipError(RuntimeError):
    """A libstereo_hip entry point returned a non-zero status."""

    def __init__(self, code, where):
        self.code = code
        msg = "?"
        try:
            msg = load_library().stx_error_string(code).decode()
        except Exception:  # pragma: no cover - only when the library is absent
            pass
        super().__init__("%s failed with status %d: %s" % (where, code, msg))


def library_path():
    return os.environ.get("STEREO_HIP_LIB", os.path.join(_HERE, "libstereo_hip.so"))


# name -> (restype, argtypes); kept in one table so tests can check that every
# symbol declared in include/stereo_hip.h is exported and bound.
_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p
PROTOTYPES = {
    "stx_version": (C.c_int, []),
    "stx_error_string": (C.c_char_p, [C.c_int]),
    "stx_device_count": (C.c_int, []),
    "stx_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "stx_destroy": (None, [_vp]),
    "stx_set_stream": (C.c_int, [_vp, _vp]),
    "stx_synchronize": (C.c_int, [_vp]),
    "stx_set_match_algo": (C.c_int, [_vp, C.c_int]),
    "stx_workspace_bytes": (C.c_size_t, [_vp]),
    "stx_staging_bytes": (C.c_size_t, [_vp]),
    "stx_host_buffer_kind": (C.c_int, [_vp, _vp, C.c_size_t]),
    "stx_match_stats": (C.c_int, [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]),
    "stx_set_disparity_algo": (C.c_int, [_vp, C.c_int]),
    "stx_set_host_pipeline": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "stx_multi_create": (C.c_int, [C.POINTER(_vp), C.POINTER(C.c_int), C.c_int]),
    "stx_multi_destroy": (None, [_vp]),
    "stx_multi_size": (C.c_int, [_vp]),
    "stx_multi_context": (_vp, [_vp, C.c_int]),
    "stx_disparity_batch_u8": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                         C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t]),
    "stx_disparity_rowband_u8": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                           C.c_int, _vp, C.c_size_t]),
    "stx_disparity_rowband_u16": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                            C.c_int, _vp, C.c_size_t]),
    "stx_match_multi": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_float,
                                  _vp, _vp, _vp, _vp, _vp]),
    "stx_match": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_float,
                            _vp, _vp, _vp, _vp, _vp]),
    "stx_match_features": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float,
                                     _vp, _vp, _vp, _vp, _vp]),
    "stx_match_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_float,
                                _vp, _vp, _vp, _vp, _vp]),
    "stx_invert_homography": (C.c_int, [_f32p, _f32p]),
    "stx_rectify_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int,
                                 C.c_int, C.c_size_t, _f32p, C.c_int]),
    "stx_rectify_multi_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int,
                                       C.c_int, C.c_size_t, _f32p, C.c_int]),
    "stx_rectify_u8_keep": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int,
                                      C.c_int, C.c_size_t, _f32p, C.c_int]),
    "stx_disparity_resident_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            _vp, C.c_size_t]),
    "stx_rectify_u8_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                     _vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, _f32p,
                                     C.c_int, C.c_int]),
    "stx_rectify_rowband_u8_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int,
                                             C.c_int, C.c_int, C.c_size_t, _f32p, C.c_int]),
    "stx_disparity_u8": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                   C.c_int, C.c_int, _vp, C.c_size_t]),
    "stx_disparity_u16": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                    C.c_int, C.c_int, _vp, C.c_size_t]),
    "stx_disparity_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                    C.c_size_t, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t,
                                    C.c_size_t, C.c_int]),
    "stx_disparity_dev_noalloc": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                            C.c_size_t, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t,
                                            C.c_size_t, C.c_int]),
    "stx_fast_detect_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, C.c_int, _vp, C.c_int,
                                     C.POINTER(C.c_int)]),
    "stx_fast_detect_u8_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, C.c_int, _vp, C.c_int, _vp]),
    "stx_score_cluster": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _f32p, C.c_float,
                                    C.c_float, _vp, _vp, _vp, C.POINTER(C.c_int)]),
    "stx_score_cluster_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _f32p,
                                        C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "stx_cluster_features": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.POINTER(C.c_int)]),
    "stx_cluster_features_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "stx_sift_descriptors": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _vp, _vp]),
    "stx_sift_descriptors_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, _vp, C.c_int, _vp, _vp]),
    "stx_ransac_score": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_float, _vp, _vp, _vp,
                                   _vp]),
    "stx_inlier_sum_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp]),
    "stx_ransac_score_dev": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_float, _vp, _vp,
                                       _vp, _vp]),
    "stx_gaussian_kernel": (C.c_int, [C.c_int, C.c_float, _f32p]),
    "stx_rectify_disparity_u8": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_size_t, _f32p, _f32p,
                                           C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp,
                                           C.c_size_t]),
    "stx_disparity_rowband_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, _vp, C.c_size_t, C.c_int]),
}


def _bind_to_torchs_hip_runtime():
    """One HIP runtime per process.

    libstereo_hip.so needs `libamdhip64.so.7` (RUNPATH /opt/rocm); torch's wheel ships its own
    runtime as torch/lib/libamdhip64.so -- a different FILE NAME with the SAME soname
    (libamdhip64.so.7), plus its own libhsa-runtime64.  The dynamic loader matches a request
    against the sonames of what is already loaded, so:
      torch first:  our request for libamdhip64.so.7 resolves to torch's copy -- one runtime;
      ours first:   /opt/rocm's copy is loaded, and torch's later request for the file name
                    `libamdhip64.so` matches nothing loaded, so a SECOND HIP runtime and a second
                    HSA runtime come up in the process and torch finds "No HIP GPUs".
    Loading torch's copy (globally) before ours makes both orders the first case.  It only
    locates the file; torch itself is not imported.  STEREO_HIP_RUNTIME=system skips this (a
    process that will never load torch); a host program without torch gets /opt/rocm's runtime
    through the RUNPATH as usual."""
    if os.environ.get("STEREO_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


def load_library():
    """Load libstereo_hip.so once and attach prototypes.  Raises if it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            "libstereo_hip.so not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C stereo_amd/csrc`.  stereo_amd has no CPU fallback." % path)
    _bind_to_torchs_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(code, where):
    if code != 0:
        raise StereoHipError(code, where)
