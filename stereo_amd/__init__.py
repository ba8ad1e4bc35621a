"""stereo_amd -- MI355X-native hot path of dmckinnon/stereo.

Three data-parallel stages behind the reference's own function names:

    MatchDescriptors   stereo/Features.cpp:1020-1079
    RectifyImage       stereo/Stereography.cpp:596-631
    ComputeDepthImage  stereo/Stereography.cpp:649-691

plus the stages either side of them (FAST-12, scoring + suppression, descriptors, RANSAC
hypothesis scoring).

All compute happens in libstereo_hip.so (hand-written HIP for gfx950, C ABI in
include/stereo_hip.h).  There is no CPU fallback: importing the compute entry
points without the built library raises.
"""
from ._lib import StereoHipError, load_library, library_path
from .api import (
    Context,
    MultiContext,
    Feature,
    MatchDescriptors,
    FindFASTFeatures,
    ScoreAndClusterFeatures,
    ClusterFeatures,
    CreateSIFTDescriptors,
    RectifyImage,
    ComputeDepthImage,
    MATCH_FAILED,
    MATCH_SCORE,
    MATCH_RATIO,
    MATCH_ACCEPTED,
    NN_RATIO,
    MAX_DIST_BETWEEN_MATCHES,
)
from .ransac import FindFundamentalMatrixWithRANSAC

__all__ = [
    "StereoHipError", "load_library", "library_path", "Context", "MultiContext", "Feature",
    "MatchDescriptors", "FindFASTFeatures", "ScoreAndClusterFeatures", "ClusterFeatures", "CreateSIFTDescriptors", "FindFundamentalMatrixWithRANSAC", "RectifyImage", "ComputeDepthImage",
    "MATCH_FAILED", "MATCH_SCORE", "MATCH_RATIO", "MATCH_ACCEPTED",
    "NN_RATIO", "MAX_DIST_BETWEEN_MATCHES",
]
