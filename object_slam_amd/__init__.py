"""MI355X-native (gfx950) front-end + local-BA hot path of yangliu9527/Object_SLAM.

Thin Python view of the C ABI in include/oslam_hip.h (the product is the HIP library;
Python is harness).  Class and method names mirror the reference's C++ API.
"""
from ._lib import KP_DTYPE, OslamError  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import BowMatcher, ORBmatcher, QUERY_DTYPE, StereoMatcher, feature_vector  # noqa: F401
from .frame import FrameOps  # noqa: F401
from .mappoint import MapPointBatch  # noqa: F401
from .optimizer import LocalBundleAdjuster, PoseOptimizer  # noqa: F401
