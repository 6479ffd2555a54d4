"""Frame construction steps between the extractor and the matchers, over the C ABI: Frame::UndistortKeyPoints,
ComputeImageBounds, ComputeStereoFromRGBD (reference src/Frame.cc:644-704, :883-904).  All arithmetic in the HIP library."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr


class FrameOps:
    def __init__(self, device=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        check(self.L.oslam_frame_create(C.byref(self.h), device))

    def close(self):
        if getattr(self, "h", None):
            self.L.oslam_frame_destroy(self.h)
            self.h = None

    __del__ = close

    @staticmethod
    def _kd(K4, dist):
        k = (C.c_float * 4)(*[float(np.float32(v)) for v in K4])
        d = np.ascontiguousarray(dist if dist is not None else [], np.float32)
        return k, d

    def UndistortKeyPoints(self, keys, K4, dist):
        keys = np.ascontiguousarray(keys, KP_DTYPE)
        out = np.zeros(max(len(keys), 1), KP_DTYPE)
        k, d = self._kd(K4, dist)
        check(self.L.oslam_frame_undistort_keypoints(self.h, len(keys), ptr(keys), k, ptr(d) if len(d) else None, len(d), ptr(out)))
        return out[:len(keys)]

    def ComputeImageBounds(self, cols, rows, K4, dist):
        k, d = self._kd(K4, dist)
        b = (C.c_float * 4)()
        check(self.L.oslam_frame_image_bounds(self.h, cols, rows, k, ptr(d) if len(d) else None, len(d), b))
        return np.array(list(b), np.float32)

    def ComputeStereoFromRGBD(self, keys, keysUn, depth, mbf):
        keys = np.ascontiguousarray(keys, KP_DTYPE)
        keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
        depth = np.ascontiguousarray(depth, np.float32)
        n = len(keys)
        ur, dp = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
        check(self.L.oslam_frame_stereo_from_rgbd(self.h, n, ptr(keys), ptr(keysUn), ptr(depth), depth.shape[0], depth.shape[1], depth.shape[1],
                                                  C.c_float(mbf), ptr(ur), ptr(dp)))
        return ur[:n], dp[:n]
