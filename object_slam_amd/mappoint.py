"""MapPoint maintenance + Frame::isInFrustum, batched over map points, over the C ABI.

Mirrors MapPoint::ComputeDistinctiveDescriptors / UpdateNormalAndDepth (reference src/MapPoint.cc:345-474) and
Frame::isInFrustum (reference src/Frame.cc:509-565).  All arithmetic runs in the HIP library."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr
from .matcher import QUERY_DTYPE


def _csr(lists, width, dtype):
    start = np.zeros(len(lists) + 1, np.int32)
    for i, l in enumerate(lists):
        start[i + 1] = start[i] + len(l)
    flat = np.zeros((max(int(start[-1]), 1), width), dtype)
    for i, l in enumerate(lists):
        if len(l):
            flat[start[i]:start[i + 1]] = np.asarray(l, dtype).reshape(-1, width)
    return start, flat


class MapPointBatch:
    def __init__(self, device=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        check(self.L.oslam_mappoint_create(C.byref(self.h), device))

    def close(self):
        if getattr(self, "h", None):
            self.L.oslam_mappoint_destroy(self.h)
            self.h = None

    __del__ = close

    def ComputeDistinctiveDescriptors(self, obs_desc_lists):
        """obs_desc_lists[p] = [n_p][32] descriptors of point p's observations, in the reference's map order.
        Returns (best_idx[P], desc[P][32])."""
        P = len(obs_desc_lists)
        start, flat = _csr(obs_desc_lists, 32, np.uint8)
        best = np.full(max(P, 1), -1, np.int32)
        out = np.zeros((max(P, 1), 32), np.uint8)
        check(self.L.oslam_mp_distinctive_descriptors(self.h, P, ptr(start), ptr(flat), ptr(best), ptr(out)))
        return best[:P], out[:P]

    def UpdateNormalAndDepth(self, Pos, obs_Ow_lists, OwRef, levelScaleFactor, lastScaleFactor):
        """Returns [P][5] = normal(3), mfMaxDistance, mfMinDistance."""
        Pos = np.ascontiguousarray(Pos, np.float32).reshape(-1, 3)
        P = len(Pos)
        start, flat = _csr(obs_Ow_lists, 3, np.float32)
        out = np.zeros((max(P, 1), 5), np.float32)
        check(self.L.oslam_mp_update_normal_depth(self.h, P, ptr(Pos), ptr(start), ptr(flat),
                                                  ptr(np.ascontiguousarray(OwRef, np.float32)),
                                                  ptr(np.ascontiguousarray(levelScaleFactor, np.float32)),
                                                  C.c_float(lastScaleFactor), ptr(out)))
        return out[:P]

    def isInFrustum(self, Pw, Pn, maxDist, minDist, obs_gt0, mp_desc, Tcw, K5, bounds, viewingCosLimit, logScaleFactor,
                    scaleFactors, th=1.0):
        """Batched Frame::isInFrustum; returns QUERY_DTYPE[M] for ORBmatcher.search_window (flags=0: outside)."""
        Pw = np.ascontiguousarray(Pw, np.float32).reshape(-1, 3)
        M = len(Pw)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        out = np.zeros(max(M, 1), QUERY_DTYPE)
        check(self.L.oslam_frame_is_in_frustum(
            self.h, M, ptr(Pw), ptr(np.ascontiguousarray(Pn, np.float32)), ptr(np.ascontiguousarray(maxDist, np.float32)),
            ptr(np.ascontiguousarray(minDist, np.float32)), ptr(np.ascontiguousarray(obs_gt0, np.uint8)),
            ptr(np.ascontiguousarray(mp_desc, np.uint8)), (C.c_float * 16)(*np.asarray(Tcw, np.float32).reshape(-1)),
            (C.c_float * 5)(*np.asarray(K5, np.float32)), (C.c_float * 4)(*np.asarray(bounds, np.float32)),
            C.c_float(viewingCosLimit), C.c_float(logScaleFactor), ptr(sf), len(sf), C.c_float(th), ptr(out)))
        return out[:M]
