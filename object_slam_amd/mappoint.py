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


class TriKF(C.Structure):
    """oslam_tri_kf_t"""
    _fields_ = [("Tcw", C.c_float * 16), ("Twc", C.c_float * 16), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("invfx", C.c_float), ("invfy", C.c_float), ("mbf", C.c_float), ("mb", C.c_float),
                ("keysUn", C.c_void_p), ("keys", C.c_void_p), ("uRight", C.c_void_p), ("depth", C.c_void_p), ("n_kps", C.c_int32)]


def make_tri_kf(Tcw, Twc, cam8, keysUn, keys, uRight, depth):
    """cam8 = fx, fy, cx, cy, invfx, invfy, mbf, mb.  Returns (struct, keepalive)."""
    from ._lib import KP_DTYPE
    keep = [np.ascontiguousarray(keysUn, KP_DTYPE), np.ascontiguousarray(keys, KP_DTYPE), np.ascontiguousarray(uRight, np.float32),
            np.ascontiguousarray(depth, np.float32)]
    k = TriKF()
    k.Tcw[:] = [float(v) for v in np.asarray(Tcw, np.float32).reshape(-1)]
    k.Twc[:] = [float(v) for v in np.asarray(Twc, np.float32).reshape(-1)]
    k.fx, k.fy, k.cx, k.cy, k.invfx, k.invfy, k.mbf, k.mb = [float(np.float32(v)) for v in cam8]
    k.keysUn, k.keys, k.uRight, k.depth = [a.ctypes.data for a in keep]
    k.n_kps = len(keep[0])
    return k, keep


def triangulate(self, kf1, kf2_list, matches, scaleFactors, levelSigma2, ratioFactor):
    """LocalMapping::CreateNewMapPoints numeric core.  kf1 / kf2_list entries from make_tri_kf; matches[p] = (idx1[], idx2[]).
    Returns (ok[M] u8, x3D[M][3]) over the concatenated matches."""
    nP = len(kf2_list)
    start = np.zeros(nP + 1, np.int32)
    for p, (a, _) in enumerate(matches):
        start[p + 1] = start[p] + len(a)
    M = int(start[-1])
    i1 = np.concatenate([np.asarray(a, np.int32) for a, _ in matches] + [np.zeros(0, np.int32)]).astype(np.int32)
    i2 = np.concatenate([np.asarray(b, np.int32) for _, b in matches] + [np.zeros(0, np.int32)]).astype(np.int32)
    arr = (TriKF * max(nP, 1))(*[k for k, _ in kf2_list])
    sf = np.ascontiguousarray(scaleFactors, np.float32)
    ls = np.ascontiguousarray(levelSigma2, np.float32)
    ok = np.zeros(max(M, 1), np.uint8)
    x = np.zeros((max(M, 1), 3), np.float32)
    nnew = C.c_int32(0)
    check(self.L.oslam_mp_triangulate(self.h, C.byref(kf1[0]), nP, arr, ptr(start), ptr(i1) if M else None, ptr(i2) if M else None,
                                      ptr(sf), ptr(ls), len(sf), C.c_float(ratioFactor), ptr(ok), ptr(x), C.byref(nnew)))
    assert nnew.value == int(ok[:M].sum())
    return ok[:M], x[:M]


MapPointBatch.triangulate = triangulate
