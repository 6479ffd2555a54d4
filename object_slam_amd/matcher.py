"""ORBmatcher — Python mirror of the projection searches of ORB_SLAM2::ORBmatcher (reference
include/ORBmatcher.h:41-83) over the C ABI.  All arithmetic runs in the HIP library."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr

QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("minLevel", "<i4"),
                        ("maxLevel", "<i4"), ("flags", "<i4"), ("angle", "<f4"), ("desc", "u1", (32,))])
assert QUERY_DTYPE.itemsize == 64


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("b", C.c_float)]


class MatchFrames(C.Structure):
    _fields_ = [("keysUn", C.c_void_p), ("kp_stride", C.c_int), ("uRight", C.c_void_p), ("desc", C.c_void_p),
                ("blocked", C.c_void_p), ("n_kps", C.c_void_p), ("n_kps_const", C.c_int),
                ("minX", C.c_float), ("minY", C.c_float), ("maxX", C.c_float), ("maxY", C.c_float)]


class MatchLast(C.Structure):
    _fields_ = [("Xw", C.c_void_p), ("has_mp", C.c_void_p), ("keys", C.c_void_p), ("mp_desc", C.c_void_p),
                ("kp_stride", C.c_int), ("n_kps", C.c_void_p), ("n_kps_const", C.c_int)]


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30   # reference src/ORBmatcher.cc:37-39

    def __init__(self, nnratio=0.6, checkOri=True, max_keypoints=2400, max_queries=4096, max_batch=1, device=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        check(self.L.oslam_matcher_create(C.byref(self.h), max_batch, max_keypoints, max_queries, device))
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)
        self.max_kps, self.max_q, self.max_batch = max_keypoints, max_queries, max_batch

    def close(self):
        if getattr(self, "h", None):
            self.L.oslam_matcher_destroy(self.h)
            self.h = None

    __del__ = close

    # SearchByProjection(Frame&, const vector<MapPoint*>&, th): src/ORBmatcher.cc:45
    # (use_ratio=True); generic windowed search otherwise.
    def search_window(self, keysUn, uRight, desc, blocked, bounds, queries, use_ratio=True, check_ori=False):
        keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
        N, M = len(keysUn), len(queries)
        desc = np.ascontiguousarray(desc, np.uint8)
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        uR = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
        bnd = (C.c_float * 4)(*bounds)
        qm, qd = np.full(max(M, 1), -1, np.int32), np.full(max(M, 1), 256, np.int32)
        km = np.full(max(N, 1), -1, np.int32)
        nm = C.c_int(0)
        check(self.L.oslam_match_search_by_projection(
            self.h, N, ptr(keysUn), ptr(uR) if uR is not None else None, ptr(desc),
            ptr(bl) if bl is not None else None, bnd, ptr(queries), M, C.c_float(self.mfNNratio), int(use_ratio),
            int(check_ori), ptr(qm), ptr(qd), ptr(km), C.byref(nm)))
        return nm.value, qm[:M], qd[:M], km[:N]

    # SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono): src/ORBmatcher.cc:1328
    def search_last_frame(self, keysUn, uRight, desc, blocked, bounds, Xw, has_mp, last_keys, mp_desc, Tcw, Tlw,
                          cam, scaleFactors, th, bMono):
        keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
        last_keys = np.ascontiguousarray(last_keys, KP_DTYPE)
        N, NL = len(keysUn), len(last_keys)
        desc = np.ascontiguousarray(desc, np.uint8)
        uR = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
        Xw = np.ascontiguousarray(Xw, np.float32)
        has_mp = np.ascontiguousarray(has_mp, np.uint8)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
        Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        Tlw = np.ascontiguousarray(Tlw, np.float32).reshape(16)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        cams = Camera(*[float(x) for x in cam])
        bnd = (C.c_float * 4)(*bounds)
        qm, qd = np.full(max(NL, 1), -1, np.int32), np.full(max(NL, 1), 256, np.int32)
        km = np.full(max(N, 1), -1, np.int32)
        nm = C.c_int(0)
        check(self.L.oslam_match_project_last_frame(
            self.h, N, ptr(keysUn), ptr(uR) if uR is not None else None, ptr(desc),
            ptr(bl) if bl is not None else None, bnd, NL, ptr(Xw), ptr(has_mp), ptr(last_keys), ptr(mp_desc),
            ptr(Tcw), ptr(Tlw), C.byref(cams), ptr(sf), len(sf), C.c_float(th), int(bMono),
            int(self.mbCheckOrientation), ptr(qm), ptr(qd), ptr(km), C.byref(nm)))
        return nm.value, qm[:NL], qd[:NL], km[:N]

    # search half of Fuse(KeyFrame*, vpMapPoints, th): src/ORBmatcher.cc:825
    def fuse_search(self, keysUn, uRight, desc, bounds, queries, invLevelSigma2):
        keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
        N, M = len(keysUn), len(queries)
        desc = np.ascontiguousarray(desc, np.uint8)
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        uR = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        inv = np.ascontiguousarray(invLevelSigma2, np.float32)
        bnd = (C.c_float * 4)(*bounds)
        qm, qd = np.full(max(M, 1), -1, np.int32), np.full(max(M, 1), 256, np.int32)
        nf = C.c_int(0)
        check(self.L.oslam_match_fuse_search(self.h, N, ptr(keysUn), ptr(uR) if uR is not None else None, ptr(desc), bnd,
                                             ptr(queries), M, ptr(inv), len(inv), ptr(qm), ptr(qd), C.byref(nf)))
        return nf.value, qm[:M], qd[:M]

    def debug_queries(self, n, b=0, q_stride=None):
        out = np.zeros(max(n, 1), QUERY_DTYPE)
        check(self.L.oslam_match_debug_get_queries(self.h, b, q_stride or self.max_q, n, ptr(out)))
        return out[:n]

    # ---- batch mode, everything in HBM ----
    def search_batch_device(self, frames, d_queries, q_stride, d_nq, nq_const, batch, use_ratio, check_ori, stream=None):
        check(self.L.oslam_match_search_batch_device(
            self.h, C.byref(frames), C.c_void_p(d_queries or 0), q_stride, C.c_void_p(d_nq or 0), nq_const, batch,
            C.c_float(self.mfNNratio), int(use_ratio), int(check_ori), self.TH_HIGH, C.c_void_p(stream or 0)))

    def project_last_batch_device(self, last, d_Tcw, d_Tlw, cam, cur, scaleFactors, th, bMono, batch, stream=None):
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        cams = Camera(*[float(x) for x in cam])
        check(self.L.oslam_match_project_last_batch_device(
            self.h, C.byref(last), C.c_void_p(d_Tcw), C.c_void_p(d_Tlw), C.byref(cams), C.byref(cur), ptr(sf), len(sf),
            C.c_float(th), int(bMono), batch, C.c_void_p(stream or 0)))

    def fetch(self, b, q_stride, n_q, kp_stride, n_kps, stream=None):
        qm, qd = np.full(max(n_q, 1), -1, np.int32), np.full(max(n_q, 1), 256, np.int32)
        km = np.full(max(n_kps, 1), -1, np.int32)
        nm, it = C.c_int(0), C.c_int(0)
        check(self.L.oslam_match_fetch(self.h, b, q_stride, n_q, kp_stride, n_kps, ptr(qm), ptr(qd), ptr(km),
                                       C.byref(nm), C.byref(it), C.c_void_p(stream or 0)))
        return nm.value, qm[:n_q], qd[:n_q], km[:n_kps], it.value


class StereoMatcher:
    """Frame::ComputeStereoMatches (reference src/Frame.cc:706-880) on two ORBextractor handles."""

    def __init__(self, max_keypoints=2400, max_batch=1, device=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        check(self.L.oslam_stereo_create(C.byref(self.h), max_batch, max_keypoints, device))

    def close(self):
        if getattr(self, "h", None):
            self.L.oslam_stereo_destroy(self.h)
            self.h = None

    __del__ = close

    def ComputeStereoMatches(self, orbL, orbR, keysL, descL, keysR, descR, bf, b):
        """orbL / orbR: ORBextractor objects that just extracted the pair. Returns (mvuRight, mvDepth)."""
        keysL = np.ascontiguousarray(keysL, KP_DTYPE)
        keysR = np.ascontiguousarray(keysR, KP_DTYPE)
        N = len(keysL)
        uR = np.zeros(max(N, 1), np.float32)
        dep = np.zeros(max(N, 1), np.float32)
        check(self.L.oslam_stereo_match(self.h, orbL.h, orbR.h, N, ptr(keysL), ptr(np.ascontiguousarray(descL, np.uint8)),
                                        len(keysR), ptr(keysR), ptr(np.ascontiguousarray(descR, np.uint8)), orbL.nlevels,
                                        C.c_float(bf), C.c_float(b), ptr(uR), ptr(dep)))
        return uR[:N], dep[:N]


class BowSide1(C.Structure):
    _fields_ = [("N", C.c_int32), ("keys", C.c_void_p), ("desc", C.c_void_p), ("uRight", C.c_void_p), ("flag", C.c_void_p),
                ("nq", C.c_int32), ("q_idx", C.c_void_p), ("q_node", C.c_void_p)]


class BowSide2(C.Structure):
    _fields_ = [("N", C.c_int32), ("keys", C.c_void_p), ("desc", C.c_void_p), ("uRight", C.c_void_p), ("has_mp", C.c_void_p),
                ("nNodes", C.c_int32), ("nodes", C.c_void_p), ("start", C.c_void_p), ("items", C.c_void_p)]


def feature_vector(node_of_kp):
    """node id per keypoint -> (flat (idx, node) list in std::map order, CSR (nodes, start, items))."""
    node_of_kp = np.asarray(node_of_kp, np.uint32)
    order = np.argsort(node_of_kp, kind="stable").astype(np.int32)
    nodes, start = np.unique(node_of_kp[order], return_index=True)
    start = np.concatenate([start, [len(order)]]).astype(np.int32)
    return order, node_of_kp[order].astype(np.uint32), nodes.astype(np.uint32), start, order.copy()


class BowMatcher:
    """SearchByBoW(KeyFrame*, Frame&) and SearchForTriangulation (reference src/ORBmatcher.cc:159, :657) with
    caller-supplied FeatureVectors (the DBoW2 vocabulary is not part of the reference tree)."""

    def __init__(self, max_keypoints=2400, device=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        check(self.L.oslam_bow_create(C.byref(self.h), max_keypoints, device))

    def close(self):
        if getattr(self, "h", None):
            self.L.oslam_bow_destroy(self.h)
            self.h = None

    __del__ = close

    @staticmethod
    def _sides(keys1, desc1, uR1, flag1, node1, keys2, desc2, uR2, mp2, node2):
        keep = []
        k1 = np.ascontiguousarray(keys1, KP_DTYPE); d1 = np.ascontiguousarray(desc1, np.uint8)
        k2 = np.ascontiguousarray(keys2, KP_DTYPE); d2 = np.ascontiguousarray(desc2, np.uint8)
        f1 = np.ascontiguousarray(flag1, np.uint8)
        qi, qn, _, _, _ = feature_vector(node1)
        _, _, nodes, start, items = feature_vector(node2)
        keep += [k1, d1, k2, d2, f1, qi, qn, nodes, start, items]
        s1, s2 = BowSide1(), BowSide2()
        s1.N, s1.keys, s1.desc, s1.flag = len(k1), k1.ctypes.data, d1.ctypes.data, f1.ctypes.data
        s1.nq, s1.q_idx, s1.q_node = len(qi), qi.ctypes.data, qn.ctypes.data
        s2.N, s2.keys, s2.desc = len(k2), k2.ctypes.data, d2.ctypes.data
        s2.nNodes, s2.nodes, s2.start, s2.items = len(nodes), nodes.ctypes.data, start.ctypes.data, items.ctypes.data
        if uR1 is not None:
            a = np.ascontiguousarray(uR1, np.float32); keep.append(a); s1.uRight = a.ctypes.data
        if uR2 is not None:
            a = np.ascontiguousarray(uR2, np.float32); keep.append(a); s2.uRight = a.ctypes.data
        if mp2 is not None:
            a = np.ascontiguousarray(mp2, np.uint8); keep.append(a); s2.has_mp = a.ctypes.data
        return s1, s2, keep

    def SearchByBoW(self, keysKF, descKF, validKF, nodeKF, keysF, descF, nodeF, nnratio=0.7, checkOri=True):
        s1, s2, keep = self._sides(keysKF, descKF, None, validKF, nodeKF, keysF, descF, None, None, nodeF)
        out = np.full(max(s2.N, 1), -1, np.int32)
        nm = C.c_int(0)
        check(self.L.oslam_match_search_by_bow(self.h, C.byref(s1), C.byref(s2), C.c_float(nnratio), int(checkOri), ptr(out), C.byref(nm)))
        return nm.value, out[:s2.N]

    def SearchForTriangulation(self, keys1, desc1, uR1, hasmp1, node1, keys2, desc2, uR2, hasmp2, node2, F12, ex, ey, scaleFactors,
                               levelSigma2, bOnlyStereo=False, checkOri=True):
        s1, s2, keep = self._sides(keys1, desc1, uR1, hasmp1, node1, keys2, desc2, uR2, hasmp2, node2)
        out = np.full(max(s1.N, 1), -1, np.int32)
        nm = C.c_int(0)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf = np.ascontiguousarray(scaleFactors, np.float32); s2l = np.ascontiguousarray(levelSigma2, np.float32)
        check(self.L.oslam_match_search_for_triangulation(self.h, C.byref(s1), C.byref(s2), ptr(F), C.c_float(ex), C.c_float(ey), ptr(sf), ptr(s2l),
                                                          len(sf), int(bOnlyStereo), int(checkOri), ptr(out), C.byref(nm)))
        return nm.value, out[:s1.N]
