"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/_build/liboslam_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboslam_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build(force=False):
    # make decides what is stale (the operator-table sources include the product's C ABI headers); where make or the sources are missing the prebuilt file is used
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    except (OSError, subprocess.CalledProcessError):
        if not os.path.exists(_SO):
            raise
    return _SO


def native_lib():
    """bench.py's CPU baseline: the oracle built -march=native ON THE HOST IT IS TIMED ON (make native); falls back to the portable build.  Returns (CDLL, march)."""
    so = os.path.join(_HERE, "_build", "liboslam_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native"])
        return C.CDLL(so), "native"
    except (OSError, subprocess.CalledProcessError):
        return lib(), "x86-64-v3 (the native build failed on this host)"


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oo_orb_create.restype = C.c_void_p
        _lib.oo_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _lib.oo_orb_destroy.argtypes = [C.c_void_p]
        _lib.oo_fast_atan2.restype = C.c_float
        _lib.oo_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.oo_brief_pattern.restype = C.POINTER(C.c_int8)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbExtractor:
    """Oracle mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-110)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = C.c_void_p(self.L.oo_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oo_orb_destroy(self.h)
            self.h = None

    def set_blur_sse2(self, on):
        self.L.oo_orb_set_blur_sse2(self.h, int(on))

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.oo_orb_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(nf), _p(um))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, nfeatures_per_level=nf, umax=um)

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = self.nfeatures * 4 + 1024
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rc = self.L.oo_orb_extract(self.h, _p(img), w, h, w, _p(kps), _p(desc), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("oracle extract rc=%d" % rc)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        self.L.oo_orb_level_size(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def level(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        self.L.oo_orb_get_level(self.h, level, _p(out))
        return out

    def blurred(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        rc = self.L.oo_orb_get_blurred(self.h, level, _p(out))
        return out if rc == 0 else None

    def candidates(self, level):
        n = self.L.oo_orb_num_candidates(self.h, level)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.oo_orb_get_candidates(self.h, level, _p(out))
        return out[:n]

    def level_keys(self, level):
        n = self.L.oo_orb_num_level_keys(self.h, level)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.oo_orb_get_level_keys(self.h, level, _p(out))
        return out[:n]

    def distribute_octree(self, cand, minX, maxX, minY, maxY, N):
        cand = np.ascontiguousarray(cand, dtype=KP_DTYPE)
        out = np.zeros(len(cand) + 16, KP_DTYPE)
        n = self.L.oo_distribute_octree(self.h, _p(cand), len(cand), minX, maxX, minY, maxY, N, _p(out), len(out))
        assert n >= 0
        return out[:n]


def fast_9_16(roi, threshold, nms=True):
    roi = np.ascontiguousarray(roi, dtype=np.uint8)
    rows, cols = roi.shape
    out = np.zeros(rows * cols + 1, KP_DTYPE)
    n = lib().oo_fast_9_16(_p(roi), cols, cols, rows, threshold, int(nms), _p(out), len(out))
    assert n >= 0
    return out[:n]


def resize_linear_u8(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], _p(dst), dw, dh)
    return dst


def gaussian_blur(src, sse2=True):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros_like(src)
    lib().oo_gaussian_blur(_p(src), src.shape[1], src.shape[0], _p(dst), int(sse2))
    return dst


def fast_atan2(y, x):
    return float(lib().oo_fast_atan2(float(y), float(x)))


def brief_pattern():
    p = lib().oo_brief_pattern()
    return np.ctypeslib.as_array(p, shape=(1024,)).copy()


# ---------------------------------------------------------------------------------------------
QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("minLevel", "<i4"),
                        ("maxLevel", "<i4"), ("flags", "<i4"), ("angle", "<f4"), ("desc", "u1", (32,))])


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().oo_descriptor_distance(_p(a), _p(b))


def search_by_projection(keysUn, uRight, desc, blocked, bounds, queries, nnratio, use_ratio, check_ori):
    keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
    N, M = len(keysUn), len(queries)
    uR = np.full(N, -1, np.float32) if uRight is None else np.ascontiguousarray(uRight, np.float32)
    bl = np.zeros(N, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8)
    desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    bnd = np.asarray(bounds, np.float32)
    qm, qd = np.full(max(M, 1), -1, np.int32), np.full(max(M, 1), 256, np.int32)
    km = np.full(max(N, 1), -1, np.int32)
    L = lib()
    L.oo_search_by_projection.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 3
    nm = L.oo_search_by_projection(N, _p(keysUn), _p(uR), _p(desc), _p(bl), _p(bnd), _p(queries), M, nnratio,
                                   int(use_ratio), int(check_ori), _p(qm), _p(qd), _p(km))
    return nm, qm[:M], qd[:M], km[:N]


def project_last_frame(Xw, has_mp, keys, mp_desc, Tcw, Tlw, cam, bounds, scaleFactors, th, bMono):
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    N = len(keys)
    out = np.zeros(max(N, 1), QUERY_DTYPE)
    L = lib()
    L.oo_project_last_frame.argtypes = [C.c_int] + [C.c_void_p] * 9 + [C.c_float, C.c_int, C.c_void_p]
    L.oo_project_last_frame(N, _p(np.ascontiguousarray(Xw, np.float32)), _p(np.ascontiguousarray(has_mp, np.uint8)),
                            _p(keys), _p(np.ascontiguousarray(mp_desc, np.uint8)),
                            _p(np.ascontiguousarray(Tcw, np.float32)), _p(np.ascontiguousarray(Tlw, np.float32)),
                            _p(np.asarray(cam, np.float32)), _p(np.asarray(bounds, np.float32)),
                            _p(np.ascontiguousarray(scaleFactors, np.float32)), th, int(bMono), _p(out))
    return out[:N]


def features_in_area(keysUn, bounds, x, y, r, minLevel=-1, maxLevel=-1):
    keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
    out = np.zeros(len(keysUn) + 1, np.int32)
    L = lib()
    L.oo_features_in_area.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int]
    n = L.oo_features_in_area(len(keysUn), _p(keysUn), _p(np.asarray(bounds, np.float32)), x, y, r, minLevel, maxLevel,
                              _p(out), len(out))
    return out[:n]


# ---------------------------------------------------------------------------------------------
def pose_optimization(Tcw, Xw, obs, invSigma2, has_mp, K5):
    """Oracle Optimizer::PoseOptimization. Returns (n_inliers, Tcw_out[4,4] f32, outlier u8[N], (its, trials))."""
    Xw = np.ascontiguousarray(Xw, np.float32)
    N = len(Xw)
    obs = np.ascontiguousarray(obs, np.float32)
    inv = np.ascontiguousarray(invSigma2, np.float32)
    has = np.ascontiguousarray(has_mp, np.uint8)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    K = np.asarray(K5, np.float32)
    out = np.zeros(16, np.float32)
    outl = np.zeros(max(N, 1), np.uint8)
    stats = np.zeros(2, np.int32)
    n = lib().oo_pose_optimization(N, _p(T), _p(Xw), _p(obs), _p(inv), _p(has), _p(K), _p(out), _p(outl), _p(stats))
    return n, out.reshape(4, 4), outl[:N], tuple(stats)


def lm_trace(fn, cap=512):
    """Runs fn() with the LM trace of the oracle's graph optimiser on; returns (fn's result, trace[n, 6]) with rows
    (F before the trial, F of the trial, rho, lambda of the trial, accepted, first trial of a round)."""
    buf = np.zeros((cap, 6), np.float64)
    L = lib()
    L.oo_lm_trace.argtypes = [C.c_void_p, C.c_int]
    L.oo_lm_trace.restype = C.c_int
    L.oo_lm_trace(_p(buf), cap)
    try:
        r = fn()
    finally:
        n = L.oo_lm_trace(None, 0)
    return r, buf[:min(n, cap)].copy()


def local_bundle_adjustment(poses, fixed, points, edge_kf, edge_pt, edge_obs, edge_inv, K5, stop=0):
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    fixed = np.ascontiguousarray(fixed, np.uint8)
    points = np.ascontiguousarray(points, np.float32)
    ekf = np.ascontiguousarray(edge_kf, np.int32)
    ept = np.ascontiguousarray(edge_pt, np.int32)
    eobs = np.ascontiguousarray(edge_obs, np.float32)
    einv = np.ascontiguousarray(edge_inv, np.float32)
    K = np.asarray(K5, np.float32)
    pout = np.zeros_like(poses)
    xout = np.zeros_like(points)
    erase = np.zeros(max(len(ekf), 1), np.uint8)
    stats = np.zeros(4, np.int32)
    st = np.array([stop], np.int32)
    lib().oo_local_bundle_adjustment(len(poses), _p(poses), _p(fixed), len(points), _p(points), len(ekf), _p(ekf), _p(ept),
                                     _p(eobs), _p(einv), _p(K), _p(st), _p(pout), _p(xout), _p(erase), _p(stats))
    return pout.reshape(-1, 4, 4), xout, erase[:len(ekf)], tuple(stats)


def se3_exp_mul(update6, T):
    out = np.zeros(16, np.float32)
    lib().oo_se3_exp_mul(_p(np.ascontiguousarray(update6, np.float64)), _p(np.ascontiguousarray(T, np.float32).reshape(16)), _p(out))
    return out.reshape(4, 4)


def edge_eval(T, X, obs, stereo, binary, K5):
    err, Jp, Jx = np.zeros(3), np.zeros(18), np.zeros(9)
    lib().oo_edge_eval(_p(np.ascontiguousarray(T, np.float32).reshape(16)), _p(np.ascontiguousarray(X, np.float64)),
                       _p(np.ascontiguousarray(obs, np.float64)), int(stereo), int(binary), _p(np.asarray(K5, np.float64)),
                       _p(err), _p(Jp), _p(Jx))
    D = 3 if stereo else 2
    return err[:D], Jp.reshape(3, 6)[:D], Jx.reshape(3, 3)[:D]


def stereo_matches(orbL, orbR, keysL, descL, keysR, descR, bf, b):
    """Oracle Frame::ComputeStereoMatches; orbL/orbR are OrbExtractor objects that just extracted the pair."""
    keysL = np.ascontiguousarray(keysL, KP_DTYPE)
    keysR = np.ascontiguousarray(keysR, KP_DTYPE)
    N = len(keysL)
    uR = np.zeros(max(N, 1), np.float32)
    dep = np.zeros(max(N, 1), np.float32)
    L = lib()
    L.oo_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.oo_stereo_matches(orbL.h, orbR.h, N, _p(keysL), _p(np.ascontiguousarray(descL, np.uint8)), len(keysR), _p(keysR),
                        _p(np.ascontiguousarray(descR, np.uint8)), bf, b, _p(uR), _p(dep))
    return uR[:N], dep[:N]


def pose_optimization2(p):
    """Oracle ObjectOptimizer::PoseOptimization2 on a synth.make_semantic_problem dict.
    Returns (n_inliers, Tcw_out, outlier, nSemNum)."""
    Xw = np.ascontiguousarray(p["Xw"], np.float32)
    N = len(Xw)
    masks = np.ascontiguousarray(p["masks"], np.uint8)
    nObj, H, W = masks.shape
    out = np.zeros(16, np.float32)
    outl = np.zeros(max(N, 1), np.uint8)
    nsem = C.c_int(0)
    L = lib()
    L.oo_pose_optimization2.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.oo_pose_optimization2(N, _p(np.ascontiguousarray(p["Tcw"], np.float32).reshape(16)), _p(Xw),
                                _p(np.ascontiguousarray(p["obs"], np.float32)), _p(np.ascontiguousarray(p["invSigma2"], np.float32)),
                                _p(np.ascontiguousarray(p["has_mp"], np.uint8)), _p(np.asarray(p["K"], np.float32)), nObj, H, W, _p(masks),
                                len(p["objmp_obj"]), _p(np.ascontiguousarray(p["objmp_Xw"], np.float32)),
                                _p(np.ascontiguousarray(p["objmp_obj"], np.int32)), len(p["joint_kp"]),
                                _p(np.ascontiguousarray(p["joint_kp"], np.int32)), _p(np.ascontiguousarray(p["joint_obj"], np.int32)),
                                _p(np.ascontiguousarray(p["kp_uv"], np.float32)), _p(np.asarray(p["bounds"], np.float32)),
                                float(p["invSigma2_0"]), _p(out), _p(outl), C.byref(nsem))
    return n, out.reshape(4, 4), outl[:N], nsem.value


def object_kp_test(keysUn, masks):
    """Oracle keypoint test of Frame::BuildObject2DsRGBD: bit o of out[k] = the 20x20 window around keysUn[k] lies inside mask o (masks [n,H,W] uint8)."""
    keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
    masks = np.ascontiguousarray(masks, np.uint8)
    n, H, W = masks.shape
    ptrs = (C.c_void_p * max(n, 1))(*[masks[i].ctypes.data for i in range(n)])
    out = np.zeros(max(len(keysUn), 1), np.uint8)
    L = lib()
    L.oo_object_kp_test.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.oo_object_kp_test.restype = None
    L.oo_object_kp_test(len(keysUn), _p(keysUn), n, ptrs, H, W, W, _p(out))
    return out[:len(keysUn)]


def fuse_search(keysUn, uRight, desc, bounds, queries, invLevelSigma2):
    keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
    N, M = len(keysUn), len(queries)
    uR = np.full(N, -1, np.float32) if uRight is None else np.ascontiguousarray(uRight, np.float32)
    qm, qd = np.full(max(M, 1), -1, np.int32), np.full(max(M, 1), 256, np.int32)
    L = lib()
    L.oo_fuse_search.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3
    n = L.oo_fuse_search(N, _p(keysUn), _p(uR), _p(np.ascontiguousarray(desc, np.uint8)), _p(np.asarray(bounds, np.float32)),
                         _p(np.ascontiguousarray(queries, QUERY_DTYPE)), M, _p(np.ascontiguousarray(invLevelSigma2, np.float32)), _p(qm), _p(qd))
    return n, qm[:M], qd[:M]


def bundle_adjustment(poses, fixed, points, edge_kf, edge_pt, edge_obs, edge_inv, K5, nIterations, bRobust):
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    points = np.ascontiguousarray(points, np.float32)
    ekf = np.ascontiguousarray(edge_kf, np.int32)
    pout, xout = np.zeros_like(poses), np.zeros_like(points)
    lib().oo_bundle_adjustment(len(poses), _p(poses), _p(np.ascontiguousarray(fixed, np.uint8)), len(points), _p(points), len(ekf), _p(ekf),
                               _p(np.ascontiguousarray(edge_pt, np.int32)), _p(np.ascontiguousarray(edge_obs, np.float32)),
                               _p(np.ascontiguousarray(edge_inv, np.float32)), _p(np.asarray(K5, np.float32)), int(nIterations), int(bRobust),
                               _p(pout), _p(xout))
    return pout.reshape(-1, 4, 4), xout


def _fv(node_of_kp):
    node_of_kp = np.asarray(node_of_kp, np.uint32)
    order = np.argsort(node_of_kp, kind="stable").astype(np.int32)
    nodes, start = np.unique(node_of_kp[order], return_index=True)
    return order, node_of_kp[order].astype(np.uint32), nodes.astype(np.uint32), np.concatenate([start, [len(order)]]).astype(np.int32), order.copy()


def search_by_bow(keysKF, descKF, validKF, nodeKF, keysF, descF, nodeF, nnratio, checkOri):
    k1, k2 = np.ascontiguousarray(keysKF, KP_DTYPE), np.ascontiguousarray(keysF, KP_DTYPE)
    qi, qn, _, _, _ = _fv(nodeKF)
    _, _, nodes, start, items = _fv(nodeF)
    out = np.full(max(len(k2), 1), -1, np.int32)
    L = lib()
    L.oo_search_by_bow.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_float, C.c_int, C.c_void_p]
    n = L.oo_search_by_bow(len(qi), _p(qi), _p(qn), _p(k1), _p(np.ascontiguousarray(descKF, np.uint8)), _p(np.ascontiguousarray(validKF, np.uint8)),
                           len(k2), _p(k2), _p(np.ascontiguousarray(descF, np.uint8)), len(nodes), _p(nodes), _p(start), _p(items),
                           nnratio, int(checkOri), _p(out))
    return n, out[:len(k2)]


def search_for_triangulation(keys1, desc1, uR1, hasmp1, node1, keys2, desc2, uR2, hasmp2, node2, F12, ex, ey, scaleFactors, levelSigma2,
                             bOnlyStereo, checkOri):
    k1, k2 = np.ascontiguousarray(keys1, KP_DTYPE), np.ascontiguousarray(keys2, KP_DTYPE)
    qi, qn, _, _, _ = _fv(node1)
    _, _, nodes, start, items = _fv(node2)
    out = np.full(max(len(k1), 1), -1, np.int32)
    L = lib()
    L.oo_search_for_triangulation.argtypes = ([C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 +
                                              [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p])
    n = L.oo_search_for_triangulation(len(qi), _p(qi), _p(qn), len(k1), _p(k1), _p(np.ascontiguousarray(desc1, np.uint8)),
                                      _p(np.ascontiguousarray(uR1, np.float32)), _p(np.ascontiguousarray(hasmp1, np.uint8)), len(k2), _p(k2),
                                      _p(np.ascontiguousarray(desc2, np.uint8)), _p(np.ascontiguousarray(uR2, np.float32)),
                                      _p(np.ascontiguousarray(hasmp2, np.uint8)), len(nodes), _p(nodes), _p(start), _p(items),
                                      _p(np.ascontiguousarray(F12, np.float32).reshape(9)), ex, ey, _p(np.ascontiguousarray(scaleFactors, np.float32)),
                                      _p(np.ascontiguousarray(levelSigma2, np.float32)), int(bOnlyStereo), int(checkOri), _p(out))
    return n, out[:len(k1)]


def distinctive_descriptor(desc):
    """MapPoint::ComputeDistinctiveDescriptors for one point: index of the chosen observation (-1 if none)."""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    L = lib()
    L.oo_distinctive_descriptor.argtypes = [C.c_int, C.c_void_p]
    return L.oo_distinctive_descriptor(len(desc), _p(desc)) if len(desc) else -1


def update_normal_depth(Pos, Ow, OwRef, levelScaleFactor, lastScaleFactor):
    """MapPoint::UpdateNormalAndDepth for one point -> [nx, ny, nz, maxD, minD]."""
    Ow = np.ascontiguousarray(Ow, np.float32).reshape(-1, 3)
    out = np.zeros(5, np.float32)
    L = lib()
    L.oo_update_normal_depth.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    L.oo_update_normal_depth(_p(np.ascontiguousarray(Pos, np.float32)), len(Ow), _p(Ow), _p(np.ascontiguousarray(OwRef, np.float32)),
                             float(levelScaleFactor), float(lastScaleFactor), _p(out))
    return out


def is_in_frustum(Pw, Pn, maxDist, minDist, obs_gt0, mp_desc, Tcw, K5, bounds, viewingCosLimit, logScaleFactor, scaleFactors, th):
    Pw = np.ascontiguousarray(Pw, np.float32).reshape(-1, 3)
    M = len(Pw)
    out = np.zeros(max(M, 1), QUERY_DTYPE)
    sf = np.ascontiguousarray(scaleFactors, np.float32)
    L = lib()
    L.oo_is_in_frustum.argtypes = [C.c_int] + [C.c_void_p] * 9 + [C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.oo_is_in_frustum(M, _p(Pw), _p(np.ascontiguousarray(Pn, np.float32)), _p(np.ascontiguousarray(maxDist, np.float32)),
                       _p(np.ascontiguousarray(minDist, np.float32)), _p(np.ascontiguousarray(obs_gt0, np.uint8)),
                       _p(np.ascontiguousarray(mp_desc, np.uint8)), _p(np.ascontiguousarray(Tcw, np.float32)),
                       _p(np.asarray(K5, np.float32)), _p(np.asarray(bounds, np.float32)), float(viewingCosLimit),
                       float(logScaleFactor), _p(sf), len(sf), float(th), _p(out))
    return out[:M]


def triangulate(kf1, kf2, idx1, idx2, scaleFactors, levelSigma2, ratioFactor):
    """LocalMapping::CreateNewMapPoints per-match core for ONE keyframe pair.
    kfN = (Tcw[4x4], Twc[4x4], cam8, keysUn, keys, uRight, depth)."""
    def pack(kf):
        Tcw, Twc, cam8, ku, k, ur, d = kf
        p = np.concatenate([np.asarray(Tcw, np.float32).reshape(-1), np.asarray(Twc, np.float32).reshape(-1), np.asarray(cam8, np.float32)])
        return [np.ascontiguousarray(p, np.float32), np.ascontiguousarray(ku, KP_DTYPE), np.ascontiguousarray(k, KP_DTYPE),
                np.ascontiguousarray(ur, np.float32), np.ascontiguousarray(d, np.float32)]
    a, b = pack(kf1), pack(kf2)
    idx1 = np.ascontiguousarray(idx1, np.int32)
    idx2 = np.ascontiguousarray(idx2, np.int32)
    M = len(idx1)
    ok = np.zeros(max(M, 1), np.uint8)
    x = np.zeros((max(M, 1), 3), np.float32)
    L = lib()
    L.oo_triangulate.argtypes = [C.c_void_p] * 10 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]
    n = L.oo_triangulate(*[_p(v) for v in a], *[_p(v) for v in b], M, _p(idx1), _p(idx2), _p(np.ascontiguousarray(scaleFactors, np.float32)),
                         _p(np.ascontiguousarray(levelSigma2, np.float32)), float(ratioFactor), _p(ok), _p(x))
    assert n == int(ok[:M].sum())
    return ok[:M], x[:M]


def undistort_keypoints(keys, K4, dist):
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    out = np.zeros(max(len(keys), 1), KP_DTYPE)
    d = np.ascontiguousarray(dist if dist is not None else [], np.float32)
    L = lib()
    L.oo_undistort_keypoints.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.oo_undistort_keypoints(len(keys), _p(keys), _p(np.asarray(K4, np.float32)), _p(d) if len(d) else None, len(d), _p(out))
    return out[:len(keys)]


def image_bounds(cols, rows, K4, dist):
    d = np.ascontiguousarray(dist if dist is not None else [], np.float32)
    b = np.zeros(4, np.float32)
    L = lib()
    L.oo_image_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.oo_image_bounds(cols, rows, _p(np.asarray(K4, np.float32)), _p(d) if len(d) else None, len(d), _p(b))
    return np.array([b[0], b[1], b[2], b[3]], np.float32)


def stereo_from_rgbd(keys, keysUn, depth, mbf):
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    keysUn = np.ascontiguousarray(keysUn, KP_DTYPE)
    depth = np.ascontiguousarray(depth, np.float32)
    n = len(keys)
    ur, dp = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
    L = lib()
    L.oo_stereo_from_rgbd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.oo_stereo_from_rgbd(n, _p(keys), _p(keysUn), _p(depth), depth.shape[1], float(mbf), _p(ur), _p(dp))
    return ur[:n], dp[:n]
