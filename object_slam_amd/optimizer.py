"""Optimizer — Python mirror of ORB_SLAM2::Optimizer's hot entry points (reference
include/Optimizer.h:45-46) over the C ABI.  All arithmetic runs in the HIP library."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr


class Semantic(C.Structure):
    _fields_ = [("nObj", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("masks", C.c_void_p), ("nObjMp", C.c_int32),
                ("objmp_Xw", C.c_void_p), ("objmp_obj", C.c_void_p), ("nJoint", C.c_int32), ("joint_kp", C.c_void_p),
                ("joint_obj", C.c_void_p), ("kp_uv", C.c_void_p), ("bounds", C.c_float * 4), ("invSigma2_0", C.c_float)]


class PoseOptimizer:
    """Optimizer::PoseOptimization (reference src/Optimizer.cc:239-451)."""

    def __init__(self, max_points=4096, max_batch=1, device=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        check(self.L.oslam_poseopt_create(C.byref(self.h), max_batch, max_points, device))
        self.max_points, self.max_batch = max_points, max_batch

    def close(self):
        if getattr(self, "h", None):
            self.L.oslam_poseopt_destroy(self.h)
            self.h = None

    __del__ = close

    def PoseOptimization(self, Tcw, Xw, obs, invSigma2, has_mp, K5):
        """Returns (n_inliers, Tcw_out[4,4] float32, outlier uint8[N], (LM iterations, trials))."""
        Xw = np.ascontiguousarray(Xw, np.float32)
        N = len(Xw)
        obs = np.ascontiguousarray(obs, np.float32)
        inv = np.ascontiguousarray(invSigma2, np.float32)
        has = np.ascontiguousarray(has_mp, np.uint8)
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        K = np.ascontiguousarray(K5, np.float32)
        out = np.zeros(16, np.float32)
        outl = np.zeros(max(N, 1), np.uint8)
        n = C.c_int(0)
        stats = np.zeros(2, np.int32)
        check(self.L.oslam_pose_optimize(self.h, N, ptr(T), ptr(Xw), ptr(obs), ptr(inv), ptr(has), ptr(K), ptr(out),
                                         ptr(outl), C.byref(n), ptr(stats)))
        return n.value, out.reshape(4, 4), outl[:N], (int(stats[0]), int(stats[1]))

    def lm_trace(self, fn, cap=512):
        """Test hook: runs fn() with the kernel's LM trace on; returns (fn's result, trace[n, 6]) with rows
        (F before the trial, F of the trial, rho, lambda of the trial, accepted, first trial of a round)."""
        check(self.L.oslam_poseopt_trace(self.h, cap))
        try:
            r = fn()
            buf = np.zeros((cap, 6), np.float64)
            n = C.c_int32(0)
            check(self.L.oslam_poseopt_trace_read(self.h, ptr(buf), C.byref(n)))
        finally:
            check(self.L.oslam_poseopt_trace(self.h, 0))
        return r, buf[:min(n.value, cap)].copy()

    def PoseOptimization2(self, p):
        """ObjectOptimizer::PoseOptimization2 (reference src/ObjectOptimizer.cc:624) on a dict with the keys of
        synth.make_semantic_problem.  Returns (n_inliers, Tcw_out, outlier, nSemNum)."""
        Xw = np.ascontiguousarray(p["Xw"], np.float32)
        N = len(Xw)
        keep = [Xw, np.ascontiguousarray(p["obs"], np.float32), np.ascontiguousarray(p["invSigma2"], np.float32),
                np.ascontiguousarray(p["has_mp"], np.uint8), np.ascontiguousarray(p["Tcw"], np.float32).reshape(16),
                np.ascontiguousarray(p["K"], np.float32), np.ascontiguousarray(p["masks"], np.uint8),
                np.ascontiguousarray(p["objmp_Xw"], np.float32), np.ascontiguousarray(p["objmp_obj"], np.int32),
                np.ascontiguousarray(p["joint_kp"], np.int32), np.ascontiguousarray(p["joint_obj"], np.int32),
                np.ascontiguousarray(p["kp_uv"], np.float32)]
        sem = Semantic()
        sem.nObj, sem.H, sem.W = keep[6].shape
        sem.masks = keep[6].ctypes.data
        sem.nObjMp, sem.objmp_Xw, sem.objmp_obj = len(keep[8]), keep[7].ctypes.data, keep[8].ctypes.data
        sem.nJoint, sem.joint_kp, sem.joint_obj = len(keep[9]), keep[9].ctypes.data, keep[10].ctypes.data
        sem.kp_uv = keep[11].ctypes.data
        for i in range(4):
            sem.bounds[i] = float(p["bounds"][i])
        sem.invSigma2_0 = float(p["invSigma2_0"])
        out = np.zeros(16, np.float32)
        outl = np.zeros(max(N, 1), np.uint8)
        n, ns = C.c_int(0), C.c_int(0)
        check(self.L.oslam_pose_optimize2(self.h, N, ptr(keep[4]), ptr(Xw), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(keep[5]),
                                          C.byref(sem), ptr(out), ptr(outl), C.byref(n), C.byref(ns)))
        return n.value, out.reshape(4, 4), outl[:N], ns.value

    def optimize_batch_device(self, batch, stride, d_n, n_const, d_Tcw, d_Xw, d_obs, d_inv, d_has, K5, stream=None):
        K = np.ascontiguousarray(K5, np.float32)
        check(self.L.oslam_pose_optimize_batch_device(self.h, batch, stride, C.c_void_p(d_n or 0), n_const, C.c_void_p(d_Tcw),
                                                      C.c_void_p(d_Xw), C.c_void_p(d_obs), C.c_void_p(d_inv), C.c_void_p(d_has),
                                                      ptr(K), C.c_void_p(stream or 0)))

    def results_device(self):
        a, b, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self.L.oslam_poseopt_results_device(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value


class LbaProblem(C.Structure):
    _fields_ = [("nKF", C.c_int32), ("poses", C.c_void_p), ("fixed", C.c_void_p), ("nP", C.c_int32), ("points", C.c_void_p),
                ("nE", C.c_int32), ("edge_kf", C.c_void_p), ("edge_pt", C.c_void_p), ("edge_obs", C.c_void_p), ("edge_invSigma2", C.c_void_p),
                ("poses_out", C.c_void_p), ("points_out", C.c_void_p), ("erase", C.c_void_p), ("stats", C.c_void_p)]


class LocalBundleAdjuster:
    """Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:453-778) on a flattened graph."""

    def __init__(self, max_keyframes=64, max_points=8192, max_edges=65536, max_batch=1, device=0):
        self.L = _lib.lib()
        self.L.oslam_lba_stop_flag.restype = C.POINTER(C.c_int32)
        self.h = C.c_void_p()
        check(self.L.oslam_lba_create(C.byref(self.h), max_batch, max_keyframes, max_points, max_edges, device))

    def close(self):
        if getattr(self, "h", None):
            self.L.oslam_lba_destroy(self.h)
            self.h = None

    __del__ = close

    def set_mode(self, wide):
        """wide=True: one problem over the whole GPU (multi-kernel LM); False: one workgroup per problem."""
        check(self.L.oslam_lba_set_mode(self.h, int(wide)))

    def set_schur(self, mode):
        """Schur complement of the wide mode: 0 pair gather (lists built on the device), 1 LDS tiles, 2 chosen per call, 3 pair gather with host-built lists (include/oslam_hip.h)."""
        check(self.L.oslam_lba_set_schur(self.h, C.c_int(mode)))

    def set_solver(self, mode):
        """Reduced-camera-system solver of the wide mode: 0 auto (LDS-resident matrix-core kernel at 133..186 unknowns), 1 global-memory matrix cores for every size, 2 never, 3 the round-3 choice (include/oslam_hip.h)."""
        check(self.L.oslam_lba_set_solver(self.h, C.c_int(mode)))

    def lm_trace(self, fn, cap=512):
        """Test hook: runs fn() with the LM trace of window 0 (wide layout) on; returns (fn's result, trace[n, 6]) with rows
        (F before the trial, F of the trial, rho, lambda of the trial, accepted, first trial of a stage)."""
        check(self.L.oslam_lba_trace(self.h, cap))
        try:
            r = fn()
            buf = np.zeros((cap, 6), np.float64)
            n = C.c_int32(0)
            check(self.L.oslam_lba_trace_read(self.h, ptr(buf), C.byref(n)))
        finally:
            check(self.L.oslam_lba_trace(self.h, 0))
        return r, buf[:min(n.value, cap)].copy()

    def stop_flag(self):
        """The pbStopFlag: a pinned int visible to the running kernel (set [0] = 1 to abort)."""
        return self.L.oslam_lba_stop_flag(self.h)

    def LocalBundleAdjustment(self, poses, fixed, points, edge_kf, edge_pt, edge_obs, edge_invSigma2, K5, use_stop_flag=False):
        """Returns (poses_out [K,4,4] f32, points_out [P,3] f32, erase u8[E], stats(it1, trials1, it2, trials2))."""
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        fixed = np.ascontiguousarray(fixed, np.uint8)
        points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        ekf = np.ascontiguousarray(edge_kf, np.int32)
        ept = np.ascontiguousarray(edge_pt, np.int32)
        eobs = np.ascontiguousarray(edge_obs, np.float32).reshape(-1, 3)
        einv = np.ascontiguousarray(edge_invSigma2, np.float32)
        K = np.ascontiguousarray(K5, np.float32)
        pout = np.zeros_like(poses)
        xout = np.zeros_like(points)
        erase = np.zeros(max(len(ekf), 1), np.uint8)
        stats = np.zeros(4, np.int32)
        check(self.L.oslam_lba_optimize(self.h, len(poses), ptr(poses), ptr(fixed), len(points), ptr(points), len(ekf), ptr(ekf),
                                        ptr(ept), ptr(eobs), ptr(einv), ptr(K), int(use_stop_flag), ptr(pout), ptr(xout),
                                        ptr(erase), ptr(stats)))
        return pout.reshape(-1, 4, 4), xout, erase[:len(ekf)], tuple(int(v) for v in stats)

    def BundleAdjustment(self, poses, fixed, points, edge_kf, edge_pt, edge_obs, edge_invSigma2, K5, nIterations=5, bRobust=True):
        """Optimizer::BundleAdjustment (reference src/Optimizer.cc:49-237). Returns (poses_out, points_out)."""
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        fixed = np.ascontiguousarray(fixed, np.uint8)
        points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        ekf = np.ascontiguousarray(edge_kf, np.int32)
        ept = np.ascontiguousarray(edge_pt, np.int32)
        eobs = np.ascontiguousarray(edge_obs, np.float32).reshape(-1, 3)
        einv = np.ascontiguousarray(edge_invSigma2, np.float32)
        K = np.ascontiguousarray(K5, np.float32)
        pout, xout = np.zeros_like(poses), np.zeros_like(points)
        check(self.L.oslam_ba_optimize(self.h, len(poses), ptr(poses), ptr(fixed), len(points), ptr(points), len(ekf), ptr(ekf), ptr(ept),
                                       ptr(eobs), ptr(einv), ptr(K), int(nIterations), int(bRobust), 0, ptr(pout), ptr(xout)))
        return pout.reshape(-1, 4, 4), xout

    def LocalBundleAdjustmentBatch(self, problems, K5):
        """Independent keyframe windows in ONE launch (one workgroup per problem). `problems`: list of dicts with the
        keys of synth.make_lba_problem. Returns a list of (poses_out, points_out, erase, stats)."""
        n = len(problems)
        arr = (LbaProblem * n)()
        keep, outs = [], []
        for i, q in enumerate(problems):
            poses = np.ascontiguousarray(q["poses"], np.float32).reshape(-1, 16)
            fixed = np.ascontiguousarray(q["fixed"], np.uint8)
            points = np.ascontiguousarray(q["points"], np.float32).reshape(-1, 3)
            ekf = np.ascontiguousarray(q["edge_kf"], np.int32)
            ept = np.ascontiguousarray(q["edge_pt"], np.int32)
            eobs = np.ascontiguousarray(q["edge_obs"], np.float32).reshape(-1, 3)
            einv = np.ascontiguousarray(q["edge_invSigma2"], np.float32)
            pout, xout = np.zeros_like(poses), np.zeros_like(points)
            erase, stats = np.zeros(max(len(ekf), 1), np.uint8), np.zeros(4, np.int32)
            keep.append((poses, fixed, points, ekf, ept, eobs, einv))
            outs.append((pout, xout, erase, stats))
            a = arr[i]
            a.nKF, a.poses, a.fixed = len(poses), poses.ctypes.data, fixed.ctypes.data
            a.nP, a.points = len(points), points.ctypes.data
            a.nE, a.edge_kf, a.edge_pt, a.edge_obs, a.edge_invSigma2 = len(ekf), ekf.ctypes.data, ept.ctypes.data, eobs.ctypes.data, einv.ctypes.data
            a.poses_out, a.points_out, a.erase, a.stats = pout.ctypes.data, xout.ctypes.data, erase.ctypes.data, stats.ctypes.data
        K = np.ascontiguousarray(K5, np.float32)
        check(self.L.oslam_lba_optimize_batch(self.h, n, arr, ptr(K)))
        return [(po.reshape(-1, 4, 4), xo, er[:len(k[3])], tuple(int(v) for v in st)) for (po, xo, er, st), k in zip(outs, keep)]
