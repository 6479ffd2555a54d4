"""End-to-end tracking + local-BA HARNESS on a synthetic RGB-D sequence (SURVEY.md §8(d) S1 shape).

This is measurement scaffolding, not a port of the reference's Tracking / LocalMapping control
flow (SURVEY.md §8(f)-1 "next"): it strings the hot-path operators together in the order the
reference calls them per frame (src/Tracking.cc:241-587, src/LocalMapping.cc:48-113) with a
deliberately small, deterministic data model (flat numpy arrays) and simplified policies
(keyframe every `kf_every` frames or on weak tracking; local window = last `window` keyframes).
The operator calls go through a backend object, so the same driver runs on the HIP library
(`HipBackend`) and on the CPU oracle (tests / bench supply that backend), and the two trajectories
can be compared frame by frame.

Per frame (RGB-D):  extract -> ComputeStereoFromRGBD (src/Frame.cc:883) -> [first frame: StereoInitialization]
  -> TrackWithMotionModel: SearchByProjection(Cur, Last, th=15[,30]) + PoseOptimization (src/Tracking.cc:948-1009)
  -> TrackLocalMap: SearchByProjection(F, local points, th=3) + PoseOptimization2 (src/Tracking.cc:1011-1056)
  -> keyframe policy -> new map points from depth (src/Tracking.cc:1328-1406) -> LocalBundleAdjustment (src/LocalMapping.cc:82).
"""
import time

import numpy as np

from ._lib import KP_DTYPE
from .matcher import QUERY_DTYPE


def horn_align_ate(est_xyz, gt_xyz):
    """ATE RMSE after rigid (no scale) Horn alignment by SVD — the definition of the reference's
    ExpResults/TUM/Localization/evaluate_ate.py:47-83,161 (model = estimate, data = ground truth)."""
    model = np.asarray(est_xyz, np.float64).T
    data = np.asarray(gt_xyz, np.float64).T
    mz = model - model.mean(1, keepdims=True)
    dz = data - data.mean(1, keepdims=True)
    W = np.zeros((3, 3))
    for c in range(model.shape[1]):
        W += np.outer(mz[:, c], dz[:, c])
    U, d, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    rot = U @ S @ Vh
    trans = data.mean(1, keepdims=True) - rot @ model.mean(1, keepdims=True)
    err = rot @ model + trans - data
    return float(np.sqrt((err * err).sum(0).mean()))


class HipBackend:
    """The product operators (HIP library)."""

    def __init__(self, width, height, nfeatures=1000, nlevels=8, max_kf=32, max_pts=16384, max_edges=131072, device=0):
        from . import LocalBundleAdjuster, ORBextractor, ORBmatcher, PoseOptimizer
        self.ex = ORBextractor(nfeatures, 1.2, nlevels, 20, 7, width, height, device=device)
        cap = max(self.ex.cap, 4096)
        self.m_last = ORBmatcher(0.9, True, max_keypoints=self.ex.cap, max_queries=self.ex.cap, device=device)
        self.m_map = ORBmatcher(0.8, True, max_keypoints=self.ex.cap, max_queries=cap * 2, device=device)
        self.po = PoseOptimizer(max_points=self.ex.cap, device=device)
        self.ba = LocalBundleAdjuster(max_keyframes=max_kf, max_points=max_pts, max_edges=max_edges, device=device)
        self.scale = self.ex.GetScaleFactors()
        self.inv_sigma2 = self.ex.GetInverseScaleSigmaSquares()

    def extract(self, img):
        return self.ex(img)

    def search_last(self, kc, uR, dc, bounds, Xw, has, kl, dl, Tcw, Tlw, cam, th):
        nm, qm, qd, km = self.m_last.search_last_frame(kc, uR, dc, None, bounds, Xw, has, kl, dl, Tcw, Tlw, cam, self.scale, th, False)
        return nm, km

    def search_map(self, kc, uR, dc, blocked, bounds, queries):
        nm, qm, qd, km = self.m_map.search_window(kc, uR, dc, blocked, bounds, queries, True, False)
        return nm, km

    def pose_opt(self, Tcw, Xw, obs, inv, has, K5):
        n, T, outl, _ = self.po.PoseOptimization(Tcw, Xw, obs, inv, has, K5)
        return n, T, outl

    def lba(self, poses, fixed, points, ekf, ept, eobs, einv, K5):
        po, xo, er, _ = self.ba.LocalBundleAdjustment(poses, fixed, points, ekf, ept, eobs, einv, K5)
        return po, xo, er


class Tracker:
    def __init__(self, backend, cam, width, height, depth, kf_every=8, window=8, max_fixed=8):
        self.b = backend
        self.fx, self.fy, self.cx, self.cy, self.bf = cam
        self.cam6 = (self.fx, self.fy, self.cx, self.cy, self.bf, self.bf / self.fx)
        self.K5 = np.array([self.fx, self.fy, self.cx, self.cy, self.bf], np.float32)
        self.bounds = (0.0, 0.0, float(width), float(height))
        self.depth = float(depth)
        self.kf_every, self.window, self.max_fixed = kf_every, window, max_fixed
        # map
        self.mp_X = np.zeros((0, 3), np.float32)
        self.mp_desc = np.zeros((0, 32), np.uint8)
        self.mp_obs = np.zeros(0, np.int32)            # number of keyframe observations
        self.mp_level = np.zeros(0, np.int32)          # octave of the creating keypoint (= predicted level at the creation distance)
        self.kfs = []                                  # dicts: pose, kps, uR, mp (index per keypoint)
        self.last = None
        self.velocity = None
        self.traj = []
        self.n_frames = 0
        self.stats = {"lba_calls": 0, "matches_last": [], "matches_map": [], "inliers": []}

    # --- helpers -------------------------------------------------------------------------
    def _unproject(self, kps, Tcw):
        z = self.depth
        Xc = np.stack([(kps["x"] - self.cx) * z / self.fx, (kps["y"] - self.cy) * z / self.fy, np.full(len(kps), z, np.float32)], 1)
        R, t = Tcw[:3, :3].astype(np.float64), Tcw[:3, 3].astype(np.float64)
        return ((Xc.astype(np.float64) - t) @ R).astype(np.float32)

    def _new_points(self, frame, mask):
        idx = np.where(mask)[0]
        if len(idx) == 0:
            return
        X = self._unproject(frame["kps"][idx], frame["pose"])
        base = len(self.mp_X)
        self.mp_X = np.concatenate([self.mp_X, X])
        self.mp_desc = np.concatenate([self.mp_desc, frame["desc"][idx]])
        self.mp_obs = np.concatenate([self.mp_obs, np.zeros(len(idx), np.int32)])
        self.mp_level = np.concatenate([self.mp_level, frame["kps"]["octave"][idx].astype(np.int32)])
        frame["mp"][idx] = base + np.arange(len(idx))

    def _pose_opt(self, frame):
        has = (frame["mp"] >= 0).astype(np.uint8)
        Xw = np.zeros((len(has), 3), np.float32)
        Xw[has > 0] = self.mp_X[frame["mp"][has > 0]]
        obs = np.stack([frame["kps"]["x"], frame["kps"]["y"], frame["uR"]], 1).astype(np.float32)
        inv = self.b.inv_sigma2[frame["kps"]["octave"]].astype(np.float32)
        n, T, outl = self.b.pose_opt(frame["pose"], Xw, obs, inv, has, self.K5)
        if has.sum() >= 3:
            frame["pose"] = T.astype(np.float32)
        drop = (outl > 0) & (has > 0)
        frame["mp"][drop] = -1          # src/Tracking.cc:985-1000 discards outliers
        return int(n)

    # --- per frame -----------------------------------------------------------------------
    def track(self, img):
        kps, desc = self.b.extract(img)
        uR = (kps["x"] - self.bf / self.depth).astype(np.float32)      # ComputeStereoFromRGBD, src/Frame.cc:883-904
        frame = dict(kps=kps, desc=desc, uR=uR, mp=np.full(len(kps), -1, np.int64), pose=np.eye(4, dtype=np.float32))
        if self.last is None:
            self._new_points(frame, np.ones(len(kps), bool))         # StereoInitialization, src/Tracking.cc:590-653
            self._insert_keyframe(frame)
        else:
            last = self.last
            frame["pose"] = (self.velocity @ last["pose"]).astype(np.float32) if self.velocity is not None else last["pose"].copy()
            # TrackWithMotionModel, src/Tracking.cc:948-1009
            has = np.where(last["mp"] >= 0, 3, 0).astype(np.uint8)
            Xw = np.zeros((len(has), 3), np.float32)
            Xw[has > 0] = self.mp_X[last["mp"][has > 0]]
            ldesc = last["desc"].copy()
            ldesc[has > 0] = self.mp_desc[last["mp"][has > 0]]
            nm, km = self.b.search_last(kps, uR, desc, self.bounds, Xw, has, last["kps"], ldesc, frame["pose"], last["pose"], self.cam6, 15.0)
            if nm < 20:
                nm, km = self.b.search_last(kps, uR, desc, self.bounds, Xw, has, last["kps"], ldesc, frame["pose"], last["pose"], self.cam6, 30.0)
            sel = km >= 0
            frame["mp"][sel] = last["mp"][km[sel]]
            self.stats["matches_last"].append(int(nm))
            self._pose_opt(frame)
            # TrackLocalMap, src/Tracking.cc:1011-1056 (local map = points of the window keyframes)
            nloc = self._search_local_points(frame)
            self.stats["matches_map"].append(int(nloc))
            ninl = self._pose_opt(frame)
            self.stats["inliers"].append(ninl)
            self.velocity = (frame["pose"].astype(np.float64) @ np.linalg.inv(last["pose"].astype(np.float64))).astype(np.float32)
            if self.n_frames % self.kf_every == 0 or ninl < 100:
                self._new_points(frame, frame["mp"] < 0)            # CreateNewKeyFrame, src/Tracking.cc:1328-1406
                self._insert_keyframe(frame)
        self.last = frame
        self.n_frames += 1
        Twc = np.linalg.inv(frame["pose"].astype(np.float64))
        self.traj.append(Twc[:3, 3].copy())
        return frame["pose"]

    def _search_local_points(self, frame):
        ids = np.unique(np.concatenate([kf["mp"][kf["mp"] >= 0] for kf in self.kfs[-self.window:]]))
        ids = np.setdiff1d(ids, frame["mp"][frame["mp"] >= 0])
        if len(ids) == 0:
            return 0
        # Frame::isInFrustum (src/Frame.cc:509-565) for a fronto-parallel scene: positive depth, inside the image
        T = frame["pose"]
        Xc = self.mp_X[ids] @ T[:3, :3].T + T[:3, 3]
        z = Xc[:, 2]
        ok = z > 0
        invz = np.where(ok, 1.0 / np.where(ok, z, 1.0), 0).astype(np.float32)
        u = (self.fx * Xc[:, 0] * invz + self.cx).astype(np.float32)
        v = (self.fy * Xc[:, 1] * invz + self.cy).astype(np.float32)
        ok &= (u >= 0) & (u <= self.bounds[2]) & (v >= 0) & (v <= self.bounds[3])
        ids, u, v, invz = ids[ok], u[ok], v[ok], invz[ok]
        q = np.zeros(len(ids), QUERY_DTYPE)
        q["u"], q["v"], q["ur"] = u, v, u - self.bf * invz
        lvl = self.mp_level[ids]                                    # PredictScale at the creation distance
        q["radius"] = (np.float32(2.5 * 3.0) * self.b.scale[lvl]).astype(np.float32)   # RadiusByViewingCos(1) * th(RGB-D=3) * scale[level]
        q["minLevel"], q["maxLevel"] = lvl - 1, lvl
        q["flags"] = 1 | ((self.mp_obs[ids] > 0).astype(np.int32) << 1)
        q["desc"] = self.mp_desc[ids]
        blocked = (frame["mp"] >= 0).astype(np.uint8)
        blocked[frame["mp"] >= 0] &= (self.mp_obs[frame["mp"][frame["mp"] >= 0]] > 0).astype(np.uint8)
        nm, km = self.b.search_map(frame["kps"], frame["uR"], frame["desc"], blocked, self.bounds, q)
        sel = km >= 0
        frame["mp"][sel] = ids[km[sel]]
        return nm

    # --- local mapping -------------------------------------------------------------------
    def _insert_keyframe(self, frame):
        kf = dict(pose=frame["pose"].copy(), kps=frame["kps"], uR=frame["uR"], mp=frame["mp"].copy())
        obs = kf["mp"][kf["mp"] >= 0]
        np.add.at(self.mp_obs, obs, 1)
        self.kfs.append(kf)
        if len(self.kfs) > 2:                                       # src/LocalMapping.cc:81
            self._local_ba()

    def _local_ba(self):
        local = list(range(max(0, len(self.kfs) - self.window), len(self.kfs)))
        pts = np.unique(np.concatenate([self.kfs[k]["mp"][self.kfs[k]["mp"] >= 0] for k in local]))
        older = [k for k in range(0, local[0]) if np.intersect1d(self.kfs[k]["mp"], pts).size > 0][-self.max_fixed:]
        kf_ids = older + local
        fixed = np.array([1] * len(older) + [0] * len(local), np.uint8)
        if kf_ids[0] == 0 and not older:
            fixed[0] = 2                                            # keyframe 0 is fixed (src/Optimizer.cc:529)
        pmap = -np.ones(len(self.mp_X), np.int64)
        pmap[pts] = np.arange(len(pts))
        ekf, ept, eobs, einv, eref = [], [], [], [], []
        for j, k in enumerate(kf_ids):
            kf = self.kfs[k]
            idx = np.where(kf["mp"] >= 0)[0]
            idx = idx[pmap[kf["mp"][idx]] >= 0]
            ekf.append(np.full(len(idx), j, np.int32))
            ept.append(pmap[kf["mp"][idx]].astype(np.int32))
            eobs.append(np.stack([kf["kps"]["x"][idx], kf["kps"]["y"][idx], kf["uR"][idx]], 1))
            einv.append(self.b.inv_sigma2[kf["kps"]["octave"][idx]])
            eref.append(np.stack([np.full(len(idx), k), idx], 1))
        ekf, ept = np.concatenate(ekf), np.concatenate(ept)
        eobs, einv, eref = np.concatenate(eobs).astype(np.float32), np.concatenate(einv).astype(np.float32), np.concatenate(eref)
        poses = np.stack([self.kfs[k]["pose"] for k in kf_ids]).astype(np.float32)
        po, xo, erase = self.b.lba(poses, fixed, self.mp_X[pts], ekf, ept, eobs, einv, self.K5)
        self.stats["lba_calls"] += 1
        for j, k in enumerate(kf_ids):
            if fixed[j] != 1:
                self.kfs[k]["pose"] = po[j].astype(np.float32)
        self.mp_X[pts] = xo
        for (k, i) in eref[erase > 0]:                              # src/Optimizer.cc:748-757
            p = self.kfs[k]["mp"][i]
            if p >= 0:
                self.mp_obs[p] -= 1
                self.kfs[k]["mp"][i] = -1
        # the tracking frame that became this keyframe follows its optimised pose
        if self.last is not None and kf_ids[-1] == len(self.kfs) - 1:
            pass


def run_sequence(backend, frames, offsets, cam, depth, **kw):
    """Runs the harness over a synth.make_stream sequence. Returns (tracker, seconds, ate_rmse_m)."""
    h, w = frames.shape[1:]
    tr = Tracker(backend, cam, w, h, depth, **kw)
    fx, fy = cam[0], cam[1]
    t0 = time.perf_counter()
    for img in frames:
        tr.track(img)
    dt = time.perf_counter() - t0
    # ground truth camera centres: the crop offset moves the camera parallel to the plane
    off = (offsets - offsets[0]).astype(np.float64)
    gt = np.stack([off[:, 0] * depth / fx, off[:, 1] * depth / fy, np.zeros(len(off))], 1)
    ate = horn_align_ate(np.array(tr.traj), gt)
    return tr, dt, ate
