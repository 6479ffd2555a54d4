"""ctypes view of the batch-of-sequences tracking + local-mapping driver (include/oslam_slam.h).

`System(cfg)` mirrors the reference's ORB_SLAM2::System for S RGB-D sequences advanced in lockstep on one GPU
(System::TrackRGBD -> `TrackRGBD`, SaveTrajectoryTUM -> `trajectory`, SaveKeyFrameTrajectoryTUM -> `keyframe_trajectory`).
The product constructor always binds the HIP operators; `ops=` takes the address of another oslam_slam_ops_t (tests pass
the CPU oracle's table) and nothing in this module imports oracle/.
"""
import ctypes as C

import numpy as np

from ._lib import check, lib, ptr


class SlamConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("dist", C.c_float * 5), ("ndist", C.c_int32), ("bf", C.c_float), ("thDepth", C.c_float), ("fps", C.c_float),
                ("nFeatures", C.c_int32), ("scaleFactor", C.c_float), ("nLevels", C.c_int32), ("iniThFAST", C.c_int32), ("minThFAST", C.c_int32),
                ("n_sequences", C.c_int32), ("device", C.c_int32), ("host_threads", C.c_int32), ("local_mapping", C.c_int32), ("sensor", C.c_int32)]


class SlamOps(C.Structure):
    _fields_ = [("ctx", C.c_void_p)] + [(n, C.c_void_p) for n in (
        "max_keypoints", "scale_tables", "image_bounds", "frames_rgbd", "search_last", "search_local", "pose_opt", "mp_update", "lba", "fuse", "bow",
        "triangulate", "destroy", "frames_stereo", "object_kps", "pose_opt2", "register_keyframes", "bow_keyed", "fuse_keyed", "mp_update_keyed", "kernel_times", "bow_nodes_keyed", "resident_points", "fuse_points_keyed", "point_record", "frames_rgbd_raw16", "release_keyframes", "lba_submit", "lba_wait", "mp_update_windows",
        # round 5 (include/oslam_slam.h): the mirror of the observation graph, arrays on demand, deferred descriptor updates
        "map_journal", "kf_culling_counts", "kf_culling_collect", "fuse_into_current", "local_points_list", "keyframe_raw_keys", "keyframe_descriptors", "frame_descriptors",
        "mp_update_keyed_async", "mp_update_collect")]


def _check_struct_sizes(L):
    """The structs above are hand-written mirrors of include/oslam_slam.h: compare their sizes with the library's own before the first call (a short SlamOps was a
    silent 80-byte overflow of every operator table handed to oslam_slam_create_with_ops until the end of round 5)."""
    if getattr(L, "_oslam_slam_sizes_ok", False):
        return
    out = (C.c_int32 * 4)()
    check(L.oslam_slam_struct_sizes(out))
    mine = (C.sizeof(SlamConfig), C.sizeof(SlamOps), C.sizeof(SlamObjects))
    if tuple(out[:3]) != mine:
        raise RuntimeError("object_slam_amd/slam.py mirrors include/oslam_slam.h with other struct sizes than the library was built with: config / ops / objects %s here, %s there"
                           % (mine, tuple(out[:3])))
    L._oslam_slam_sizes_ok = True


class SlamObjects(C.Structure):
    """oslam_slam_objects_t: the semantic detections of one frame."""
    _fields_ = [("n", C.c_int32), ("masks", C.POINTER(C.c_void_p)), ("track_id", C.POINTER(C.c_int32)), ("label", C.POINTER(C.c_int32))]


# reference Examples/RGB-D/TUM2.yaml (distortion left at zero: the synthetic streams are rendered without it)
TUM2 = dict(fx=520.908620, fy=521.007327, cx=325.141442, cy=249.701764, bf=40.0, thDepth=40.0, fps=30.0)

# reference Examples/Stereo/KITTI00-02.yaml
KITTI00 = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448, thDepth=35.0, fps=10.0)

OK, LOST, NOT_INITIALIZED = 2, 3, 1
LM_SYNC, LM_DEFERRED = 0x1F, 0x3F   # oslam_slam_config_t::local_mapping: the whole LocalMapping::Run pass; bit 5 = the deferred schedule (include/oslam_slam.h)
STEREO, RGBD = 1, 2


def make_config(width, height, n_sequences, cam=TUM2, dist=None, nFeatures=1000, scaleFactor=1.2, nLevels=8, iniThFAST=20, minThFAST=7, device=0,
                host_threads=0, local_mapping=0x1F, sensor=RGBD):
    c = SlamConfig()
    c.width, c.height = width, height
    c.fx, c.fy, c.cx, c.cy, c.bf, c.thDepth, c.fps = cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"], cam["thDepth"], cam["fps"]
    if dist is not None:
        for i, v in enumerate(dist):
            c.dist[i] = v
        c.ndist = len(dist)
    c.nFeatures, c.scaleFactor, c.nLevels, c.iniThFAST, c.minThFAST = nFeatures, scaleFactor, nLevels, iniThFAST, minThFAST
    c.n_sequences, c.device, c.host_threads, c.local_mapping, c.sensor = n_sequences, device, host_threads, local_mapping, sensor
    return c


class System:
    def __init__(self, cfg, ops=None):
        self.L = lib()
        _check_struct_sizes(self.L)
        self.cfg = cfg
        self.S = cfg.n_sequences
        self.h = C.c_void_p()
        if ops is None:
            check(self.L.oslam_slam_create(C.byref(self.h), C.byref(cfg)))
        else:
            check(self.L.oslam_slam_create_with_ops(C.byref(self.h), C.byref(cfg), C.byref(ops)))
        self._gp = (C.c_void_p * self.S)()
        self._dp = (C.c_void_p * self.S)()
        self.Tcw = np.zeros((self.S, 4, 4), np.float32)
        self.state = np.zeros(self.S, np.int32)

    def close(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            self.L.oslam_slam_destroy(h)
            h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: module globals may already be gone
            pass

    def prepare_rgbd(self, gray, depth, timestamps=None, objects=None, on_device=False, gray_stride=None, depth_pitch=None, mask_stride=None):
        """Marshals one TrackRGBD call ahead of time (pointer tables as ctypes arrays): callers that replay recorded streams keep the Python work out of
        their frame loop.  Returns an opaque tuple for `track_prepared`; the arrays behind the pointers must stay alive."""
        gp, dp = (C.c_void_p * self.S)(), (C.c_void_p * self.S)()
        for i in range(self.S):
            gp[i] = gray[i] if on_device else gray[i].__array_interface__["data"][0]
            dp[i] = depth[i] if on_device else depth[i].__array_interface__["data"][0]
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, np.float64)
        gs = gray_stride if on_device else gray[0].strides[0]
        dpp = depth_pitch if on_device else depth[0].strides[0] // 4
        arr = keep = None
        ms = 0
        if objects is not None:
            arr, ms, keep = self._objects(objects, on_device)
            ms = mask_stride if on_device else (ms or self.cfg.width)
        return (gp, dp, ts, C.c_int(gs), C.c_int(dpp), C.c_int(1 if on_device else 0), arr, C.c_int(ms), (keep, gray, depth, objects), None)

    def prepare_rgbd_bulk(self, gray, depth, stamps, gray_stride, depth_pitch, masks=None, track_ids=None, labels=None, mask_stride=0, depth_u16_factor=None, mask_bits=False,
                          on_device=1):
        """prepare_rgbd for T steps at once from address tables: gray / depth uint64 [T, S] (device-accessible addresses), stamps float64 [T, S], masks uint64
        [T, S, n] with track_ids / labels int32 [S, n].  Returns a list of T tuples for `track_prepared`."""
        T, S = gray.shape
        assert S == self.S
        gray, depth, stamps = np.ascontiguousarray(gray, np.uint64), np.ascontiguousarray(depth, np.uint64), np.ascontiguousarray(stamps, np.float64)
        objs = None
        if masks is not None:
            n = masks.shape[2]
            masks = np.ascontiguousarray(masks, np.uint64)
            track_ids, labels = np.ascontiguousarray(track_ids, np.int32), np.ascontiguousarray(labels, np.int32)
            odt = np.dtype([("n", np.int32), ("pad", np.int32), ("masks", np.uint64), ("track_id", np.uint64), ("label", np.uint64)])
            assert odt.itemsize == C.sizeof(SlamObjects)
            objs = np.zeros((T, S), odt)
            objs["n"] = n
            objs["masks"] = masks.ctypes.data + (np.arange(T * S, dtype=np.uint64).reshape(T, S)) * np.uint64(8 * n)
            objs["track_id"] = (track_ids.ctypes.data + np.arange(S, dtype=np.uint64) * np.uint64(4 * n))[None, :]
            objs["label"] = (labels.ctypes.data + np.arange(S, dtype=np.uint64) * np.uint64(4 * n))[None, :]
        keep = (gray, depth, stamps, masks, track_ids, labels, objs)
        out = []
        for t in range(T):
            gp = C.cast(gray[t].ctypes.data, C.POINTER(C.c_void_p)); dp = C.cast(depth[t].ctypes.data, C.POINTER(C.c_void_p))
            arr = C.cast(objs[t].ctypes.data, C.POINTER(SlamObjects)) if objs is not None else None
            ms = (0 if mask_bits else mask_stride) if objs is not None else 0
            dpp = C.c_int(depth_pitch)
            out.append((gp, dp, stamps[t], C.c_int(gray_stride), dpp, C.c_int(on_device), arr, C.c_int(ms), keep, depth_u16_factor))
        return out

    def prepare_stereo_bulk(self, left, right, stamps, stride, on_device=1):
        T, S = left.shape
        assert S == self.S
        left, right, stamps = np.ascontiguousarray(left, np.uint64), np.ascontiguousarray(right, np.uint64), np.ascontiguousarray(stamps, np.float64)
        keep = (left, right, stamps)
        return [(C.cast(left[t].ctypes.data, C.POINTER(C.c_void_p)), C.cast(right[t].ctypes.data, C.POINTER(C.c_void_p)), stamps[t], C.c_int(stride), C.c_int(on_device), keep)
                for t in range(T)]

    def track_prepared(self, prep):
        gp, dp, ts, gs, dpp, dev, arr, ms, _, u16 = prep
        if u16 is not None:   # raw 16-bit depth images + DepthMapFactor (the reference's own input, src/Tracking.cc:262)
            check(self.L.oslam_slam_track_rgbd_raw16(self.h, gp, gs, dp, dpp, C.c_float(u16), dev, ptr(ts) if ts is not None else None, arr, ms, ptr(self.Tcw), ptr(self.state)))
        elif arr is not None:
            check(self.L.oslam_slam_track_rgbd_objects(self.h, gp, gs, dp, dpp, dev, ptr(ts) if ts is not None else None, arr, ms, ptr(self.Tcw), ptr(self.state)))
        else:
            check(self.L.oslam_slam_track_rgbd(self.h, gp, gs, dp, dpp, dev, ptr(ts) if ts is not None else None, ptr(self.Tcw), ptr(self.state)))
        return self.Tcw, self.state

    def prepare_stereo(self, left, right, timestamps=None, on_device=False, stride=None):
        gp, rp = (C.c_void_p * self.S)(), (C.c_void_p * self.S)()
        for i in range(self.S):
            gp[i] = left[i] if on_device else left[i].__array_interface__["data"][0]
            rp[i] = right[i] if on_device else right[i].__array_interface__["data"][0]
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, np.float64)
        return (gp, rp, ts, C.c_int(stride if on_device else left[0].strides[0]), C.c_int(1 if on_device else 0), (left, right))

    def track_stereo_prepared(self, prep):
        gp, rp, ts, st, dev, _ = prep
        check(self.L.oslam_slam_track_stereo(self.h, gp, rp, st, dev, ptr(ts) if ts is not None else None, ptr(self.Tcw), ptr(self.state)))
        return self.Tcw, self.state

    def _objects(self, objects, on_device):
        """objects: per sequence None or dict(masks = list of uint8 [H,W] arrays (device addresses with on_device), track_ids, labels (optional)).
        Returns (array of SlamObjects, mask stride in bytes, keep-alive list)."""
        arr = (SlamObjects * self.S)()
        keep, stride = [], 0
        for i in range(self.S):
            ob = objects[i] if objects is not None else None
            n = 0 if ob is None else len(ob["masks"])
            arr[i].n = n
            if n == 0:
                continue
            mp = (C.c_void_p * n)()
            for m in range(n):
                if on_device:
                    mp[m] = ob["masks"][m]
                else:
                    a = ob["masks"][m]
                    assert a.dtype == np.uint8 and a.strides[1] == 1
                    mp[m] = a.__array_interface__["data"][0]
                    stride = a.strides[0]
            tid = (C.c_int32 * n)(*[int(t) for t in ob["track_ids"]])
            lab = (C.c_int32 * n)(*[int(t) for t in ob.get("labels", [0] * n)])
            arr[i].masks, arr[i].track_id, arr[i].label = C.cast(mp, C.POINTER(C.c_void_p)), tid, lab
            keep += [mp, tid, lab]
        return arr, stride, keep

    def TrackRGBD(self, gray, depth, timestamps=None, objects=None, on_device=False, gray_stride=None, depth_pitch=None, mask_stride=None):
        """gray: S uint8 arrays [H, W] (C-contiguous rows), depth: S float32 arrays [H, W] in metres; with on_device=True both are lists of device
        addresses with gray_stride (bytes) / depth_pitch (floats).  objects: the frames' semantic detections (see _objects) for
        ObjectOptimizer::PoseOptimization2 in TrackLocalMap."""
        if objects is not None:
            for i in range(self.S):
                self._gp[i] = gray[i] if on_device else gray[i].__array_interface__["data"][0]
                self._dp[i] = depth[i] if on_device else depth[i].__array_interface__["data"][0]
            ts = None if timestamps is None else np.ascontiguousarray(timestamps, np.float64)
            arr, ms, keep = self._objects(objects, on_device)
            gs = gray_stride if on_device else gray[0].strides[0]
            dpp = depth_pitch if on_device else depth[0].strides[0] // 4
            check(self.L.oslam_slam_track_rgbd_objects(self.h, self._gp, C.c_int(gs), self._dp, C.c_int(dpp), C.c_int(1 if on_device else 0),
                                                       ptr(ts) if ts is not None else None, arr, C.c_int(mask_stride if on_device else (ms or self.cfg.width)),
                                                       ptr(self.Tcw), ptr(self.state)))
            return self.Tcw, self.state
        if on_device:
            return self.TrackRGBD_device(gray, gray_stride, depth, depth_pitch, timestamps)
        for i in range(self.S):
            g, d = gray[i], depth[i]
            assert g.dtype == np.uint8 and d.dtype == np.float32 and g.strides[1] == 1 and d.strides[1] == 4
            self._gp[i] = g.__array_interface__["data"][0]
            self._dp[i] = d.__array_interface__["data"][0]
        ts = None
        if timestamps is not None:
            ts = np.ascontiguousarray(timestamps, np.float64)
        check(self.L.oslam_slam_track_rgbd(self.h, self._gp, C.c_int(gray[0].strides[0]), self._dp, C.c_int(depth[0].strides[0] // 4), C.c_int(0),
                                           ptr(ts) if ts is not None else None, ptr(self.Tcw), ptr(self.state)))
        return self.Tcw, self.state

    def TrackStereo(self, left, right, timestamps=None, on_device=False, stride=None, objects=None, mask_stride=None):
        """left / right: S uint8 arrays [H, W] (or device addresses with on_device=True and the row stride in bytes); objects: the frames' semantic detections
        (see _objects; Tracking::GrabImageStereo reads them per frame, src/Tracking.cc:229-232)."""
        if not hasattr(self, "_rp"):
            self._rp = (C.c_void_p * self.S)()
        for i in range(self.S):
            if on_device:
                self._gp[i], self._rp[i] = left[i], right[i]
            else:
                assert left[i].dtype == np.uint8 and right[i].dtype == np.uint8 and left[i].strides == right[i].strides
                self._gp[i] = left[i].__array_interface__["data"][0]
                self._rp[i] = right[i].__array_interface__["data"][0]
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, np.float64)
        st = stride if on_device else left[0].strides[0]
        if objects is not None:
            arr, ms, keep = self._objects(objects, on_device)
            ms = mask_stride if on_device else (ms or self.cfg.width)
            check(self.L.oslam_slam_track_stereo_objects(self.h, self._gp, self._rp, C.c_int(st), C.c_int(1 if on_device else 0),
                                                         ptr(ts) if ts is not None else None, arr, C.c_int(ms), ptr(self.Tcw), ptr(self.state)))
            return self.Tcw, self.state
        check(self.L.oslam_slam_track_stereo(self.h, self._gp, self._rp, C.c_int(st), C.c_int(1 if on_device else 0),
                                             ptr(ts) if ts is not None else None, ptr(self.Tcw), ptr(self.state)))
        return self.Tcw, self.state

    def TrackRGBD_device(self, gray_ptrs, gray_stride, depth_ptrs, depth_pitch, timestamps=None):
        """Device-resident inputs: lists of S device addresses (e.g. torch tensors' data_ptr())."""
        for i in range(self.S):
            self._gp[i] = gray_ptrs[i]
            self._dp[i] = depth_ptrs[i]
        ts = None
        if timestamps is not None:
            ts = np.ascontiguousarray(timestamps, np.float64)
        check(self.L.oslam_slam_track_rgbd(self.h, self._gp, C.c_int(gray_stride), self._dp, C.c_int(depth_pitch), C.c_int(1),
                                           ptr(ts) if ts is not None else None, ptr(self.Tcw), ptr(self.state)))
        return self.Tcw, self.state

    def _traj(self, fn, seq):
        n = C.c_int32(0)
        cap = 1 << 16
        st = np.zeros(cap, np.float64)
        T = np.zeros((cap, 3, 4), np.float32)
        check(fn(self.h, C.c_int(seq), C.c_int(cap), ptr(st), ptr(T), C.byref(n)))
        return st[:n.value].copy(), T[:n.value].copy()

    def finish(self):
        """Deferred schedule: applies the pending half of the last local-mapping pass (System::Shutdown's wait for the local mapper)."""
        check(self.L.oslam_slam_finish(self.h))

    def trajectory(self, seq):
        """(stamps, Twc[n,3,4]) as System::SaveTrajectoryTUM would write them (reference src/System.cc:378-440)."""
        return self._traj(self.L.oslam_slam_trajectory, seq)

    def keyframe_trajectory(self, seq):
        return self._traj(self.L.oslam_slam_keyframe_trajectory, seq)

    def stats(self, seq):
        out = np.zeros(16, np.int64)
        check(self.L.oslam_slam_stats(self.h, C.c_int(seq), ptr(out)))
        names = ("frames", "keyframes_created", "keyframes_in_map", "points_created", "points_in_map", "local_bas", "tracked_motion_model",
                 "tracked_reference_kf", "lost_frames", "points_fused", "points_triangulated", "keyframes_culled", "points_culled", "last_inliers",
                 "lba_edges", "map_violations")
        d = dict(zip(names, out.tolist()))
        so = np.zeros(8, np.int64)
        check(self.L.oslam_slam_object_stats(self.h, C.c_int(seq), ptr(so)))
        d.update(zip(("semantic_edges", "semantic_frames", "semantic_frames_nonzero", "object3ds", "object_points", "object2ds"), so[:6].tolist()))
        return d

    def lba_window_stats(self, seq):
        """Local-BA window sizes of one sequence since creation: windows and the sums of local / fixed keyframes, points and edges over them."""
        out = np.zeros(8, np.int64)
        check(self.L.oslam_slam_lba_window_stats(self.h, C.c_int(seq), ptr(out)))
        d = dict(zip(("windows", "local_kfs", "fixed_kfs", "points", "edges", "lba_windows_degraded"), out[:6].tolist()))
        d["operator_failures"] = int(out[7])
        return d

    def inject_failure(self, seq):
        """Test hook (include/oslam_slam.h oslam_slam_inject_failure): the next local-BA window of `seq` is refused by the operator."""
        check(self.L.oslam_slam_inject_failure(self.h, C.c_int(seq)))

    KT_GROUPS = ("frames", "pose_opt", "lba", "search", "fuse", "bow_triangulate", "mp_update", "other")

    def kernel_times(self, enable=True):
        """Device milliseconds / launches / algorithmic work (bytes, flop, flop, -) per kernel group since the last call (include/oslam_slam.h)."""
        out = np.zeros(len(self.KT_GROUPS) * 3, np.float64)
        check(self.L.oslam_slam_kernel_times(self.h, C.c_int(1 if enable else 0), ptr(out)))
        return {g: dict(ms=out[3 * i], launches=out[3 * i + 1], work=out[3 * i + 2]) for i, g in enumerate(self.KT_GROUPS)}

    def debug_point(self, seq, pid):
        """Test hook: (host record[64], resident record[64], bad) of map point `pid` (include/oslam_slam.h)."""
        a, b, bad = np.zeros(64, np.uint8), np.zeros(64, np.uint8), C.c_int32(0)
        check(self.L.oslam_slam_debug_point(self.h, seq, pid, ptr(a), ptr(b), C.byref(bad)))
        return a, b, bool(bad.value)

    def bad_keyframe_observations(self):
        """Observations in culled keyframes that ComputeDistinctiveDescriptors left out (include/oslam_slam.h)."""
        out = C.c_int64(0)
        check(self.L.oslam_slam_bad_keyframe_observations(self.h, C.byref(out)))
        return int(out.value)

    def local_map_reuse(self):
        """(frames whose local map was reused from the previous frame, tracked frames) summed over the handle's sequences."""
        out = np.zeros(2, np.int64)
        check(self.L.oslam_slam_local_map_reuse(self.h, ptr(out)))
        return int(out[0]), int(out[1])

    def stage_seconds(self, cpu=False):
        """Wall seconds per stage since creation; cpu=True: core-seconds (stepping thread + workers) of the same stages."""
        out = np.zeros(16, np.float64)
        check((self.L.oslam_slam_stage_cpu_seconds if cpu else self.L.oslam_slam_stage_seconds)(self.h, ptr(out)))
        names = ("frames", "search_last", "pose_opt", "search_local", "host_tracking", "mp_update", "lba", "host_mapping", "fuse_bow_triangulate",
                 "hm_process_kf", "hm_create_points", "hm_search_neighbors", "hm_lba_gather", "hm_kf_culling", "ht_initial_and_after", "ht_local_map")
        return dict(zip(names, out[:16].tolist()))
