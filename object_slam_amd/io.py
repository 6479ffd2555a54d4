"""Data formats either side of the hot path: host-side Python, no device work.

* association file reader                 reference Examples/RGB-D/rgbd_tum.cc:144-169 (LoadImages)
* TUM / KITTI trajectory writers          reference src/System.cc:378-432, :435-470, :472-528
* KITTI ground-truth pose reader          ExpResults/KITTI/groundtruth/NN.txt (rows of a 3x4 T_w_c)
* semantic directory reader               reference src/Semantic.cc:14-96
* timestamp association + ATE evaluation  reference ExpResults/TUM/Localization/associate.py:71-101, evaluate_ate.py:47-83
"""
import os

import numpy as np

VALID_LABELS_TUM = (0, 39, 41, 56, 58, 62, 63, 64, 65, 66, 73, 77)   # src/Semantic.cc:10
VALID_LABELS_KITTI = (2,)                                             # src/Semantic.cc:11


def load_associations(path):
    """rgbd_tum.cc LoadImages: rows `t_rgb rgb_file t_depth depth_file`; the frame timestamp is the RGB one.
    Returns (rgb_files, depth_files, timestamps float64)."""
    rgb, dep, ts = [], [], []
    with open(path) as f:
        for s in f.read().split("\n"):
            if not s:
                continue
            tok = s.split()
            ts.append(float(tok[0]))
            rgb.append(tok[1] if len(tok) > 1 else "")
            dep.append(tok[3] if len(tok) > 3 else "")
    return rgb, dep, np.asarray(ts, np.float64)


def load_kitti_poses(path):
    """KITTI ground truth: each row 12 floats = row-major 3x4 [R|t] of T_w_c.  Returns [n][4][4] float64."""
    rows = np.loadtxt(path, dtype=np.float64, ndmin=2)
    T = np.tile(np.eye(4), (len(rows), 1, 1))
    T[:, :3, :] = rows.reshape(-1, 3, 4)
    return T


def _quat_xyzw(R):
    """Eigen::Quaterniond(Matrix3d) (Converter::toQuaternion, src/Converter.cc:138-150): x, y, z, w as float32."""
    R = np.asarray(R, np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)   # w, x, y, z
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (R[2, 1] - R[1, 2]) * t
        q[2] = (R[0, 2] - R[2, 0]) * t
        q[3] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return np.array([q[1], q[2], q[3], q[0]]).astype(np.float32)


def _rwc_twc(Tcw):
    """Rwc = Rcw.t(); twc = -Rwc*tcw  (one cv::gemm, alpha = -1, float accumulation: src/System.cc:424-425)."""
    T = np.asarray(Tcw, np.float32)
    Rwc = T[:3, :3].T.copy()
    t = T[:3, 3]
    twc = np.empty(3, np.float32)
    for r in range(3):
        t0 = np.float32(np.float32(np.float32(Rwc[r, 0] * t[0]) + np.float32(Rwc[r, 1] * t[1])) + np.float32(Rwc[r, 2] * t[2]))
        twc[r] = np.float32(np.float64(t0) * -1.0)
    return Rwc, twc


def _fx(v, prec):
    return "%.*f" % (prec, float(v))   # ostream << fixed << setprecision(prec) << float (promoted to double)


def save_trajectory_tum(path, timestamps, Tcw_list, lost=None):
    """System::SaveTrajectoryTUM line format (src/System.cc:429): `t tx ty tz qx qy qz qw`, t with 6 decimals,
    the rest with 9; lost frames skipped.  Tcw_list = final per-frame camera poses (the relative-pose chaining of
    :408-422 is the caller's bookkeeping)."""
    with open(path, "w") as f:
        for i, (t, T) in enumerate(zip(timestamps, Tcw_list)):
            if lost is not None and lost[i]:
                continue
            Rwc, twc = _rwc_twc(T)
            q = _quat_xyzw(Rwc)
            f.write(_fx(t, 6) + " " + " ".join(_fx(v, 9) for v in (twc[0], twc[1], twc[2], q[0], q[1], q[2], q[3])) + "\n")


def save_keyframe_trajectory_tum(path, timestamps, Tcw_list, bad=None):
    """System::SaveKeyFrameTrajectoryTUM (src/System.cc:435-470): keyframes sorted by id, bad ones skipped; 7 decimals."""
    with open(path, "w") as f:
        for i, (t, T) in enumerate(zip(timestamps, Tcw_list)):
            if bad is not None and bad[i]:
                continue
            Rwc, Ow = _rwc_twc(T)   # GetCameraCenter = -Rwc*tcw (src/KeyFrame.cc:74)
            q = _quat_xyzw(Rwc)
            f.write(_fx(t, 6) + " " + " ".join(_fx(v, 7) for v in (Ow[0], Ow[1], Ow[2], q[0], q[1], q[2], q[3])) + "\n")


def save_trajectory_kitti(path, Tcw_list):
    """System::SaveTrajectoryKITTI (src/System.cc:521-523): row-major 3x4 [Rwc|twc], 9 decimals, every frame."""
    with open(path, "w") as f:
        for T in Tcw_list:
            Rwc, twc = _rwc_twc(T)
            vals = []
            for r in range(3):
                vals += [Rwc[r, 0], Rwc[r, 1], Rwc[r, 2], twc[r]]
            f.write(" ".join(_fx(v, 9) for v in vals) + "\n")


def _read_mask(path):
    from PIL import Image   # cv::imread(path, -1): unchanged bit depth / channels
    if not os.path.exists(path):
        return np.zeros((0, 0), np.uint8)   # imread returns an empty Mat
    return np.asarray(Image.open(path))


def _read_semantic_dir(semanticpath, filename, valid, prob_threshold, remap_63):
    out = []
    txt = os.path.join(semanticpath, filename + ".txt")
    if not os.path.exists(txt):
        return out
    with open(txt) as f:
        for line in f.read().split("\n"):
            if not line:
                continue
            tok = line.split()
            label = int(tok[0])
            if remap_63 and label == 63:   # src/Semantic.cc:75-78
                label = 62
            prob = float(np.float32(tok[1]))
            if prob <= prob_threshold:
                continue
            x, y, w, h, inst = (int(v) for v in tok[2:7])
            if label in valid:
                out.append({"label": label, "prob": prob, "x": x, "y": y, "w": w, "h": h,
                            "mask": _read_mask(os.path.join(semanticpath, "%d.png" % inst))})
    return out


def read_semantic_tum(path, timestamp, prob_threshold):
    """Semantic::ReadSemanticTUMRGBD (src/Semantic.cc:59-96): directory `<path><to_string(timestamp)>/`, i.e. "%f"."""
    name = "%f" % timestamp
    return _read_semantic_dir(path + name, name, VALID_LABELS_TUM, prob_threshold, True)


def read_semantic_kitti(path, kitti_id, prob_threshold):
    """Semantic::ReadSemanticKittiStereo (src/Semantic.cc:14-57): directory `<path><%06d frame id>/`."""
    name = "%06d" % kitti_id
    return _read_semantic_dir(path + name, name, VALID_LABELS_KITTI, prob_threshold, False)


def read_file_list(path):
    """associate.py:50-69: `stamp d1 d2 ...` rows, '#' comments, commas / tabs as separators -> {stamp: [fields]}."""
    out = {}
    with open(path) as f:
        for line in f.read().replace(",", " ").replace("\t", " ").split("\n"):
            if len(line) > 0 and line[0] != "#":
                v = [s.strip() for s in line.split(" ") if s.strip() != ""]
                if len(v) > 1:
                    out[float(v[0])] = v[1:]
    return out


def associate(first_stamps, second_stamps, offset=0.0, max_difference=0.02):
    """associate.py:71-101: all pairs closer than max_difference, sorted by (difference, a, b), greedily matched;
    result sorted by a.  Windowed candidate generation instead of the reference's n*m scan; same output."""
    a = np.asarray(sorted(first_stamps), np.float64)
    b = np.asarray(sorted(second_stamps), np.float64)
    cand = []
    lo = 0
    for x in a:
        while lo < len(b) and (b[lo] + offset) <= x - max_difference - 1e-9:
            lo += 1
        j = lo
        while j < len(b) and (b[j] + offset) < x + max_difference + 1e-9:
            d = abs(x - (b[j] + offset))
            if d < max_difference:
                cand.append((d, float(x), float(b[j])))
            j += 1
    cand.sort()
    ua, ub, matches = set(), set(), []
    for d, x, y in cand:
        if x not in ua and y not in ub:
            ua.add(x)
            ub.add(y)
            matches.append((x, y))
    matches.sort()
    return matches


def align_horn(model, data):
    """evaluate_ate.py:47-83: rigid alignment of model (3xn) onto data (3xn); returns rot, trans, per-point error."""
    model = np.asarray(model, np.float64)
    data = np.asarray(data, np.float64)
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
    return rot, trans, np.sqrt((err * err).sum(0))


def horn_align_ate(est_xyz, gt_xyz):
    """ATE RMSE of an estimate [n,3] against ground truth [n,3] after the rigid (no scale) Horn alignment of evaluate_ate.py:47-83,161 (model = estimate,
    data = ground truth): the number bench.py and the driver tests report."""
    _, _, e = align_horn(np.asarray(est_xyz, np.float64).T, np.asarray(gt_xyz, np.float64).T)
    return float(np.sqrt((e * e).mean()))


def evaluate_ate(gt_file, est_file, offset=0.0, scale=1.0, max_difference=0.02):
    """evaluate_ate.py main (:118-161): associate ground truth (first) with the estimate (second), align the estimate
    onto the ground truth, return the statistics it prints (rmse, mean, median, std, min, max) and the pair count."""
    first, second = read_file_list(gt_file), read_file_list(est_file)
    matches = associate(list(first.keys()), list(second.keys()), offset, max_difference)
    if len(matches) < 2:
        raise ValueError("Couldn't find matching timestamp pairs between groundtruth and estimated trajectory!")
    first_xyz = np.array([[float(v) for v in first[a][0:3]] for a, _ in matches]).T
    second_xyz = np.array([[float(v) * scale for v in second[b][0:3]] for _, b in matches]).T
    _, _, e = align_horn(second_xyz, first_xyz)
    return {"pairs": len(e), "rmse": float(np.sqrt(np.dot(e, e) / len(e))), "mean": float(e.mean()), "median": float(np.median(e)),
            "std": float(e.std()), "min": float(e.min()), "max": float(e.max())}


# ---- dataset side of the runners in examples/ (reference Examples/RGB-D/rgbd_tum.cc, Examples/Stereo/stereo_kitti.cc, src/Tracking.cc:60-170, :195-275) ----
def load_settings(path):
    """The settings file the reference reads with cv::FileStorage (`%YAML:1.0`, `Key.sub: value` rows, Examples/RGB-D/TUM2.yaml): {key: float | str}."""
    out = {}
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if not line or line.startswith("%") or line.startswith("---") or ":" not in line:
                continue
            k, v = line.split(":", 1)
            v = v.strip().strip('"')
            try:
                out[k.strip()] = float(v)
            except ValueError:
                out[k.strip()] = v
    return out


def config_from_settings(st, n_sequences=1, sensor=None, **kw):
    """oslam_slam_config_t from the reference's settings keys (src/Tracking.cc:60-170): Camera.fx .. Camera.k3, Camera.bf, ThDepth, Camera.fps, ORBextractor.*."""
    from . import slam
    cam = dict(fx=st["Camera.fx"], fy=st["Camera.fy"], cx=st["Camera.cx"], cy=st["Camera.cy"], bf=st["Camera.bf"], thDepth=st.get("ThDepth", 40.0), fps=st.get("Camera.fps", 30.0) or 30.0)
    dist = [st.get("Camera.k1", 0.0), st.get("Camera.k2", 0.0), st.get("Camera.p1", 0.0), st.get("Camera.p2", 0.0)]
    if st.get("Camera.k3", 0.0) != 0.0:     # src/Tracking.cc:96-101: k3 only when non-zero
        dist.append(st["Camera.k3"])
    if not any(dist):
        dist = None
    return slam.make_config(int(st["Camera.width"]), int(st["Camera.height"]), n_sequences, cam=cam, dist=dist, nFeatures=int(st["ORBextractor.nFeatures"]),
                            scaleFactor=float(st["ORBextractor.scaleFactor"]), nLevels=int(st["ORBextractor.nLevels"]), iniThFAST=int(st["ORBextractor.iniThFAST"]),
                            minThFAST=int(st["ORBextractor.minThFAST"]), sensor=slam.RGBD if sensor is None else sensor, **kw)


def read_image(path):
    """cv::imread(path, CV_LOAD_IMAGE_UNCHANGED): uint8 [H,W] / [H,W,3|4] or uint16 [H,W] (TUM depth); an unreadable file -> an empty array."""
    from PIL import Image
    if not os.path.exists(path):
        return np.zeros((0, 0), np.uint8)
    im = Image.open(path)
    a = np.asarray(im)
    if a.dtype == np.int32:      # PIL mode "I" for 16-bit PNGs
        a = a.astype(np.uint16)
    return a


def write_png(path, a):
    """uint8 gray / RGB or uint16 gray PNG (for writing synthetic sequences in the datasets' layouts)."""
    from PIL import Image
    a = np.asarray(a)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if a.dtype == np.uint16:
        Image.fromarray(a.astype(np.uint16)).save(path)     # mode I;16
    else:
        Image.fromarray(a.astype(np.uint8)).save(path)


def to_gray(img, rgb_order=True):
    """cv::cvtColor(.., CV_RGB2GRAY / CV_BGR2GRAY) on 8-bit images as Tracking::GrabImageRGBD applies it by Camera.RGB (src/Tracking.cc:248-262): OpenCV's 14-bit
    fixed point, Y = (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14.  Gray images pass through."""
    a = np.asarray(img)
    if a.ndim == 2:
        return np.ascontiguousarray(a.astype(np.uint8))
    c = a[..., :3].astype(np.int64)
    r, g, b = (c[..., 0], c[..., 1], c[..., 2]) if rgb_order else (c[..., 2], c[..., 1], c[..., 0])
    return np.ascontiguousarray(((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8))


def depth_to_float(imD, depth_map_factor):
    """imDepth.convertTo(CV_32F, mDepthMapFactor) with mDepthMapFactor = 1 / DepthMapFactor (src/Tracking.cc:163-167, :264-265): float32 product, saturate-free."""
    a = np.asarray(imD)
    if a.dtype == np.float32 and abs(depth_map_factor - 1.0) <= 1e-5:
        return np.ascontiguousarray(a)
    f = np.float32(1.0) / np.float32(depth_map_factor) if abs(depth_map_factor) >= 1e-5 else np.float32(1.0)
    return np.ascontiguousarray((a.astype(np.float64) * np.float64(f)).astype(np.float32))


def load_kitti_sequence(path):
    """stereo_kitti.cc LoadImages (Examples/Stereo/stereo_kitti.cc:127-158): times.txt, image_0/%06d.png, image_1/%06d.png."""
    ts = [float(s) for s in open(os.path.join(path, "times.txt")).read().split()]
    left = [os.path.join(path, "image_0", "%06d.png" % i) for i in range(len(ts))]
    right = [os.path.join(path, "image_1", "%06d.png" % i) for i in range(len(ts))]
    return left, right, np.asarray(ts, np.float64)


def save_trajectory_tum_twc(path, stamps, Twc, prec=9):
    """System::SaveTrajectoryTUM (prec 9, src/System.cc:429) / SaveKeyFrameTrajectoryTUM (prec 7, :466) rows from [R_wc | t_wc] (oslam_slam_trajectory's output)."""
    with open(path, "w") as f:
        for t, T in zip(stamps, np.asarray(Twc, np.float32)):
            q = _quat_xyzw(T[:3, :3])
            f.write(_fx(t, 6) + " " + " ".join(_fx(v, prec) for v in (T[0, 3], T[1, 3], T[2, 3], q[0], q[1], q[2], q[3])) + "\n")


def save_trajectory_kitti_twc(path, Twc):
    """System::SaveTrajectoryKITTI (src/System.cc:521-523) rows from [R_wc | t_wc]."""
    with open(path, "w") as f:
        for T in np.asarray(Twc, np.float32):
            f.write(" ".join(_fx(v, 9) for v in T[:3, :4].reshape(-1)) + "\n")


def detections_for_driver(sem, height, width):
    """The driver's detection input (include/oslam_slam.h: masks + the CALLER's track id per detection) from one frame's semantic entries (read_semantic_*).
    Association across frames is the out-of-scope object layer of the reference (ObjectMatcher); the substitute used by the runners: the k-th detection of a label,
    ordered by its box's x, keeps track id label * 100 + k."""
    per = {}
    out = dict(masks=[], track_ids=[], labels=[])
    for e in sorted(sem, key=lambda e: (e["label"], e["x"])):
        m = np.asarray(e["mask"])
        if m.ndim == 3:
            m = m[..., 0]
        if m.shape != (height, width):
            continue
        k = per.get(e["label"], 0)
        per[e["label"]] = k + 1
        out["masks"].append(np.ascontiguousarray(np.where(m > 0, 255, 0).astype(np.uint8)))
        out["track_ids"].append(e["label"] * 100 + k)
        out["labels"].append(e["label"])
    return out if out["masks"] else None
