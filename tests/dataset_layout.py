"""Writes a synthetic sequence to disk in the datasets' layouts (TUM RGB-D + the fork's semantic directory, KITTI stereo) through object_slam_amd/io.py, for the
tests of the runners in examples/ (reference Examples/RGB-D/rgbd_tum.cc, Examples/Stereo/stereo_kitti.cc, src/Semantic.cc:14-96)."""
import os

import numpy as np

from object_slam_amd import io

TUM_FACTOR = 5000.0


def settings_text(cam, width, height, n_features, dataset_path=None, depth_factor=None, fps=30.0, dist=None):
    rows = ["%YAML:1.0", "", "# Camera calibration and distortion parameters (OpenCV)",
            "Camera.fx: %.6f" % cam["fx"], "Camera.fy: %.6f" % cam["fy"], "Camera.cx: %.6f" % cam["cx"], "Camera.cy: %.6f" % cam["cy"], ""]
    d = list(dist or [0, 0, 0, 0]) + [0.0] * 5
    rows += ["Camera.k1: %.6f" % d[0], "Camera.k2: %.6f" % d[1], "Camera.p1: %.6f" % d[2], "Camera.p2: %.6f" % d[3]]
    if dist is not None and len(dist) > 4:
        rows.append("Camera.k3: %.6f" % d[4])
    rows += ["", "Camera.width: %d" % width, "Camera.height: %d" % height, "", "# Camera frames per second", "Camera.fps: %.1f" % fps, "",
             "Camera.bf: %.6f" % cam["bf"], "Camera.RGB: 1", "ThDepth: %.1f" % cam["thDepth"]]
    if depth_factor is not None:
        rows.append("DepthMapFactor: %.1f" % depth_factor)
    rows += ["", "ORBextractor.nFeatures: %d" % n_features, "ORBextractor.scaleFactor: 1.2", "ORBextractor.nLevels: 8", "ORBextractor.iniThFAST: 20", "ORBextractor.minThFAST: 7"]
    if dataset_path is not None:
        rows += ["", 'DataSetPath: "%s"' % dataset_path, "MinSemanticConfidence: 0.5"]
    return "\n".join(rows) + "\n"


def write_tum_sequence(root, q, cam, n, labels=(56, 62, 41), t0=1311868164.363181, fps=30.0, rgb=True):
    """q: object_slam_amd.scene sequence (gray [n,H,W] u8, depth [n,H,W] f32 metres, masks [n,K,H,W] u8 {0,255}).  Returns (settings path, association path, stamps)."""
    H, W = q["gray"].shape[1:]
    stamps = [t0 + i / fps for i in range(n)]
    assoc = []
    for i, t in enumerate(stamps):
        name = "%.6f" % t
        g = q["gray"][i]
        io.write_png(os.path.join(root, "rgb", name + ".png"), np.stack([g, g, g], -1) if rgb else g)
        io.write_png(os.path.join(root, "depth", name + ".png"), np.rint(q["depth"][i].astype(np.float64) * TUM_FACTOR).clip(0, 65535).astype(np.uint16))
        assoc.append("%s rgb/%s.png %s depth/%s.png" % (name, name, name, name))
        if q.get("masks") is not None:
            sem = os.path.join(root, "semantic", "%f" % t)
            os.makedirs(sem, exist_ok=True)
            rows = []
            for k in range(q["masks"].shape[1]):
                m = q["masks"][i, k]
                ys, xs = np.nonzero(m)
                if len(xs) == 0:
                    continue
                io.write_png(os.path.join(sem, "%d.png" % k), m)
                rows.append("%d %.4f %d %d %d %d %d" % (labels[k % len(labels)], 0.9, xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, k))
            rows.append("56 0.2000 0 0 5 5 99")     # below MinSemanticConfidence: dropped (src/Semantic.cc:79)
            rows.append("1 0.9000 0 0 5 5 98")      # not a valid label (src/Semantic.cc:10): dropped
            open(os.path.join(sem, "%f.txt" % t), "w").write("\n".join(rows) + "\n")
    ap = os.path.join(root, "associations.txt")
    open(ap, "w").write("\n".join(assoc) + "\n")
    sp = os.path.join(root, "settings.yaml")
    open(sp, "w").write(settings_text(cam, W, H, 1000, dataset_path=root if q.get("masks") is not None else None, depth_factor=TUM_FACTOR, fps=fps))
    return sp, ap, np.array(stamps)


def write_kitti_sequence(root, left, right, cam, fps=10.0):
    n = len(left)
    for i in range(n):
        io.write_png(os.path.join(root, "image_0", "%06d.png" % i), left[i])
        io.write_png(os.path.join(root, "image_1", "%06d.png" % i), right[i])
    open(os.path.join(root, "times.txt"), "w").write("\n".join("%e" % (i / fps) for i in range(n)) + "\n")
    sp = os.path.join(root, "settings.yaml")
    H, W = left[0].shape
    open(sp, "w").write(settings_text(cam, W, H, 2000, fps=fps))
    return sp
