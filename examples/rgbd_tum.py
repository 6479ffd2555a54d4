#!/usr/bin/env python3
"""RGB-D runner in the reference's argv shape (Examples/RGB-D/rgbd_tum.cc:39-134):

    python examples/rgbd_tum.py path_to_vocabulary path_to_settings path_to_sequence path_to_association [--out DIR] [--no-sleep]

Reads the settings YAML (Examples/RGB-D/TUM2.yaml keys), the association file (LoadImages, :144-169), RGB / depth PNGs and — when the settings name a DataSetPath —
the semantic directory `<DataSetPath>/semantic/<timestamp>/` (src/Tracking.cc:70-74, src/Semantic.cc:57-96); runs ONE sequence through the HIP driver (oslam_slam);
prints the median / mean tracking time like the reference (:126-134) and writes CameraTrajectory.txt and KeyFrameTrajectory.txt (src/System.cc:378-470).
path_to_vocabulary is accepted for argv compatibility: the DBoW2 vocabulary is not used (substitute node assignment)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from object_slam_amd import io, slam  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(usage="rgbd_tum.py path_to_vocabulary path_to_settings path_to_sequence path_to_association")
    ap.add_argument("vocabulary"); ap.add_argument("settings"); ap.add_argument("sequence"); ap.add_argument("association")
    ap.add_argument("--out", default=".", help="directory of CameraTrajectory.txt / KeyFrameTrajectory.txt")
    ap.add_argument("--no-sleep", action="store_true", help="do not wait for the next frame's timestamp (the reference sleeps: rgbd_tum.cc:111-118)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    rgb, dep, stamps = io.load_associations(a.association)
    if not rgb:
        print("\nNo images found in provided path.", file=sys.stderr)
        return 1
    if len(rgb) != len(dep):
        print("\nDifferent number of images for rgb and depth.", file=sys.stderr)
        return 1
    st = io.load_settings(a.settings)
    cfg = io.config_from_settings(st, 1, slam.RGBD, device=a.device)
    sysm = slam.System(cfg)
    sem_path = (str(st["DataSetPath"]) + "/semantic/") if "DataSetPath" in st else None
    sem_th = float(st.get("MinSemanticConfidence", 0.5))
    rgb_order = bool(int(st.get("Camera.RGB", 1)))
    dfac = float(st.get("DepthMapFactor", 1.0))
    print("\n-------\nStart processing sequence ...\nImages in the sequence: %d\n" % len(rgb))
    times = np.zeros(len(rgb))
    for ni in range(len(rgb)):
        im = io.read_image(os.path.join(a.sequence, rgb[ni]))
        imD = io.read_image(os.path.join(a.sequence, dep[ni]))
        if im.size == 0:
            print("\nFailed to load image at: %s/%s" % (a.sequence, rgb[ni]), file=sys.stderr)
            return 1
        gray = io.to_gray(im, rgb_order)
        depth = io.depth_to_float(imD, dfac)
        objs = None
        if sem_path:
            objs = io.detections_for_driver(io.read_semantic_tum(sem_path, stamps[ni], sem_th), gray.shape[0], gray.shape[1])
        t1 = time.perf_counter()
        sysm.TrackRGBD([gray], [depth], [stamps[ni]], objects=[objs] if objs else None)
        times[ni] = time.perf_counter() - t1
        T = (stamps[ni + 1] - stamps[ni]) if ni < len(rgb) - 1 else (stamps[ni] - stamps[ni - 1] if ni > 0 else 0.0)
        if not a.no_sleep and times[ni] < T:
            time.sleep(T - times[ni])
    sysm.finish()   # System::Shutdown
    srt = np.sort(times)
    print("-------\n\nmedian tracking time: %g\nmean tracking time: %g" % (srt[len(srt) // 2], times.sum() / len(times)))
    os.makedirs(a.out, exist_ok=True)
    s, Twc = sysm.trajectory(0)
    io.save_trajectory_tum_twc(os.path.join(a.out, "CameraTrajectory.txt"), s, Twc, 9)
    ks, kT = sysm.keyframe_trajectory(0)
    io.save_trajectory_tum_twc(os.path.join(a.out, "KeyFrameTrajectory.txt"), ks, kT, 7)
    print("\ntrajectory saved!  (%d frames, %d keyframes; %s)" % (len(s), len(ks), {k: v for k, v in sysm.stats(0).items() if k in ("local_bas", "lost_frames", "semantic_edges")}))
    sysm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
