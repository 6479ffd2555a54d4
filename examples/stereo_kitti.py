#!/usr/bin/env python3
"""Stereo runner in the reference's argv shape (Examples/Stereo/stereo_kitti.cc:33-125):

    python examples/stereo_kitti.py path_to_vocabulary path_to_settings path_to_sequence [--out DIR] [--no-sleep]

times.txt + image_0 / image_1 (LoadImages, :127-158), settings as Examples/Stereo/KITTI00-02.yaml, optional `<DataSetPath>/semantic/<%06d>/` (src/Semantic.cc:14-57);
ONE sequence through the HIP driver; median / mean tracking time; CameraTrajectory.txt in the KITTI format (src/System.cc:472-528)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from object_slam_amd import io, slam  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(usage="stereo_kitti.py path_to_vocabulary path_to_settings path_to_sequence")
    ap.add_argument("vocabulary"); ap.add_argument("settings"); ap.add_argument("sequence")
    ap.add_argument("--out", default=".")
    ap.add_argument("--no-sleep", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    left, right, stamps = io.load_kitti_sequence(a.sequence)
    st = io.load_settings(a.settings)
    cfg = io.config_from_settings(st, 1, slam.STEREO, device=a.device)
    sysm = slam.System(cfg)
    sem_path = (str(st["DataSetPath"]) + "/semantic/") if "DataSetPath" in st else None
    sem_th = float(st.get("MinSemanticConfidence", 0.5))
    rgb_order = bool(int(st.get("Camera.RGB", 1)))
    n = len(left)
    print("\n-------\nStart processing sequence ...\nImages in the sequence: %d\n" % n)
    times = np.zeros(n)
    for ni in range(n):
        imL, imR = io.read_image(left[ni]), io.read_image(right[ni])
        if imL.size == 0:
            print("\nFailed to load image at: %s" % left[ni], file=sys.stderr)
            return 1
        gl, gr = io.to_gray(imL, rgb_order), io.to_gray(imR, rgb_order)
        objs = None
        if sem_path:
            objs = io.detections_for_driver(io.read_semantic_kitti(sem_path, ni, sem_th), gl.shape[0], gl.shape[1])
        t1 = time.perf_counter()
        sysm.TrackStereo([gl], [gr], [stamps[ni]], objects=[objs] if objs else None)
        times[ni] = time.perf_counter() - t1
        T = (stamps[ni + 1] - stamps[ni]) if ni < n - 1 else (stamps[ni] - stamps[ni - 1] if ni > 0 else 0.0)
        if not a.no_sleep and times[ni] < T:
            time.sleep(T - times[ni])
    sysm.finish()
    srt = np.sort(times)
    print("-------\n\nmedian tracking time: %g\nmean tracking time: %g" % (srt[n // 2], times.sum() / n))
    os.makedirs(a.out, exist_ok=True)
    s, Twc = sysm.trajectory(0)
    io.save_trajectory_kitti_twc(os.path.join(a.out, "CameraTrajectory.txt"), Twc)
    print("\ntrajectory saved!  (%d frames; %s)" % (len(s), {k: v for k, v in sysm.stats(0).items() if k in ("keyframes_created", "local_bas", "lost_frames")}))
    sysm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
