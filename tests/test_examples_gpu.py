"""GPU: the dataset runners in the reference's argv shape (examples/rgbd_tum.py, examples/stereo_kitti.py; reference Examples/RGB-D/rgbd_tum.cc:39-134,
Examples/Stereo/stereo_kitti.cc:33-125) on synthetic sequences written to disk in the datasets' layouts: the trajectory files must be what the driver
produces from the same (decoded) inputs handed over directly."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from object_slam_amd import io, scene, slam  # noqa: E402

pytestmark = pytest.mark.gpu


def test_rgbd_tum_runner_on_a_tum_layout_with_semantic_directory(tmp_path, capsys):
    import rgbd_tum
    from dataset_layout import write_tum_sequence
    n = 14
    q = scene.make_rgbd_sequence(0, n, speed=2.0)
    root = str(tmp_path / "seq")
    sp, ap, stamps = write_tum_sequence(root, q, slam.TUM2, n)
    out = str(tmp_path / "out")
    assert rgbd_tum.main(["ORBvoc.txt", sp, root, ap, "--out", out, "--no-sleep"]) == 0
    txt = capsys.readouterr().out
    assert "median tracking time:" in txt and "mean tracking time:" in txt and "Images in the sequence: %d" % n in txt
    rows = np.loadtxt(os.path.join(out, "CameraTrajectory.txt"), ndmin=2)
    krows = np.loadtxt(os.path.join(out, "KeyFrameTrajectory.txt"), ndmin=2)
    assert rows.shape == (n, 8) and krows.shape[1] == 8 and 1 <= len(krows) <= n
    assert np.allclose(rows[:, 0], stamps, atol=1e-6) and np.allclose(np.linalg.norm(rows[:, 4:], axis=1), 1.0, atol=1e-5)
    # the same decoded inputs handed to the driver directly
    st = io.load_settings(sp)
    sysm = slam.System(io.config_from_settings(st, 1, slam.RGBD))
    rgb, dep, ts = io.load_associations(ap)
    for i in range(n):
        gray = io.to_gray(io.read_image(os.path.join(root, rgb[i])), True)
        depth = io.depth_to_float(io.read_image(os.path.join(root, dep[i])), st["DepthMapFactor"])
        det = io.detections_for_driver(io.read_semantic_tum(root + "/semantic/", ts[i], 0.5), 480, 640)
        sysm.TrackRGBD([gray], [depth], [ts[i]], objects=[det])
    s, Twc = sysm.trajectory(0)
    assert sysm.stats(0)["semantic_edges"] > 0 and sysm.stats(0)["lost_frames"] == 0
    io.save_trajectory_tum_twc(str(tmp_path / "direct.txt"), s, Twc, 9)
    assert open(str(tmp_path / "direct.txt")).read() == open(os.path.join(out, "CameraTrajectory.txt")).read()
    # ATE of the written file against the scene's ground truth through the evaluation tool of io.py (evaluate_ate.py's definition)
    T0inv = np.linalg.inv(q["Twc"][0])
    gt = np.array([T0inv @ x for x in q["Twc"][:n]])
    assert io.horn_align_ate(rows[:, 1:4], gt[:, :3, 3]) < 0.02


def test_stereo_kitti_runner_on_a_kitti_layout(tmp_path, capsys):
    import stereo_kitti
    from dataset_layout import write_kitti_sequence
    from slam_common import make_stereo_streams
    n = 8
    left, right = make_stereo_streams(1, n)[0][:2]
    root = str(tmp_path / "00")
    sp = write_kitti_sequence(root, left, right, slam.KITTI00)
    out = str(tmp_path / "out")
    assert stereo_kitti.main(["ORBvoc.txt", sp, root, "--out", out, "--no-sleep"]) == 0
    assert "median tracking time:" in capsys.readouterr().out
    rows = np.loadtxt(os.path.join(out, "CameraTrajectory.txt"), ndmin=2)
    assert rows.shape == (n, 12)
    sysm = slam.System(io.config_from_settings(io.load_settings(sp), 1, slam.STEREO))
    for i in range(n):
        sysm.TrackStereo([np.ascontiguousarray(left[i])], [np.ascontiguousarray(right[i])], [i / 10.0])
    _, Twc = sysm.trajectory(0)
    assert np.allclose(rows.reshape(n, 3, 4), Twc, atol=5e-7)     # (the file holds 9 decimals)
