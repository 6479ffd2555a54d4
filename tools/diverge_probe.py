"""Where do the HIP and the oracle operator tables part on the 200-frame S1 stream (tests/test_slam_driver_gpu.py), and by how much did the poses differ before?
usage: diverge_probe.py [sync|deferred]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from object_slam_amd import slam, scene
from slam_common import H, W, oracle_ops
lm = slam.LM_DEFERRED if (len(sys.argv) > 1 and sys.argv[1] == "deferred") else slam.LM_SYNC
n = 200
q = scene.make_rgbd_sequence(5, n, speed=1.0)
hip = slam.System(slam.make_config(W, H, 1, local_mapping=lm))
cfg_o = slam.make_config(W, H, 1, local_mapping=lm)
ora = slam.System(cfg_o, oracle_ops(cfg_o))
mx = 0.0
for t in range(n):
    objs = [dict(masks=[q["masks"][t, o] for o in range(3)], track_ids=q["track_ids"])]
    Th, _ = hip.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=objs)
    To, _ = ora.TrackRGBD([q["gray"][t]], [q["depth"][t]], [t / 30.0], objects=objs)
    d = float(np.abs(Th[0] - To[0]).max())
    a, b = hip.stats(0), ora.stats(0)
    diff = {k: (a[k], b[k]) for k in a if a[k] != b[k]}
    mx = max(mx, d)
    if diff or t % 20 == 0:
        print("frame %3d: |dT| %.3e (max so far %.3e) %s" % (t, d, mx, diff))
    if diff:
        break
