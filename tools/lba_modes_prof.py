"""Batched local BA on driver-sized windows: wide mode (every LM trial of ALL windows as six whole-GPU launches) against compact mode (one workgroup
per window).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.  Usage: lba_modes_prof.py [K_local K_fixed P track windows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import LocalBundleAdjuster, synth
KL, KF, P, TR, NB = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 4, 1000, 4, 28)))
ba = LocalBundleAdjuster(max_batch=64, max_keyframes=128, max_points=8192, max_edges=65536)
MODES = (1,) if KL + KF > 64 else (1, 0)   # the compact kernel takes minutes on big windows
probs = [synth.make_lba_problem(1234 + i, K_local=KL, K_fixed=KF, P=P, track=TR) for i in range(NB)]
for mode in MODES:
    ba.set_mode(mode)
    ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
    t0 = time.time()
    for _ in range(5):
        out = ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
    dt = (time.time() - t0) / 5
    print("%s: %d windows (%d+%d KF, %d points, %d edges): %.2f ms per batch host to host, stats %s" % ("wide" if mode else "compact", NB, KL, KF, P, len(probs[0]["edge_kf"]), dt * 1e3, out[0][3]))
