"""LBA S5 under rocprofv3 --kernel-trace --stats: per-kernel durations of the wide-mode LM schedule."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_slam_amd import LocalBundleAdjuster, synth
q = synth.make_lba_problem(1234, K_local=20, K_fixed=20, P=4000)
ba = LocalBundleAdjuster(max_keyframes=128, max_points=16384, max_edges=131072)
args = (q["poses"], q["fixed"], q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"])
ba.LocalBundleAdjustment(*args)
t0 = time.time()
for _ in range(5): r = ba.LocalBundleAdjustment(*args)
print("LBA S5 %.2f ms/call stats %s" % ((time.time() - t0) / 5 * 1e3, r[3]))
