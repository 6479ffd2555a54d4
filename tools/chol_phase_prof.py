"""Phase times inside k_w_chol (profiling build: OSLAM_LBA_PROFILE=1, library path via OSLAM_LIB_PATH)."""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import LocalBundleAdjuster, synth
q = synth.make_lba_problem(1234, K_local=20, K_fixed=20, P=4000)
ba = LocalBundleAdjuster(max_keyframes=128, max_points=16384, max_edges=131072)
args = (q["poses"], q["fixed"], q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"])
ba.LocalBundleAdjustment(*args)
a = np.zeros(16, np.int32); b = np.zeros(16, np.int32)
ba.L.oslam_lba_debug_stats(ba.h, a.ctypes.data_as(C.c_void_p))
r = ba.LocalBundleAdjustment(*args)
ba.L.oslam_lba_debug_stats(ba.h, b.ctypes.data_as(C.c_void_p))
d = (b - a)[8:12].astype(np.float64) * 10.0 / 1e3   # 100 MHz ticks -> us
launches = sum(r[3][1::2]) if len(r[3]) >= 4 else 15
print("chol launches (trials):", launches)
for n, v in zip(["stage to LDS", "first panel", "factor loop", "back-substitution"], d): print("%-18s %8.1f us total  %6.2f us/launch" % (n, v, v / max(launches, 1)))
