"""Phase times inside k_w_chol_lds_mfma on a steady-state-shaped window (profiling build: OSLAM_LBA_PROFILE=1 python object_slam_amd/build.py -f)."""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import LocalBundleAdjuster, synth
KL = int(sys.argv[1]) if len(sys.argv) > 1 else 27
q = synth.make_lba_problem(1234, K_local=KL, K_fixed=0, P=1500, track=13, stereo_frac=0.9)
ba = LocalBundleAdjuster(max_keyframes=64, max_points=8192, max_edges=65536)
args = (q["poses"], q["fixed"], q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"])
ba.LocalBundleAdjustment(*args)
a = np.zeros(16, np.int32); b = np.zeros(16, np.int32)
ba.L.oslam_lba_debug_stats(ba.h, a.ctypes.data_as(C.c_void_p))
r = ba.LocalBundleAdjustment(*args)
ba.L.oslam_lba_debug_stats(ba.h, b.ctypes.data_as(C.c_void_p))
d = (b - a)[8:13].astype(np.float64) * 10.0 / 1e3   # 100 MHz ticks -> us
launches = r[3][1] + r[3][3]
print("n =", 6 * int((q["fixed"] == 0).sum()), "chol launches (trials):", launches)
for n, v in zip(["load to LDS", "diagonal factors", "row panels", "trailing (MFMA)", "back-substitution"], d): print("%-18s %8.1f us total  %6.2f us/launch" % (n, v, v / max(launches, 1)))
sub = (b - a)[13:16].astype(np.float64) * 10.0 / 1e3
for n, v in zip(["  wave 0: tile (0, 0)", "  wave 0: factor", "  wave 1: its strips"], sub): print("%-22s %8.1f us total  %6.2f us/launch" % (n, v, v / max(launches, 1)))
