import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import LocalBundleAdjuster, synth
q = synth.make_lba_problem(1234, K_local=20, K_fixed=20, P=4000)
ba = LocalBundleAdjuster(max_keyframes=128, max_points=16384, max_edges=131072)
args = (q["poses"], q["fixed"], q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"])
ba.LocalBundleAdjustment(*args); ba.LocalBundleAdjustment(*args)
st = np.zeros(16, np.int32)
ba.L.oslam_lba_debug_stats(ba.h, st.ctypes.data_as(C.c_void_p))
names = ["lin point-major", "lin pose-major", "reduce+Dinv", "schur", "cholesky+backsub", "landmarks+update", "eval+accept"]
tot = st[8:15].sum()
for n, v in zip(names, st[8:15]): print("%-20s %8d kcyc  %5.1f%%" % (n, v, 100.0 * v / max(tot, 1)))
print("total kcycles", tot, "=> %.2f ms at 2.1 GHz" % (tot * 1e3 / 2.1e9 * 1e3))
