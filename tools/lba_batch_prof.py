"""Phase cycles and timing of the compact LBA kernel (k_lba, one workgroup per window) on driver-sized windows.
Phase cycles need a profiling build (OSLAM_LBA_PROFILE=1 python -m object_slam_amd.build -f)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from object_slam_amd import LocalBundleAdjuster, synth
KL, KF, P, TR = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (6, 3, 1500, 3)))
ba = LocalBundleAdjuster(max_batch=64, max_keyframes=64, max_points=8192, max_edges=65536)
for nb in (1, 16, 64):
    probs = [synth.make_lba_problem(1234 + i, K_local=KL, K_fixed=KF, P=P, track=TR) for i in range(nb)]
    ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
    t0 = time.time()
    for _ in range(3): out = ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
    dt = (time.time() - t0) / 3
    print("windows %3d: %.2f ms per batch (host to host), edges %d, stats %s" % (nb, dt * 1e3, len(probs[0]["edge_kf"]), out[0][3]))
st = np.zeros(16, np.int32)
if hasattr(ba.L, "oslam_lba_debug_stats"):
    ba.L.oslam_lba_debug_stats(ba.h, st.ctypes.data_as(C.c_void_p))
    names = ["lin point-major", "lin pose-major", "reduce+Dinv", "schur", "cholesky+backsub", "landmarks+update", "eval+accept"]
    tot = st[8:15].sum()
    for n, v in zip(names, st[8:15]): print("%-20s %8d kcyc  %5.1f%%" % (n, v, 100.0 * v / max(tot, 1)))
