"""Phase split of k_lba_win (one workgroup per window) for window 0 of a batch: needs a profiling build (OSLAM_LBA_PROFILE=1 python object_slam_amd/build.py -f).
Usage: lba_win_phases.py [K_local K_fixed P track] ; NB=windows in the batch."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from object_slam_amd import LocalBundleAdjuster, synth
KL, KF, P, TR = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (27, 0, 1500, 13)))
for nb in [int(x) for x in os.environ.get("NB", "1,256").split(",")]:
    base = [synth.make_lba_problem(1234 + i, K_local=KL, K_fixed=KF, P=P, track=TR, stereo_frac=0.9) for i in range(min(nb, 4))]
    probs = [base[i % len(base)] for i in range(nb)]
    ba = LocalBundleAdjuster(max_batch=nb, max_keyframes=64, max_points=8192, max_edges=65536)
    ba.set_mode(2)
    ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
    out = ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
    st = np.zeros(16, np.int32)
    ba.L.oslam_lba_debug_stats(ba.h, st.ctypes.data_as(C.c_void_p))
    names = ["lin (pose + point parts)", "pose sums + lambda", "schur: tile points + staging", "schur: block sums -> Hs", "cholesky + backsub", "landmarks + poses", "eval + accept + loop", "schur: tile pairs"]
    tot = int(st[8:16].sum())
    tr = out[0][3][1] + out[0][3][3]
    print("windows %d: window 0 has %d edges, stats %s; %d us in the kernel (%.1f us per trial)" % (nb, len(probs[0]["edge_kf"]), out[0][3], tot, tot / max(tr, 1)))
    for n, v in zip(names, st[8:16]):
        print("  %-28s %8d us  %5.1f%%" % (n, v, 100.0 * v / max(tot, 1)))
    ba.close()
