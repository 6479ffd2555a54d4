"""Reduced-camera-system solvers of the wide local BA, per-kernel device time at several system orders: the LDS-resident scalar kernel (k_w_chol<true>,
n <= 132), the in-place global-memory scalar kernel (k_w_chol<false>) and the matrix-core kernel (k_w_chol_mfma).  Run under
`rocprofv3 --kernel-trace --stats --output-format csv`; one problem per (size, solver), 3 repetitions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import LocalBundleAdjuster, synth
ba = LocalBundleAdjuster(max_batch=4, max_keyframes=128, max_points=8192, max_edges=65536)
for KL, KF, P in ((8, 4, 1000), (20, 20, 4000), (40, 60, 10000)):
    q = synth.make_lba_problem(1234, K_local=KL, K_fixed=KF, P=P)
    args = (q["poses"], q["fixed"], q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"])
    for mode, name in ((2, "scalar"), (1, "mfma")):
        ba.set_solver(mode)
        ba.LocalBundleAdjustment(*args)
        t0 = time.time()
        for _ in range(3):
            out = ba.LocalBundleAdjustment(*args)
        print("n = %3d (%d+%d KF, %d points, %d edges) solver %-6s: %.2f ms host to host, trials %s" % (6 * KL, KL, KF, P, len(q["edge_kf"]), name, (time.time() - t0) / 3 * 1e3, out[3]))
