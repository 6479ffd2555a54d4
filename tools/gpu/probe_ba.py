import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from object_slam_amd import LocalBundleAdjuster, synth
from oracle import oracle_py as O
for seed in (21, 22, 23, 24, 25, 26, 27, 28):
    q = synth.make_lba_problem(seed, K_local=8, K_fixed=0, P=400, outlier_frac=0.0)
    fixed = np.zeros(8, np.uint8); fixed[0] = 1
    opo, oxo = O.bundle_adjustment(q["poses"], fixed, q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"], 20, False)
    row = []
    for mode in (0, 1, 2):
        ba = LocalBundleAdjuster(max_keyframes=16, max_points=1024, max_edges=8192)
        ba.set_mode(mode)
        po, xo = ba.BundleAdjustment(q["poses"], fixed, q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"], 20, False)
        err = np.abs(xo - oxo).max(axis=1) / max(1.0, np.abs(oxo).max())
        row.append((float(np.abs(po - opo).max() / max(1.0, np.abs(opo).max())), float(err.max()), int((err > 1e-4).sum())))
        ba.close()
    print(seed, row, flush=True)
