"""The matrix-core reduced-system solver (k_w_chol_mfma) on the S5-large local BA (40 + 60 keyframes, 10 000 points: n = 240) and on a driver-shaped batch of
windows with 34 free keyframes (n = 204), for `rocprofv3 --pmc` passes (MFMA instruction / busy counters) and `--kernel-trace`.  One process, one stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import LocalBundleAdjuster, synth
ba = LocalBundleAdjuster(max_batch=40, max_keyframes=128, max_points=16384, max_edges=131072)
q = synth.make_lba_problem(1234, K_local=40, K_fixed=60, P=10000)
for _ in range(2):
    out = ba.LocalBundleAdjustment(q["poses"], q["fixed"], q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"])
print("S5-large: n = 240, %d edges, trials %s" % (len(q["edge_kf"]), out[3]))
base = [synth.make_lba_problem(1234 + i, K_local=34, K_fixed=0, P=2000, track=17, stereo_frac=0.9) for i in range(4)]
probs = [base[i % 4] for i in range(40)]
for _ in range(2):
    outs = ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
print("40 windows of 34 free keyframes: n = 204, %d edges each, trials %s" % (len(probs[0]["edge_kf"]), outs[0][3]))
