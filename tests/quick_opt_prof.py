import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import PoseOptimizer, LocalBundleAdjuster, synth
from oracle import oracle_py as O
B, N = 256, 1000
probs = [synth.make_pose_problem(100 + b, N=N) for b in range(B)]
t = lambda k, dt: torch.from_numpy(np.stack([p[k] for p in probs]).astype(dt)).cuda()
Tcw, Xw, obs, inv, has = t("Tcw", np.float32), t("Xw", np.float32), t("obs", np.float32), t("invSigma2", np.float32), t("has_mp", np.uint8)
po = PoseOptimizer(max_points=N, max_batch=B)
st = torch.cuda.current_stream().cuda_stream
for nb in (1, 16, 256):
    for _ in range(2):
        po.optimize_batch_device(nb, N, None, N, Tcw.data_ptr(), Xw.data_ptr(), obs.data_ptr(), inv.data_ptr(), has.data_ptr(), probs[0]["K"], st)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        po.optimize_batch_device(nb, N, None, N, Tcw.data_ptr(), Xw.data_ptr(), obs.data_ptr(), inv.data_ptr(), has.data_ptr(), probs[0]["K"], st)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print("pose-opt batch %d: %.3f ms/launch, %.1f us/frame" % (nb, dt * 1e3, dt / nb * 1e6))
t0 = time.time()
for p in probs[:10]:
    O.pose_optimization(p["Tcw"], p["Xw"], p["obs"], p["invSigma2"], p["has_mp"], p["K"])
print("oracle pose-opt: %.3f ms/frame" % ((time.time() - t0) / 10 * 1e3))
for (KL, KF, P) in ((20, 20, 4000), (6, 0, 200), (40, 60, 10000)):
    q = synth.make_lba_problem(1234, K_local=KL, K_fixed=KF, P=P)
    ba = LocalBundleAdjuster(max_keyframes=128, max_points=16384, max_edges=131072)
    args = (q["poses"], q["fixed"], q["points"], q["edge_kf"], q["edge_pt"], q["edge_obs"], q["edge_invSigma2"], q["K"])
    r = ba.LocalBundleAdjustment(*args)
    t0 = time.time(); r = ba.LocalBundleAdjustment(*args); dt = time.time() - t0
    ba.set_mode(False); ba.LocalBundleAdjustment(*args)
    t0 = time.time(); rc = ba.LocalBundleAdjustment(*args); dtc = time.time() - t0
    print("   compact mode %.2f ms, max |wide - compact| pose diff %.2e" % (dtc * 1e3, np.abs(rc[0] - r[0]).max()))
    t0 = time.time(); o = O.local_bundle_adjustment(*args); do = time.time() - t0
    print("LBA %d+%d KF, %d pts, %d edges: GPU %.2f ms (host API incl. copies) stats %s | oracle %.1f ms stats %s | max pose diff %.2e" % (
        KL, KF, P, len(q["edge_kf"]), dt * 1e3, r[3], do * 1e3, o[3], np.abs(r[0] - o[0]).max()))
    ba.close()
