import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import PoseOptimizer, synth
B, N = int(os.environ.get("POSE_PROF_B", "1024")), int(sys.argv[1]) if len(sys.argv) > 1 else 1000
probs = [synth.make_pose_problem(100 + b, N=N) for b in range(B)]
t = lambda k, dt: torch.from_numpy(np.stack([p[k] for p in probs]).astype(dt)).cuda()
Tcw, Xw, obs, inv, has = t("Tcw", np.float32), t("Xw", np.float32), t("obs", np.float32), t("invSigma2", np.float32), t("has_mp", np.uint8)
po = PoseOptimizer(max_points=N, max_batch=B)
st = torch.cuda.current_stream().cuda_stream
for nb in sorted(set([1, 64, 256, 512, 1024, B])):
    for _ in range(2):
        po.optimize_batch_device(nb, N, None, N, Tcw.data_ptr(), Xw.data_ptr(), obs.data_ptr(), inv.data_ptr(), has.data_ptr(), probs[0]["K"], st)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        po.optimize_batch_device(nb, N, None, N, Tcw.data_ptr(), Xw.data_ptr(), obs.data_ptr(), inv.data_ptr(), has.data_ptr(), probs[0]["K"], st)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print("pose-opt batch %d: %.3f ms/launch, %.1f us/frame" % (nb, dt * 1e3, dt / nb * 1e6))
if hasattr(po.L, "oslam_pose_debug_profile"):   # profiling build (OSLAM_EXTRA_FLAGS=-DOSLAM_POSE_PROFILE): phase cycles of frame 0
    import ctypes as C
    po.optimize_batch_device(1, N, None, N, Tcw.data_ptr(), Xw.data_ptr(), obs.data_ptr(), inv.data_ptr(), has.data_ptr(), probs[0]["K"], st)
    out = (C.c_ulonglong * 8)()
    po.L.oslam_pose_debug_profile(out)
    names = ["build pass", "sum of 28", "solve + exp", "eval pass", "sum of 1", "classify", "prologue", "total"]
    for n_, v in zip(names, out): print("%-12s %10d cyc %5.1f%%" % (n_, v, 100.0 * v / max(out[7], 1)))
