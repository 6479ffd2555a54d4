"""Single-frame latencies of the host-pointer C-ABI entry points (what a drop-in Tracking thread sees)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_slam_amd import ORBextractor, ORBmatcher, PoseOptimizer, synth

frames, offs = synth.make_stream(8, 640, 480)
ex = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480)
for _ in range(3): k, d = ex(frames[0])
def t(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
print("extract host->host  %.3f ms  (n=%d)" % (t(lambda: ex(frames[1])), len(k)))
d_img = torch.from_numpy(frames[:1]).cuda()
st = torch.cuda.current_stream().cuda_stream
def dev():
    ex.extract_batch_device(d_img.data_ptr(), 1, 640, 640 * 480, st); torch.cuda.synchronize()
print("extract device B=1  %.3f ms" % t(dev))
ex.set_profiling(1)
for _ in range(20): dev()
ms, nb, ni = ex.get_profile(); ex.set_profiling(0)
print("  kernel groups us:", [round(m / nb * 1e3, 1) for m in ms])
p = synth.make_pose_problem(1, N=1000)
po = PoseOptimizer(max_points=1000)
print("pose_optimize       %.3f ms" % t(lambda: po.PoseOptimization(p["Tcw"], p["Xw"], p["obs"], p["invSigma2"], p["has_mp"], p["K"])))
