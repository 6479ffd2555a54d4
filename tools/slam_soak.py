"""Long-run soak of the batch-of-sequences driver on the HIP operators: stability, map statistics, ATE."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from object_slam_amd import slam
from slam_common import H, W, ate, ate_stereo, make_streams, make_stereo_streams, run, run_stereo, stereo_config
mode = sys.argv[1] if len(sys.argv) > 1 else "rgbd"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
t0 = time.time()
if mode == "rgbd":
    streams = make_streams(S, n)
    cfg = slam.make_config(W, H, S, host_threads=16)
    sysm = slam.System(cfg)
    t1 = time.time(); _, st = run(sysm, streams, n); dt = time.time() - t1
    a = [ate(sysm, cfg, streams, s)[0] for s in range(S)]
else:
    streams = make_stereo_streams(S, n)
    cfg = stereo_config(S, host_threads=16)
    sysm = slam.System(cfg)
    t1 = time.time(); _, st = run_stereo(sysm, streams, n); dt = time.time() - t1
    a = [ate_stereo(sysm, cfg, streams, s)[0] for s in range(S)]
print(mode, "S", S, "n", n, "fps", round(S * n / dt, 1), "gen s", round(t1 - t0, 1), "all OK", bool((st == slam.OK).all()))
print("ATE max", max(a), "mean", float(np.mean(a)))
for s in range(min(S, 3)):
    print(s, sysm.stats(s))
print({k: round(v, 3) for k, v in sysm.stage_seconds().items()})
