import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from object_slam_amd import slam
from slam_common import H, W, ate, make_streams, run
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
base = make_streams(min(S, 8), n)
streams = [base[s % len(base)] for s in range(S)]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg = slam.make_config(W, H, S, host_threads=thr)
t0 = time.time(); sysm = slam.System(cfg); print("create s", time.time() - t0)
t0 = time.time()
run(sysm, streams, n)
dt = time.time() - t0
print("S", S, "n", n, "fps", S * n / dt)
print(sysm.stats(0))
print(ate(sysm, cfg, streams, 0)[0])
print({k: round(v, 4) for k, v in sysm.stage_seconds().items()})
