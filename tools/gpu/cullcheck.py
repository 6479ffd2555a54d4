import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from object_slam_amd import slam
from slam_common import H, W, make_streams, run
n, S = int(sys.argv[1]) if len(sys.argv) > 1 else 30, 3
streams = make_streams(S, n)
hip = slam.System(slam.make_config(W, H, S))
p, st = run(hip, streams, n)
print("ok", [hip.stats(s)["keyframes_culled"] for s in range(S)], [hip.stats(s)["keyframes_created"] for s in range(S)], flush=True)
