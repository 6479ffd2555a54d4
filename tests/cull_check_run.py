"""Child process of test_slam_driver_gpu.py::test_device_culling_counts_equal_the_host_walk: the driver reads its culling knobs (OSLAM_SLAM_CULL_CHECK,
OSLAM_SLAM_CULL_HOST) once per process, so each setting runs in a process of its own.  Prints one JSON line: per-sequence statistics and a digest of the poses."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from object_slam_amd import slam, synth  # noqa: E402
from slam_common import H, W, run  # noqa: E402


def main():
    n, S = 54, 2
    out = {}
    for name, lm in (("sync", slam.LM_SYNC), ("deferred", slam.LM_DEFERRED)):
        streams = [synth.make_occluded_stream(n, W, H, seed=sd) for sd in (11, 14)]
        sy = slam.System(slam.make_config(W, H, S, local_mapping=lm))
        poses, st = run(sy, streams, n)
        out[name] = {"stats": [sy.stats(s) for s in range(S)], "poses": hashlib.sha256(np.ascontiguousarray(poses).tobytes()).hexdigest(),
                     "status_ok": bool((st == slam.OK).all())}
        sy.close()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
