"""How k_octree's slot time splits over the pyramid levels (one workgroup per (image, level)): sum of the workgroups' durations per level (100 MHz ticks from the
kernel's own stamps, OrbCtx::dbg[8 + level]) on the S2 stream at 512 frames per launch, beside the launch's duration."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from object_slam_amd import ORBextractor, scene

B = int(os.environ.get("B", "512"))
q = scene.make_rgbd_sequence(0, 40, speed=1.0)
frames = np.ascontiguousarray(np.stack([q["gray"][i % 40] for i in range(B)]))
d = torch.from_numpy(frames).cuda()
ex = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, max_batch=B)
st = torch.cuda.current_stream().cuda_stream
out = (C.c_ulonglong * 16)()
for _ in range(3):
    ex.extract_batch_device(d.data_ptr(), B, 640, 640 * 480, st)
torch.cuda.synchronize()
ex.L.oslam_orb_debug_counters(ex.h, out, 1)
R = 10
t0 = time.perf_counter()
for _ in range(R):
    ex.extract_batch_device(d.data_ptr(), B, 640, 640 * 480, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
ex.L.oslam_orb_debug_counters(ex.h, out, 1)
v = [out[8 + l] for l in range(8)]
tot = sum(v)
print("extraction of %d frames: %.3f ms per batch" % (B, dt * 1e3))
for l in range(8):
    print("level %d: %8.1f us per workgroup (mean), %5.1f %% of the launch's workgroup time" % (l, v[l] / (R * B) / 100.0, 100.0 * v[l] / max(tot, 1)))
print("levels 4-7 together: %.1f %% of k_octree's workgroup time; all workgroups: %.1f ms of slot time per batch" % (100.0 * sum(v[4:]) / max(tot, 1), tot / R / 1e5))
