import sys, ctypes as C
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import ORBextractor, synth
B = 64
frames, _ = synth.make_stream(B, 640, 480)
d = torch.from_numpy(frames).cuda()
ex = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, max_batch=B)
st = torch.cuda.current_stream().cuda_stream
ex.extract_batch_device(d.data_ptr(), B, 640, 640 * 480, st)
out = (C.c_ulonglong * 16)()
ex.L.oslam_orb_debug_counters(ex.h, out, 1)
ex.extract_batch_device(d.data_ptr(), B, 640, 640 * 480, st)
ex.L.oslam_orb_debug_counters(ex.h, out, 1)
v = list(out)
names = ["tile load", "A quick test", "B score", "C nms", "D output"]
tot = sum(v[:5])
for n, x in zip(names, v[:5]): print("%-14s %12d cyc %5.1f%%" % (n, x, 100.0 * x / max(tot, 1)))
print("cells", v[6], "avg worklist", v[5] / max(v[6], 1), "avg px", v[7] / max(v[6], 1), "avg cycles/cell", tot / max(v[6], 1))
