import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from object_slam_amd import ORBextractor, synth
B = 8
frames, _ = synth.make_stream(B, 640, 480)
ex = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, max_batch=B)
import torch
d = torch.from_numpy(frames).cuda()
ex.extract_batch_device(d.data_ptr(), B, 640, 640 * 480, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
import ctypes as C
out = np.zeros((8192, 3), np.int32); n = C.c_int(0)
mx = []
for b in range(B):
    row = []
    for l in range(8):
        ex.L.oslam_orb_debug_get_candidates(ex.h, b, l, out.ctypes.data_as(C.c_void_p), 8192, C.byref(n)); row.append(n.value)
    mx.append(row)
print(np.array(mx).max(0), np.array(mx).mean(0))
