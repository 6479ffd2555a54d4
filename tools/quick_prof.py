import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import ORBextractor, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H, NF = (640, 480, 1000) if len(sys.argv) < 3 or sys.argv[2] == 'tum' else (1241, 376, 2000)
frames, _ = synth.make_stream(B, W, H)
pitch = (W + 63) // 64 * 64
d = torch.zeros((B, H, pitch), dtype=torch.uint8, device='cuda')
d[:, :, :W] = torch.from_numpy(frames).cuda()
ex = ORBextractor(NF, 1.2, 8, 20, 7, W, H, max_batch=B)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ex.extract_batch_device(d.data_ptr(), B, pitch, pitch * H, st)
torch.cuda.synchronize()
ex.set_profiling(1)
t = time.time(); K = 10
for _ in range(K):
    ex.extract_batch_device(d.data_ptr(), B, pitch, pitch * H, st)
torch.cuda.synchronize()
dt = time.time() - t
ms, nb, ni = ex.get_profile()
k, _ = ex.fetch(0)
print("B=%d %dx%d: %.3f ms/batch, %.1f frames/s, kps[0]=%d" % (B, W, H, dt / K * 1e3, B * K / dt, len(k)))
names = ["pyramid", "fast", "blur", "octree", "orient+desc"]
for n, m in zip(names, ms):
    print("  %-12s %8.3f ms/batch  %7.3f us/frame" % (n, m / nb, m / ni * 1e3))
print("  alg bytes/frame", ex.algorithmic_bytes(len(k)), "-> GB/s", ex.algorithmic_bytes(len(k)) * B * K / dt / 1e9)
