"""Two (or G) driver handles on one GPU, each advanced by its own host thread: the host bookkeeping of one group overlaps the kernels of the other."""
import sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from object_slam_amd import slam, synth
W, H, Z0 = 640, 480, 2.0
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
thr = int(sys.argv[4]) if len(sys.argv) > 4 else 8
base = [synth.make_stream(n, W, H, seed=11 + s, margin=1200) for s in range(8)]
d_base = [torch.from_numpy(b[0]).cuda() for b in base]
d_depth = torch.full((H, W), Z0, dtype=torch.float32, device="cuda")
systems = [slam.System(slam.make_config(W, H, S // G, host_threads=thr)) for _ in range(G)]
torch.cuda.synchronize()
def work(g):
    sysm = systems[g]; Sg = S // G
    dptr = [d_depth.data_ptr()] * Sg
    for t in range(n):
        sysm.TrackRGBD_device([d_base[(g * Sg + s) % 8][t].data_ptr() for s in range(Sg)], W, dptr, W, [t / 30.0] * Sg)
t0 = time.time()
ths = [threading.Thread(target=work, args=(g,)) for g in range(G)]
[t.start() for t in ths]; [t.join() for t in ths]
dt = time.time() - t0
print("S", S, "groups", G, "threads/group", thr, "fps", round(S * n / dt, 1), systems[0].stats(0)["keyframes_created"], systems[-1].stats(0)["map_violations"])
