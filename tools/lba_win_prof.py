"""Timing of batches of local-BA windows in the layouts of oslam_lba_set_mode (1: every LM trial of all windows as whole-GPU launches, 2: one workgroup per
window with the reduced system in LDS) on driver-shaped windows.  Usage: lba_win_prof.py [K_local K_fixed P track] ; kernel time from the handle's HIP events."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from object_slam_amd import LocalBundleAdjuster, synth
KL, KF, P, TR = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (27, 0, 1500, 13)))
NB = [int(x) for x in os.environ.get("NB", "1,8,40,256").split(",")]
cache = {}
for mode in [int(x) for x in os.environ.get("MODES", "1,2").split(",")]:
    ba = LocalBundleAdjuster(max_batch=max(NB), max_keyframes=64, max_points=8192, max_edges=65536)
    ba.set_mode(mode)
    if os.environ.get("SCHUR"):
        ba.set_schur(int(os.environ["SCHUR"]))
    if os.environ.get("SOLVER"):
        ba.set_solver(int(os.environ["SOLVER"]))
    ms, ln = C.c_double(0), C.c_longlong(0)
    for nb in NB:
        if nb not in cache:
            base = [synth.make_lba_problem(1234 + i, K_local=KL, K_fixed=KF, P=P, track=TR, stereo_frac=0.9) for i in range(min(nb, 8))]
            cache[nb] = [base[i % len(base)] for i in range(nb)]
        probs = cache[nb]
        ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
        ba.L.oslam_lba_kernel_time(ba.h, 1, C.byref(ms), C.byref(ln))
        t0 = time.time()
        reps = 2
        for _ in range(reps):
            out = ba.LocalBundleAdjustmentBatch(probs, probs[0]["K"])
        dt = (time.time() - t0) / reps
        ba.L.oslam_lba_kernel_time(ba.h, 1, C.byref(ms), C.byref(ln))
        E = len(probs[0]["edge_kf"])
        trials = out[0][3][1] + out[0][3][3]
        print("mode %d windows %3d: %8.2f ms host to host, %8.2f ms kernels (%d launches) per batch; window 0: %d edges, %d points, stats %s -> %.1f us per trial and window-slot"
              % (mode, nb, dt * 1e3, ms.value / reps, ln.value // reps, E, len(probs[0]["points"]), out[0][3], ms.value / reps * 1e3 / max(trials, 1)), flush=True)
    ba.close()
