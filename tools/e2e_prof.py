"""cProfile of the single-sequence e2e harness on the HIP backend."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from object_slam_amd import e2e, synth
W, H, Z0 = 640, 480, 2.0
import bench
ef, eo = synth.make_stream(64, W, H, seed=11)
ecam = (bench.FX, bench.FY, bench.CX, bench.CY, bench.BF)
Z0 = bench.Z0
be = e2e.HipBackend(W, H)
e2e.run_sequence(be, ef[:8], eo[:8], ecam, Z0)
pr = cProfile.Profile()
pr.enable()
tr, dt, ate = e2e.run_sequence(be, ef, eo, ecam, Z0)
pr.disable()
print("fps %.1f ate %.5f" % (64 / dt, ate))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
