"""Timing of the stereo association (Frame::ComputeStereoMatches) on KITTI-shaped pairs: host API, one pair per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from object_slam_amd import ORBextractor, StereoMatcher, synth
W, H, NF = 1241, 376, 2000
left, right, _ = synth.make_stereo_stream(2, W, H, seed=21, margin=600, disparity=32)
exL, exR = ORBextractor(NF, 1.2, 8, 20, 7, W, H), ORBextractor(NF, 1.2, 8, 20, 7, W, H)
kL, dL = exL(left[0]); kR, dR = exR(right[0])
sm = StereoMatcher()
bf = 386.1448; b = bf / 718.856
for _ in range(3): uR, dep = sm.ComputeStereoMatches(exL, exR, kL, dL, kR, dR, bf, b)
t0 = time.time()
for _ in range(20): uR, dep = sm.ComputeStereoMatches(exL, exR, kL, dL, kR, dR, bf, b)
print("stereo match, 1 pair, %d/%d keypoints, host to host: %.3f ms, %d matched" % (len(kL), len(kR), (time.time() - t0) / 20 * 1e3, int((uR >= 0).sum())))
