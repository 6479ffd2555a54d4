import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from object_slam_amd import PoseOptimizer, synth
po = PoseOptimizer(max_points=1000, max_batch=1)
for sd in (100,101,102,103):
    p = synth.make_pose_problem(sd, N=1000)
    r = po.PoseOptimization(p["Tcw"], p["Xw"], p["obs"], p["invSigma2"], p["has_mp"], p["K"])
    print(sd, "inliers", r[0], "its/trials", r[3])
