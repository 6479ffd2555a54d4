"""The S2 front-end stage of bench.py alone (512 frames per step, ORBextractor + ORBmatcher), for `rocprofv3 --pmc` passes and kernel traces:
every launch of a front-end kernel in the trace is then the bench batch.  Usage: frontend_pmc.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from object_slam_amd import seqbench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
q = seqbench.rgbd_workload().make_sequence(0, 33)
print(json.dumps(bench.frontend_stage(q["gray"], q["Twc"], q["depth"], 0, steps)))
