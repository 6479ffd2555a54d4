cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4ab
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_poseopt_gpu.py -x -q > $O/pytest_pose.log 2>&1 || { tail -40 $O/pytest_pose.log; exit 1; }
tail -2 $O/pytest_pose.log
POSE_PROF_B=4096 timeout -k 10 200 python tools/pose_prof.py 2>&1 | grep "batch"
timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1])
print("frames/s", d["value"], "frac", d["roofline"]["frac"], {k:(round(v["device_ms"]),v["launches"]) for k,v in d["roofline"]["groups"].items()})
PY
