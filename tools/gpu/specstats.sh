set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_SPEC_STATS=1 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_spec.json 2> gpurun_out/r05_spec.err || { tail -20 gpurun_out/r05_spec.err; exit 1; }
grep "fuse spec stats" gpurun_out/r05_spec.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_spec.json").read().strip().splitlines()[-1])
print(d["value"], "keyframes (whole run)", d["keyframes"], "local_bas", d["local_bas"])
PY
