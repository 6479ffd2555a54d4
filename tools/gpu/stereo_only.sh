set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --workload stereo --no-extras --no-cpu-baseline > gpurun_out/r05_stereo_only.json 2> gpurun_out/r05_stereo_only.err || { tail -20 gpurun_out/r05_stereo_only.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_stereo_only.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k:v["device_ms"] for k,v in d["roofline"]["groups"].items()})
PY
