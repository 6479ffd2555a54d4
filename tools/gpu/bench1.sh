set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py ${BENCH_ARGS:-} > gpurun_out/${OUT:-r05_bench_v1}.json 2> gpurun_out/${OUT:-r05_bench_v1}.err || { tail -20 gpurun_out/${OUT:-r05_bench_v1}.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/${OUT:-r05_bench_v1}.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"].get("frac"), d.get("stereo",{}).get("value"), d.get("frontend",{}).get("value"))
PY
