set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b_bench; mkdir -p $O
python bench.py --no-extras --no-cpu-baseline > $O/headline.json 2> $O/headline.err || { tail -20 $O/headline.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/headline.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("headline", d["value"], d["ms_per_step"], "frac", r["frac"], "launch_us", r.get("launch_us"), "lba_ms", r["groups"]["lba"]["device_ms"], "kf", d["keyframes"], "ate", d["ate_rmse_m"], "lba wait", d["stage_seconds_timed_sum_over_handles"]["lba"], flush=True)
print({k:v["device_ms"] for k,v in r["groups"].items()})
PY
python bench.py --workload stereo --no-extras --no-cpu-baseline > $O/stereo.json 2> $O/stereo.err || { tail -20 $O/stereo.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/stereo.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("stereo", d["value"], d["ms_per_step"], "lba_ms", r["groups"]["lba"]["device_ms"], "kf", d["keyframes"], "ate", d["ate_rmse_m"], flush=True)
PY
