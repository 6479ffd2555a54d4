set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu/final_prof.sh
python bench.py > gpurun_out/r5final/bench.json 2> gpurun_out/r5final/bench.err || { tail -30 gpurun_out/r5final/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r5final/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"], "frac", r["frac"], "lba_alone", r["lba_alone"].get("frac_of_fp64_peak"), [x.get("frac_of_fp64_peak") for x in r["lba_alone_by_batch"]], "stereo", d["stereo"]["frames_per_s"], "bases32", d["bases32"]["frames_per_s"], "frontend", d["frontend"]["frames_per_s"], "failed", d["extras_failed"], "rss", d["host_max_rss_gb"])
PY
