set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for CFG in "8 8192" "16 8192" "12 8184"; do
set -- $CFG
python bench.py --no-extras --no-cpu-baseline --handles $1 --seqs $2 > gpurun_out/r05_ab_h$1.json 2> gpurun_out/r05_ab_h.err || { tail -20 gpurun_out/r05_ab_h.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab_h$1.json").read().strip().splitlines()[-1])
st=d["stage_seconds_timed_sum_over_handles"]; co=d["stage_core_seconds_timed_sum_over_handles"]
print("handles=$1", d["value"], "lba ms", d["roofline"]["groups"]["lba"]["device_ms"], "lba wait", st["lba"], "core-s", round(sum(v for k,v in co.items() if not k.startswith(("hm_","ht_"))),1))
PY
done
