set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_SN_STATS=1 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_sn.json 2> gpurun_out/r05_sn.err || { tail -20 gpurun_out/r05_sn.err; exit 1; }
grep "search-neighbors" gpurun_out/r05_sn.err | awk '{a+=$4; b+=$6; c+=$10; d+=$13; e+=$15} END {print "sum over handles (whole run, 225 steps): targets+masks", a, "rounds", b, "second-direction list", c, "its round", d, "updates+connections", e}'
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_sn.json").read().strip().splitlines()[-1])
print(d["value"], d["stage_core_seconds_sum_over_handles"]["hm_search_neighbors"], d["stage_core_seconds_timed_sum_over_handles"]["hm_search_neighbors"])
PY
