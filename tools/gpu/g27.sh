set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_LOCLIST_CHECK=1 OSLAM_SLAM_FUSECUR_CHECK=1 OSLAM_SLAM_CULL_CHECK=1 timeout -k 10 1100 python -m pytest tests/test_slam_driver_gpu.py tests/test_adapter_gpu.py tests/test_examples_gpu.py tests/test_mp_table_gpu.py -x -q -s > gpurun_out/r05_loclist_check.log 2>&1 || { grep -n "_CHECK\|Fatal\|rror" gpurun_out/r05_loclist_check.log | head -20; tail -30 gpurun_out/r05_loclist_check.log; exit 1; }
grep -n "_CHECK\|passed\|failed" gpurun_out/r05_loclist_check.log | head -10
run() {
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_ab_l.json 2> gpurun_out/r05_ab_l.err || { tail -20 gpurun_out/r05_ab_l.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab_l.json").read().strip().splitlines()[-1])
st=d["stage_seconds_timed_sum_over_handles"]; co=d["stage_core_seconds_timed_sum_over_handles"]
print("host=$OSLAM_SLAM_LOCLIST_HOST", d["value"], "kf", d["keyframes"], "ht_local_map core", co["ht_local_map"], "wall", st["ht_local_map"], "host_tracking core", co["host_tracking"], "core total", round(sum(v for k,v in co.items() if not k.startswith(("hm_","ht_"))),1), "reuse", d["local_map_reuse_frac"])
PY
}
OSLAM_SLAM_LOCLIST_HOST=1 run
unset OSLAM_SLAM_LOCLIST_HOST; run
OSLAM_SLAM_LOCLIST_HOST=1 run
unset OSLAM_SLAM_LOCLIST_HOST; run
