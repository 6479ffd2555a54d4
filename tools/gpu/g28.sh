set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_MPU_ASYNC=1 OSLAM_SLAM_FUSECUR_CHECK=1 OSLAM_SLAM_CULL_CHECK=1 timeout -k 10 1100 python -m pytest tests/test_slam_driver_gpu.py tests/test_mp_table_gpu.py tests/test_examples_gpu.py -x -q > gpurun_out/r05_g28.log 2>&1 || { tail -30 gpurun_out/r05_g28.log; exit 1; }
tail -2 gpurun_out/r05_g28.log
run() {
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_ab_m.json 2> gpurun_out/r05_ab_m.err || { tail -20 gpurun_out/r05_ab_m.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab_m.json").read().strip().splitlines()[-1])
st=d["stage_seconds_timed_sum_over_handles"]; co=d["stage_core_seconds_timed_sum_over_handles"]
print("mpu_async=$OSLAM_SLAM_MPU_ASYNC", d["value"], "kf", d["keyframes"], "wall: mpu", st["mp_update"], "fuse", st["fuse_bow_triangulate"], "hm_sn", st["hm_search_neighbors"], "lba", st["lba"], "| core: mpu", co["mp_update"], "fuse", co["fuse_bow_triangulate"], "total", round(sum(v for k,v in co.items() if not k.startswith(("hm_","ht_"))),1))
PY
}
unset OSLAM_SLAM_MPU_ASYNC; run
OSLAM_SLAM_MPU_ASYNC=1 run
unset OSLAM_SLAM_MPU_ASYNC; run
OSLAM_SLAM_MPU_ASYNC=1 run
