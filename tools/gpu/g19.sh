cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_CULL_CHECK=1 timeout -k 10 1100 python -m pytest tests/test_slam_driver_gpu.py tests/test_adapter_gpu.py tests/test_examples_gpu.py tests/test_mp_table_gpu.py -x -q -s > gpurun_out/r05_cullcheck.log 2>&1
grep -n "CULL_CHECK\|Fatal\|passed\|failed" gpurun_out/r05_cullcheck.log | head -10
for V in 0 1; do
if [ $V = 1 ]; then export OSLAM_SLAM_CULL_HOST=1; else unset OSLAM_SLAM_CULL_HOST; fi
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_ab_cull$V.json 2> gpurun_out/r05_ab_cull.err || { tail -20 gpurun_out/r05_ab_cull.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab_cull$V.json").read().strip().splitlines()[-1])
co=d["stage_core_seconds_timed_sum_over_handles"]; st=d["stage_seconds_timed_sum_over_handles"]
print("cull_host=$V", d["value"], "hm_kf_culling core-s", co["hm_kf_culling"], "wall", st["hm_kf_culling"], "host_mapping", co["host_mapping"], "keyframes", d["keyframes"])
PY
done
