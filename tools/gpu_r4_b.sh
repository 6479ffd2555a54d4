set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4b
timeout -k 10 1100 python -m pytest tests/test_slam_driver_gpu.py -q > gpurun_out/r4b/pytest_driver.log 2>&1 || (tail -40 gpurun_out/r4b/pytest_driver.log; exit 1)
tail -3 gpurun_out/r4b/pytest_driver.log
# A/B of the schedules on the headline workload (no extras, no CPU baseline)
for lm in sync deferred; do
  timeout -k 10 400 python bench.py --lm $lm --no-extras --no-cpu-baseline > gpurun_out/r4b/bench_$lm.json 2> gpurun_out/r4b/bench_$lm.err || (tail -5 gpurun_out/r4b/bench_$lm.err; exit 1)
  python - <<PY
import json
d=json.load(open("gpurun_out/r4b/bench_$lm.json"))
print("$lm", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], d["lba_windows_timed"], {k:v["device_ms"] for k,v in d["roofline"]["groups"].items()})
print(d["stage_seconds_timed_sum_over_handles"])
PY
done
