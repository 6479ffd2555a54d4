set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r05_gpu_suite.log; exit 1; }
tail -3 gpurun_out/r05_gpu_suite.log
run() {
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_ab_k.json 2> gpurun_out/r05_ab_k.err || { tail -20 gpurun_out/r05_ab_k.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab_k.json").read().strip().splitlines()[-1])
st=d["stage_seconds_timed_sum_over_handles"]; co=d["stage_core_seconds_timed_sum_over_handles"]
print("eager=$OSLAM_SLAM_EAGER_DESC", d["value"], "kf", d["keyframes"], "frames wall", st["frames"], "core", co["frames"], "dev", d["roofline"]["groups"]["frames"]["device_ms"])
PY
}
OSLAM_SLAM_EAGER_DESC=1 run
unset OSLAM_SLAM_EAGER_DESC; run
OSLAM_SLAM_EAGER_DESC=1 run
unset OSLAM_SLAM_EAGER_DESC; run
