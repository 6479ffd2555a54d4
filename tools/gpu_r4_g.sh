set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4g
run() {   # tag, lm, env...
  tag=$1; lm=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --lm $lm --no-extras --no-cpu-baseline > gpurun_out/r4g/bench_$tag.json 2> gpurun_out/r4g/bench_$tag.err || (tail -5 gpurun_out/r4g/bench_$tag.err; exit 1)
  python - <<PY
import json
d=json.load(open("gpurun_out/r4g/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], d["lba_windows_timed"]["windows"], {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()})
print({k:d["stage_seconds_timed_sum_over_handles"][k] for k in ("lba","mp_update","host_mapping","host_tracking","frames","pose_opt")})
PY
  grep "lba service" gpurun_out/r4g/bench_$tag.err | tail -1 || true
}
run def2prio deferred OSLAM_LBA_SERVICE_STATS=1
run def3prio deferred OSLAM_LBA_SERVICE_STATS=1 OSLAM_LBA_SERVICE_THREADS=3
run def3prio_win2 deferred OSLAM_LBA_SERVICE_STATS=1 OSLAM_LBA_SERVICE_THREADS=3 OSLAM_LBA_SERVICE_MODE=2
