set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4c
run() {   # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 400 python bench.py --lm deferred --no-extras --no-cpu-baseline > gpurun_out/r4c/bench_$tag.json 2> gpurun_out/r4c/bench_$tag.err || (tail -5 gpurun_out/r4c/bench_$tag.err; exit 1)
  python - <<PY
import json
d=json.load(open("gpurun_out/r4c/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], d["lba_windows_timed"]["windows"], {k:v["device_ms"] for k,v in d["roofline"]["groups"].items()})
print({k:d["stage_seconds_timed_sum_over_handles"][k] for k in ("lba","mp_update","host_mapping","host_tracking","frames","pose_opt")})
PY
  grep "lba service" gpurun_out/r4c/bench_$tag.err | tail -1
}
run win2 OSLAM_LBA_SERVICE_STATS=1 OSLAM_LBA_SERVICE_MODE=2
run win2_cus128 OSLAM_LBA_SERVICE_STATS=1 OSLAM_LBA_SERVICE_MODE=2 OSLAM_LBA_SERVICE_CUS=128
run mixed200 OSLAM_LBA_SERVICE_STATS=1 OSLAM_LBA_SERVICE_MODE=1 OSLAM_LBA_SERVICE_MODE_BIG=2 OSLAM_LBA_SERVICE_BIG_FROM=200
