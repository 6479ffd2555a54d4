set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4o
run() {   # tag, lm, env...
  tag=$1; lm=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --lm $lm --no-extras --no-cpu-baseline > gpurun_out/r4o/bench_$tag.json 2> gpurun_out/r4o/bench_$tag.err || (tail -5 gpurun_out/r4o/bench_$tag.err; exit 1)
  python - <<PY
import json
d=json.load(open("gpurun_out/r4o/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], d["lba_windows_timed"]["windows"], {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()})
print({k:d["stage_seconds_timed_sum_over_handles"][k] for k in ("lba","mp_update","host_mapping","host_tracking","frames","pose_opt","fuse_bow_triangulate")})
PY
}
run q12 deferred GPU_MAX_HW_QUEUES=12
run q16 deferred GPU_MAX_HW_QUEUES=16
run q8 deferred GPU_MAX_HW_QUEUES=8
run sync8 sync GPU_MAX_HW_QUEUES=8
