set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4q
run() {   # tag, lm, env...
  tag=$1; lm=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --lm $lm --no-extras --no-cpu-baseline > gpurun_out/r4q/bench_$tag.json 2> gpurun_out/r4q/bench_$tag.err || (tail -5 gpurun_out/r4q/bench_$tag.err; exit 1)
  python - <<PY
import json
d=json.load(open("gpurun_out/r4q/bench_$tag.json"))
print("$tag", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], d["lba_windows_timed"]["windows"], {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()})
print({k:d["stage_seconds_timed_sum_over_handles"][k] for k in ("lba","mp_update","host_mapping","host_tracking","frames","pose_opt","fuse_bow_triangulate")})
PY
}
run t2c1 deferred OSLAM_LBA_SERVICE_THREADS=2 OSLAM_LBA_CONCURRENCY=1
run t3c1 deferred OSLAM_LBA_SERVICE_THREADS=3 OSLAM_LBA_CONCURRENCY=1
run t2 deferred OSLAM_LBA_SERVICE_THREADS=2
run t2c1b deferred OSLAM_LBA_SERVICE_THREADS=2 OSLAM_LBA_CONCURRENCY=1
