set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_ab_p.json 2> gpurun_out/r05_ab_p.err || { tail -20 gpurun_out/r05_ab_p.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab_p.json").read().strip().splitlines()[-1])
st=d["stage_seconds_timed_sum_over_handles"]
print("prio=$OSLAM_LBA_SERVICE_PRIORITY", d["value"], "lba ms", d["roofline"]["groups"]["lba"]["device_ms"], "frac", d["roofline"]["frac"], "lba wait", st["lba"], "frames wall", st["frames"], "fuse", st["fuse_bow_triangulate"], "mpu", st["mp_update"])
PY
}
OSLAM_LBA_SERVICE_PRIORITY=low run
OSLAM_LBA_SERVICE_PRIORITY=high run
OSLAM_LBA_SERVICE_PRIORITY=low run
OSLAM_LBA_SERVICE_PRIORITY=high run
