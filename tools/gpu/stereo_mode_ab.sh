# A/B of the local-BA service layout on the stereo workload (10-keyframe windows): multi-launch (mode 1, default) against one workgroup per window (mode 2)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {  # tag, env...
  tag=$1; shift
  env "$@" python bench.py --workload stereo --no-extras --no-cpu-baseline > gpurun_out/r05_stereo_ab_$tag.json 2> gpurun_out/r05_stereo_ab_$tag.err || { tail -20 gpurun_out/r05_stereo_ab_$tag.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_stereo_ab_$tag.json").read().strip().splitlines()[-1])
g=d["roofline"]["groups"]
print("$tag", d["value"], d["ms_per_step"], "lba_ms", g["lba"]["device_ms"], "lba wait", d["stage_seconds_timed_sum_over_handles"]["lba"], "kf", d["keyframes"], "ate", d["ate_rmse_m"], flush=True)
PY
}
run m1_a OSLAM_X=0
run m2_a OSLAM_LBA_SERVICE_MODE=2
run m1_b OSLAM_X=0
run m2_b OSLAM_LBA_SERVICE_MODE=2
