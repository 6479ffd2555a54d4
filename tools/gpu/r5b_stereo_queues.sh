# intermittent slow stereo runs (frames stage 30x its device time): does the number of hardware queues matter?  stereo-only runs from the start of a fresh box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b_stereo_queues; mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload stereo --no-extras --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$? $(grep 'pre-roll 40' $O/$tag.err)"
  python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["value"], d["ms_per_step"], "frames_ms", d["roofline"]["groups"]["frames"]["device_ms"], "queues", d.get("gpu_max_hw_queues"), flush=True)
except Exception as e: print("$tag no line", e, flush=True)
PY
}
run svcdef_a OSLAM_X=0
run svclow_a OSLAM_LBA_SERVICE_PRIORITY=low
run svcdef_b OSLAM_X=0
run svclow_b OSLAM_LBA_SERVICE_PRIORITY=low
