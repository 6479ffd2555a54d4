cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b_headline_side; mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    g=d["roofline"]["groups"]
    print("$tag", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "frames_ms", g["frames"]["device_ms"], "lba_ms", g["lba"]["device_ms"], "frames stage", d["stage_seconds_timed_sum_over_handles"]["frames"], flush=True)
except Exception as e: print("$tag no line", e, flush=True)
PY
}
run svc_normal_a OSLAM_LBA_SERVICE_NO_PRIORITY=1
run svc_low_a OSLAM_X=0
run svc_normal_b OSLAM_LBA_SERVICE_NO_PRIORITY=1
