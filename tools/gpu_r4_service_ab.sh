cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4af
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" OSLAM_LBA_SERVICE_STATS=1 timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/b_$tag.json 2> $O/b_$tag.err || { tail -3 $O/b_$tag.err; return 1; }
  python - <<PY
import json
d=json.loads([l for l in open("$O/b_$tag.json") if l.startswith("{")][-1])
c=d["stage_core_seconds_timed_sum_over_handles"]; w=d["stage_seconds_timed_sum_over_handles"]
print("$tag", "frames/s", d["value"], "frac", d["roofline"]["frac"], "lba ms", round(d["roofline"]["groups"]["lba"]["device_ms"]), "lba wait", round(w["lba"],1), "core_s", round(sum(v for k,v in c.items() if not k.startswith(("hm_","ht_"))),1), flush=True)
PY
  grep "lba service" $O/b_$tag.err | tail -1
}
run noprio OSLAM_LBA_SERVICE_NO_PRIORITY=1
run base A=1
