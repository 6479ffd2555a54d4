cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4w
mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/r4w/bench_$tag.json"))
print("$tag:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()}, "lba wait", d["stage_seconds_timed_sum_over_handles"]["lba"])
PY
}
run sleep A=1
run spin OSLAM_LBA_SERVICE_SPIN_US=-1
run spin100 OSLAM_LBA_SERVICE_SPIN_US=100
run sleep2 A=1
