cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4v
mkdir -p $O
for i in 1 2; do
  timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; echo "rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/r4v/bench_$i.json"))
print("run $i:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()}, "lba wait", d["stage_seconds_timed_sum_over_handles"]["lba"])
PY
done
