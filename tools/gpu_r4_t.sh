cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4t
mkdir -p $O
for b in 8 32; do
  timeout -k 10 560 python bench.py --bases $b --no-extras --no-cpu-baseline > $O/bases_$b.json 2> $O/bases_$b.err; echo "rc=$?"
  python - <<PY
import json
line=[l for l in open("gpurun_out/r4t/bases_$b.json").read().splitlines() if l.startswith("{")][-1]
d=json.loads(line)
print("bases $b:", d["value"], d["ms_per_step"], d["config"]["distinct_streams_per_gpu"], d["config"]["replicas_per_stream"], {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()}, d["stage_seconds_timed_sum_over_handles"]["frames"], d["device_mem_used_gb_after_headline"], d["input_render_s"])
PY
done
