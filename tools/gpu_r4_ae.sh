cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4ae
mkdir -p $O
timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1])
c=d["stage_core_seconds_timed_sum_over_handles"]
print("frames/s", d["value"], "frac", d["roofline"]["frac"], "core_s", round(sum(v for k,v in c.items() if not k.startswith(("hm_","ht_"))),1), {k:(round(v["device_ms"]),v["launches"]) for k,v in d["roofline"]["groups"].items()})
PY
