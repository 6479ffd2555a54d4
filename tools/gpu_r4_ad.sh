cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4ad
mkdir -p $O
for hn in 16 8 16; do
  timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline --handles $hn > $O/b_$hn.json 2> $O/b_$hn.err || { tail -5 $O/b_$hn.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("$O/b_$hn.json") if l.startswith("{")][-1])
c=d["stage_core_seconds_timed_sum_over_handles"]
print("handles", $hn, "frames/s", d["value"], "frac", d["roofline"]["frac"], "core_s", round(sum(v for k,v in c.items() if not k.startswith(("hm_","ht_"))),1), "rss", d["host_max_rss_gb"], flush=True)
PY
done
