# catch the intermittent slow start of the stereo workload on a fresh box: kernel trace of a short stereo run as the FIRST GPU process of the call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b_stereo_trace; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --workload stereo --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/stereo.json 2> $O/stereo.err
echo rc=$?
grep "pre-roll" $O/stereo.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-60s calls %6s avg %10.1f us max %10.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
try:
    d=json.loads(open("$O/stereo.json").read().strip().splitlines()[-1]); print("value", d["value"], d["ms_per_step"])
except Exception as e: print("no line", e)
PY
rm -f $O/kt/*kernel_trace.csv
