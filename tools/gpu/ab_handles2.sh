set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {
python bench.py --no-extras --no-cpu-baseline --handles $1 --seqs 8192 > gpurun_out/r05_ab_x.json 2> gpurun_out/r05_ab_x.err || { tail -20 gpurun_out/r05_ab_x.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab_x.json").read().strip().splitlines()[-1])
st=d["stage_seconds_timed_sum_over_handles"]; co=d["stage_core_seconds_timed_sum_over_handles"]
print("handles=$1 threads=$OSLAM_BENCH_HOST_THREADS", d["value"], "lba ms", d["roofline"]["groups"]["lba"]["device_ms"], "lba wait", st["lba"], "core-s", round(sum(v for k,v in co.items() if not k.startswith(("hm_","ht_"))),1))
PY
}
OSLAM_BENCH_HOST_THREADS=3 run 8
OSLAM_BENCH_HOST_THREADS=4 run 4
OSLAM_BENCH_HOST_THREADS=3 true
