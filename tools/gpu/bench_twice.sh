set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for T in 1 0; do
if [ $T = 1 ]; then export OSLAM_BENCH_NO_TRIM=1; else unset OSLAM_BENCH_NO_TRIM; fi
python bench.py --no-cpu-baseline > gpurun_out/r05_bench_trim$T.json 2> gpurun_out/r05_bench_trim$T.err || { tail -20 gpurun_out/r05_bench_trim$T.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_bench_trim$T.json").read().strip().splitlines()[-1])
print("no_trim=$T", d["value"], "stereo", d["stereo"]["frames_per_s"], d["stereo"]["roofline"]["groups"]["frames"]["device_ms"], "b32", d["bases32"]["frames_per_s"], "host_in", d["host_inputs"]["frames_per_s"], "rss", d["host_max_rss_gb"])
PY
done
