set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5final2
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["stereo"]["frames_per_s"], d["bases32"]["frames_per_s"], d["frontend"].get("frames_per_s"), d["extras_failed"])
PY
export HIP_FORCE_DEV_KERNARG=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bt -- python3 bench.py --seqs 1024 --handles 2 --preroll 200 --no-extras --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err
echo "rc=$? (trace)"
grep -n "copyBuffer\|k_copy_to_host" $O/trace/bt_kernel_stats.csv | cut -d, -f1-5
