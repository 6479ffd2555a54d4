cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5ranks
mkdir -p $O
OSLAM_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --seqs 2048 --handles 4 --no-extras --no-cpu-baseline > $O/rgbd.json 2> $O/rgbd.err || { tail -5 $O/rgbd.err; exit 1; }
OSLAM_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --workload stereo --gpus 2 --seqs 512 --handles 4 --no-extras --no-cpu-baseline > $O/stereo.json 2> $O/stereo.err || { tail -5 $O/stereo.err; exit 1; }
python - <<PY
import json
for n in ("rgbd", "stereo"):
    d=json.loads([l for l in open("$O/%s.json" % n) if l.startswith("{")][-1])
    print(n, "2 ranks on one card:", d["value"], "frames/s, n_gpus", d["n_gpus"], "per rank elapsed", [r.get("elapsed_s") for r in d.get("per_rank", [])], "lost", d.get("lost_frames"), "violations", d.get("map_violations"))
PY
export HIP_FORCE_DEV_KERNARG=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bt -- python3 bench.py --seqs 1024 --handles 2 --preroll 200 --no-extras --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err
echo "rc=$? (trace)"
head -30 $O/trace/bt_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
