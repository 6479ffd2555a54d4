set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r05_gpu_suite.log; exit 1; }
tail -3 gpurun_out/r05_gpu_suite.log
O=gpurun_out/r5shared
mkdir -p $O
OSLAM_BENCH_SHARED_BASES=1 OSLAM_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --seqs 2048 --handles 4 --no-extras --no-cpu-baseline > $O/rgbd.json 2> $O/rgbd.err || { tail -5 $O/rgbd.err; exit 1; }
grep -i "shared\|unavailable" $O/rgbd.err | head -3
ls /dev/shm | grep oslam || echo "no segment left in /dev/shm"
python - <<PY
import json
d=json.loads([l for l in open("$O/rgbd.json") if l.startswith("{")][-1])
print("rgbd 2 ranks, shared base streams:", d["value"], "frames/s, per rank elapsed", [r.get("elapsed_s") for r in d.get("per_rank", [])], "keyframes per rank", [r.get("keyframes") for r in d.get("per_rank", [])], "lost", d.get("lost_frames"), "violations", d.get("map_violations"))
PY
