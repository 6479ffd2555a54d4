cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4ranks
mkdir -p $O
OSLAM_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --seqs 2048 --handles 4 --no-extras --no-cpu-baseline > $O/rgbd.json 2> $O/rgbd.err || { tail -5 $O/rgbd.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("$O/rgbd.json") if l.startswith("{")][-1])
print("rgbd 2 ranks on one card:", d["value"], "frames/s, n_gpus", d["n_gpus"], "per rank elapsed", [r.get("elapsed_s") for r in d.get("per_rank", [])], "lost", d.get("lost_frames"), "violations", d.get("map_violations"))
PY
