set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5ranks3
mkdir -p $O
OSLAM_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --seqs 2048 --handles 4 --no-extras --no-cpu-baseline > $O/rgbd.json 2> $O/rgbd.err || { tail -5 $O/rgbd.err; exit 1; }
OSLAM_BENCH_SHARE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --seqs 1024 --handles 4 --steps 5 --warmup 2 --preroll 30 > $O/torchrun.json 2> $O/torchrun.err || { tail -8 $O/torchrun.err; exit 1; }
python - <<PY
import json
for n in ("rgbd", "torchrun"):
    d=json.loads([l for l in open("$O/%s.json" % n) if l.startswith("{")][-1])
    print(n, d["value"], "frames/s, n_gpus", d["n_gpus"], "steps", d["steps"], "lost", d.get("lost_frames"), "violations", d.get("map_violations"), d["config"].get("sequences_per_gpu"), d["config"].get("sequences_per_gpu_reduced_from_for_host_memory"))
PY
