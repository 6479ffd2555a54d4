set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r05_gpu_suite.log; exit 1; }
tail -3 gpurun_out/r05_gpu_suite.log
O=gpurun_out/r5ranks2
mkdir -p $O
OSLAM_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --seqs 2048 --handles 4 --no-extras --no-cpu-baseline > $O/rgbd.json 2> $O/rgbd.err || { tail -5 $O/rgbd.err; exit 1; }
OSLAM_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --workload stereo --gpus 2 --seqs 512 --handles 4 --no-extras --no-cpu-baseline > $O/stereo.json 2> $O/stereo.err || { tail -5 $O/stereo.err; exit 1; }
python - <<PY
import json
for n in ("rgbd", "stereo"):
    d=json.loads([l for l in open("$O/%s.json" % n) if l.startswith("{")][-1])
    print(n, "2 ranks on one card:", d["value"], "frames/s, n_gpus", d["n_gpus"], "lost", d.get("lost_frames"), "violations", d.get("map_violations"))
PY
