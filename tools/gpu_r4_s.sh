cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s
mkdir -p $O
export OSLAM_BENCH_SHARE_GPU=1
timeout -k 10 500 python bench.py --gpus 2 --seqs 2048 --no-extras --no-cpu-baseline > $O/rgbd_2ranks.json 2> $O/rgbd_2ranks.err; echo "rc=$?"
timeout -k 10 500 python bench.py --gpus 2 --workload stereo --seqs 512 --handles 4 --no-extras --no-cpu-baseline > $O/stereo_2ranks.json 2> $O/stereo_2ranks.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("rgbd_2ranks","stereo_2ranks"):
    try:
        d=json.load(open("gpurun_out/r4s/%s.json"%f))
        print(f, d["value"], d["n_gpus"], d["ms_per_step"], d["roofline"]["frac"], d["per_rank"], d["config"]["local_mapping_schedule"])
    except Exception as ex:
        print(f, "failed", ex); print(open("gpurun_out/r4s/%s.err"%f).read()[-800:])
PY
