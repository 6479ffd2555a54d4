set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b_stereo; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_stereo_gpu.py -x -q -m gpu > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -1 $O/test.log
# stereo-shaped windows (10 keyframes, 2200 points, ~3.4 observations per point): scalar LDS solver (default) against the matrix-core solver (mode 4)
NB=100 MODES=1 timeout -k 10 200 python3 tools/lba_win_prof.py 10 0 2200 4 | tee $O/solver_default.txt
SOLVER=4 NB=100 MODES=1 timeout -k 10 200 python3 tools/lba_win_prof.py 10 0 2200 4 | tee $O/solver4.txt
NB=100 MODES=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 tools/lba_win_prof.py 10 0 2200 4 > $O/kt.log 2>&1
SOLVER=4 NB=100 MODES=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt4 -o kt -- python3 tools/lba_win_prof.py 10 0 2200 4 > $O/kt4.log 2>&1
for d in kt kt4; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo $d; head -8 $f | cut -d, -f1-4 | sed 's/(oslam::LbaProblem[^"]*"/"/' ; done
python bench.py --workload stereo --no-extras --no-cpu-baseline > $O/stereo.json 2> $O/stereo.err || { tail -20 $O/stereo.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/stereo.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("stereo", d["value"], d["ms_per_step"], {k:v["device_ms"] for k,v in r["groups"].items()}, "kf", d["keyframes"], flush=True)
PY
