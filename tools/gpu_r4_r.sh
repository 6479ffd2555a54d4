set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4r
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_slam_driver_gpu.py -q -k "deferred" > $O/pytest.log 2>&1 || (tail -30 $O/pytest.log; exit 1)
tail -2 $O/pytest.log
timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err || (tail -5 $O/bench.err; exit 1)
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4r/bench.json"))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()})
PY
for sch in sync deferred; do
  timeout -k 10 500 python tests/soak_s1.py 2 1000 1 $sch > $O/soak_$sch.log 2>&1 || (tail -5 $O/soak_$sch.log; exit 1)
  tail -1 $O/soak_$sch.log > $O/soak_$sch.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r4r/soak_$sch.json"))
print("$sch", d["hip_frames_per_s"], [ (p["ate_rmse_m"], p["lost_frames"], p["map_violations"], p["keyframes_created"], p["keyframes_culled"]) for p in d["per_sequence"]], d.get("hip_vs_oracle_seq0"), d.get("first_stat_difference"), d["oracle_seq0"]["ate_rmse_m"])
PY
done
