cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4z
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_slam_driver_gpu.py tests/test_mp_table_gpu.py -x -q > $O/pytest_drv.log 2>&1 || { tail -30 $O/pytest_drv.log; exit 1; }
tail -2 $O/pytest_drv.log
for m in 1 1; do
  OSLAM_SLAM_MPU_FUSED=$m timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/b_$m.json 2> $O/b_$m.err || exit 1
  python - <<PY
import json
d=json.loads([l for l in open("$O/b_$m.json") if l.startswith("{")][-1])
c=d["stage_core_seconds_timed_sum_over_handles"]; w=d["stage_seconds_timed_sum_over_handles"]; g=d["roofline"]["groups"]
print("fused=$m frames/s", d["value"], "frac", d["roofline"]["frac"], "core_s", round(sum(v for k,v in c.items() if not k.startswith(("hm_","ht_"))),1), "mp_update wall", round(w["mp_update"],1), "dev", round(g["mp_update"]["device_ms"]), g["mp_update"]["launches"], "lba wait", round(w["lba"],1), flush=True)
PY
done
