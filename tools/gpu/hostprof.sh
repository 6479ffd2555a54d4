set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_LBA_HOSTPROF=1 OSLAM_LBA_SERVICE_STATS=1 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_hostprof.json 2> gpurun_out/r05_hostprof.err || { tail -20 gpurun_out/r05_hostprof.err; exit 1; }
grep "lba hostprof\|lba service" gpurun_out/r05_hostprof.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_hostprof.json").read().strip().splitlines()[-1])
print(d["value"], d["stage_core_seconds_timed_sum_over_handles"])
PY
