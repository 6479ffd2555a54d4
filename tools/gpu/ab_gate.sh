set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for V in 0.25 0 0.25 0; do
OSLAM_LBA_GATE_RELEASE=$V python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_ab_gate_$V.json 2> gpurun_out/r05_ab_gate.err || { tail -20 gpurun_out/r05_ab_gate.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab_gate_$V.json").read().strip().splitlines()[-1])
print("release=$V", d["value"], "lba ms", d["roofline"]["groups"]["lba"]["device_ms"], "wait", d["stage_seconds_timed_sum_over_handles"]["lba"])
PY
done
