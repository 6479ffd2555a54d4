cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4p
for cfg in "512 4" "1024 4" "2048 8" "2048 4"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --workload stereo --seqs $1 --handles $2 --no-extras --no-cpu-baseline > gpurun_out/r4p/st_$1_$2.json 2> gpurun_out/r4p/st_$1_$2.err || { tail -3 gpurun_out/r4p/st_$1_$2.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r4p/st_$1_$2.json"))
print("stereo seqs $1 handles $2:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["device_mem_used_gb_after_headline"], d["host_max_rss_gb"], {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()})
PY
done
