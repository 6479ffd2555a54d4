set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4final
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || (tail -40 $O/pytest_gpu.log; exit 1)
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/time.txt || (tail -20 $O/bench_default.err; exit 1)
cat $O/time.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4final/bench_default.json"))
print("value", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["bound"], d["config"]["local_mapping_schedule"])
print("lba_alone", d["roofline"].get("lba_alone"))
print("by batch", [(x["windows"], x["layout"][:12], x["ms_per_call"], x["frac_of_fp64_peak"]) for x in d["roofline"].get("lba_alone_by_batch")])
print("cpu", {k:d["cpu_baseline"][k] for k in ("value","three_core_bracket_frames_per_s","timed_frames","lba_windows_timed")})
s=d["stereo"]; print("stereo", s["frames_per_s"], s["regime"], s["roofline"]["frac"], s["lba_windows_timed"], s.get("cpu_baseline_stereo",{}).get("value"), s.get("cpu_baseline_stereo",{}).get("three_core_bracket_frames_per_s"))
print("host_inputs", d["host_inputs"] and d["host_inputs"].get("frames_per_s"), "frontend", d["frontend"] and d["frontend"].get("frames_per_s"))
print("groups", {k:round(v["device_ms"]) for k,v in d["roofline"]["groups"].items()})
print("rss", d["host_max_rss_gb"], "mem", d["device_mem_used_gb_after_headline"])
PY
