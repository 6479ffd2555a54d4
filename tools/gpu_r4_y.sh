cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4y
mkdir -p $O
for m in -1 0 20 -1; do
  OSLAM_WAIT_SPIN_US=$m timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > $O/b_$m.json 2> $O/b_$m.err || exit 1
  python - <<PY
import json
d=json.loads([l for l in open("$O/b_$m.json") if l.startswith("{")][-1])
c=d["stage_core_seconds_timed_sum_over_handles"]; w=d["stage_seconds_timed_sum_over_handles"]
print("spin_us", $m, "frames/s", d["value"], "frac", d["roofline"]["frac"], "core_s", round(sum(v for k,v in c.items() if not k.startswith(("hm_","ht_"))),1), {k:(round(w[k],1),round(c[k],1)) for k in ("frames","pose_opt","fuse_bow_triangulate","host_tracking","host_mapping","mp_update")}, flush=True)
PY
done
