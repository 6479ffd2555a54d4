cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4soak
mkdir -p $O
for sch in sync deferred; do
  OSLAM_SLAM_VOTE_CHECK=1 timeout -k 10 500 python tests/soak_s1.py 2 1000 1 $sch > $O/soak_$sch.log 2>&1 || { tail -5 $O/soak_$sch.log; exit 1; }
  tail -1 $O/soak_$sch.log > $O/soak_$sch.json
  python - <<PY
import json
d=json.load(open("$O/soak_$sch.json"))
print("$sch", d["hip_frames_per_s"], d.get("hip_vs_oracle_seq0"), d.get("first_stat_difference"), {k:d["per_sequence"][0][k] for k in ("keyframes_created","keyframes_culled","local_bas","lost_frames","map_violations","ate_rmse_m")})
PY
done
