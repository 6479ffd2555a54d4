set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b_ctrl_pose; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_lba_gpu.py tests/test_poseopt_gpu.py -x -q -m gpu > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -1 $O/test.log
for nb in 40 128; do
    NB=$nb MODES=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$nb -o kt -- python3 tools/lba_win_prof.py > $O/kt_$nb.log 2>&1
    f=$(find $O/kt_$nb -name "*kernel_stats.csv" | head -1)
    python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
d={r["Name"].split("(")[0].replace("void ","").replace("oslam::",""): float(r["AverageNs"])/1e3 for r in rows}
keys=["k_w_schur_rec","k_w_chol_lds_mfma","k_w_lin<true>","k_w_update<true, false, 2>","k_w_ctrlB","k_w_edgeW<true>"]
print($nb, " ".join("%s %.1f" % (k, d.get(k, 0)) for k in keys), "trial %.1f" % sum(d.get(k, 0) for k in keys), flush=True)
PY
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pose -o kt -- python3 tools/pose_prof.py > $O/pose.log 2>&1
tail -5 $O/pose.log
f=$(find $O/pose -name "*kernel_stats.csv" | head -1); head -6 $f | cut -d, -f1-4 | cut -c1-160
