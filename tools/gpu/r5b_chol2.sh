set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b_chol2; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_lba_gpu.py -x -q -m gpu > $O/test_lba.log 2>&1 || { tail -30 $O/test_lba.log; exit 1; }
tail -1 $O/test_lba.log
OSLAM_LIB_PATH=tools/_build/liboslam_hip_prof.so timeout -k 10 120 python tools/chol_lds_phase_prof.py 27 | tee $O/phase27.txt
OSLAM_LIB_PATH=tools/_build/liboslam_hip_prof.so timeout -k 10 120 python tools/chol_lds_phase_prof.py 31 | tee $O/phase31.txt
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
