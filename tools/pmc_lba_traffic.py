"""Summary of two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) over tools/lba_win_prof.py (MODES=1 NB=40): memory-side bytes per launch of every local-BA kernel.
usage: pmc_lba_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json"""
import csv, collections, json, sys

def load(path, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"].split("(")[0].replace("oslam::", "").replace("void ", "")].append(float(r["Counter_Value"]))
    return acc

f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(f):
    fk = sum(f[k]) / len(f[k]); wk = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    out[k] = {"launches": len(f[k]), "FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1), "bytes_per_launch": round((fk + wk) * 1024)}
per_trial = ["k_w_lin", "k_w_ctrlA", "k_w_edgeW", "k_w_schur", "k_w_schur_rec", "k_w_chol_lds_mfma", "k_w_chol_packed", "k_w_update", "k_w_ctrlB"]
base = lambda k: k.split("<")[0]   # (k_w_schur<true>: template arguments are not part of the role)
trial_keys = [k for k in out if base(k) in per_trial]
trial_bytes = sum(out[k]["bytes_per_launch"] for k in trial_keys)
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on tools/lba_win_prof.py MODES=1 NB=40: one call = 40 steady-state-shaped windows "
             "(27 keyframes, 1500 points, 13 075 edges each), multi-launch layout, pair gather over COMPACT 32-BYTE EDGE RECORDS (round 5: k_w_lin<true> stores (x, y, 1/z, weight) "
             "per edge instead of the 144-byte B_e block, k_w_schur_rec forms the pair products from them; per-landmark inverses; 6 launches per trial); KB per launch, mean over the "
             "launches of the run.  FETCH_SIZE is reported RAW: MI355X_MICROARCH.md's x2 correction holds for wide coalesced streaming reads; these kernels gather 16-byte pieces per "
             "lane.  Infinity-Cache hits are counted.",
           "kernels": out, "bytes_per_lm_trial_40_windows": trial_bytes, "launches_per_trial": len(trial_keys),
           "mean_bytes_per_launch_of_a_trial": round(trial_bytes / max(1, len(trial_keys)))}, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print("%-28s n=%4d fetch %8.2f MB write %8.2f MB" % (k, v["launches"], v["FETCH_SIZE_KB"] / 1024, v["WRITE_SIZE_KB"] / 1024))
print("per LM trial (40 windows): %.1f MB" % (trial_bytes / 1e6))
