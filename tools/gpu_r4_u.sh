cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_r4_pmc.sh > gpurun_out/r4pmc_final.log 2>&1; tail -18 gpurun_out/r4pmc_final.log
O=gpurun_out/r4k2
mkdir -p $O
export MODES=1 NB=40
timeout -k 10 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/p2 -o p2 -- python3 tools/lba_win_prof.py > $O/p2.log 2>&1 || tail -5 $O/p2.log
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/p1 -o p1 -- python3 tools/lba_win_prof.py > $O/p1.log 2>&1 || tail -5 $O/p1.log
python3 - <<'PY'
import csv, glob, collections, json
res={}
for d in ("p1","p2"):
    for f in glob.glob("gpurun_out/r4k2/%s/*counter_collection.csv"%d):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("oslam::","").replace("void ","")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in ("k_w_schur","k_w_lin","k_w_edgeW","k_w_update","k_w_chol_lds_mfma"):
            if k in acc: res.setdefault(k,{}).update({c: round(sum(v)/len(v)) for c,v in acc[k].items()})
json.dump(res, open("gpurun_out/r4k2/r04_pmc_lba_wait_ta.json","w"), indent=1)
for k,v in res.items(): print(k, v)
PY
