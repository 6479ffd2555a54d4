cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)\s*:\s*\S+|\b(TA_[A-Z_a-z0-9]+|TCP_[A-Z_a-z0-9]+|SQ_WAIT[A-Z_a-z0-9]*|SQ_ACTIVE_INST[A-Z_a-z0-9]*|SQ_INST_CYCLES[A-Z_a-z0-9]*|SQ_WAVE_CYCLES|SQ_INSTS_VMEM[A-Z_a-z0-9]*|SQ_INSTS_LDS|SQ_LDS[A-Z_a-z0-9]*)\b" | sort -u | tr '\n' ' ' > $O/counters.txt
wc -c $O/counters.txt
export MODES=1 NB=40
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/p1 -o p1 -- python3 tools/lba_win_prof.py > $O/p1.log 2>&1 || tail -5 $O/p1.log
timeout -k 10 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/p2 -o p2 -- python3 tools/lba_win_prof.py > $O/p2.log 2>&1 || tail -5 $O/p2.log
python3 - <<'PY'
import csv, glob, collections
for d in ("p1","p2"):
    for f in glob.glob("gpurun_out/r4k/%s/*counter_collection.csv"%d):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("oslam::","").replace("void ","")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in ("k_w_schur","k_w_lin","k_w_edgeW","k_w_update"):
            if k in acc: print(k, {c: round(sum(v)/len(v)) for c,v in acc[k].items()})
PY
