set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5pmc
mkdir -p $O
export MODES=1 NB=${NB:-40}
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/p1 -o p1 -- python3 tools/lba_win_prof.py > $O/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/p2 -o p2 -- python3 tools/lba_win_prof.py > $O/p2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/p3 -o p3 -- python3 tools/lba_win_prof.py > $O/p3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 tools/lba_win_prof.py > $O/pf.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 tools/lba_win_prof.py > $O/pw.log 2>&1
python tools/pmc_summary.py $O/r05_pmc_lba_counters.json $(find $O/p1 $O/p2 $O/p3 -name "*counter_collection.csv") > $O/summary.txt
python tools/pmc_lba_traffic.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/r05_pmc_lba_traffic.json | tail -15
