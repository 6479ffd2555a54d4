set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r5final
mkdir -p $O
export MODES=1 NB=40
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 tools/lba_win_prof.py > $O/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/p1 -o p1 -- python3 tools/lba_win_prof.py > $O/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 tools/lba_win_prof.py > $O/pf.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 tools/lba_win_prof.py > $O/pw.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -o p2 -- python3 tools/lba_win_prof.py > $O/p2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/p3 -o p3 -- python3 tools/lba_win_prof.py > $O/p3.log 2>&1
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/pmc_mfma_summary.py $O/r05_pmc_lba_mfma.json $KT $(find $O/p1 -name "*counter_collection.csv" | head -1) | grep -E "chol|schur|k_w_lin" | cut -c1-300
python tools/pmc_lba_traffic.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) $O/r05_pmc_lba_traffic.json | tail -3
python tools/pmc_summary.py $O/r05_pmc_lba_wait_ta.json $(find $O/p2 $O/p3 -name "*counter_collection.csv") > $O/summary.txt
cp $O/kt/kt_kernel_stats.csv $O/r05_lba_win_prof_40windows_kernel_stats.csv
export NB=128
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt128 -o kt -- python3 tools/lba_win_prof.py > $O/kt128.log 2>&1
cp $O/kt128/kt_kernel_stats.csv $O/r05_lba_win_prof_128windows_kernel_stats.csv
head -8 $O/r05_lba_win_prof_40windows_kernel_stats.csv | sed 's/(oslam::LbaProblem[^"]*"/"/' | cut -d, -f1-5
