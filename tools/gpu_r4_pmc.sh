set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4pmc
mkdir -p $O
export MODES=1 NB=40
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 tools/lba_win_prof.py > $O/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/p1 -o p1 -- python3 tools/lba_win_prof.py > $O/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 tools/lba_win_prof.py > $O/pf.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 tools/lba_win_prof.py > $O/pw.log 2>&1
find $O -name "*.csv" | head -20
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1)
P1=$(find $O/p1 -name "*counter_collection.csv" | head -1)
PF=$(find $O/pf -name "*counter_collection.csv" | head -1)
PW=$(find $O/pw -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma_summary.py $O/r04_pmc_lba_mfma.json $KT $P1 | grep -E "chol|schur|k_w_lin"
python tools/pmc_lba_traffic.py $PF $PW $O/r04_pmc_lba_traffic.json
