set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4fe
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 tools/frontend_pmc.py 3 > $O/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 tools/frontend_pmc.py 3 > $O/pf.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 tools/frontend_pmc.py 3 > $O/pw.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/ps -o ps -- python3 tools/frontend_pmc.py 3 > $O/ps.log 2>&1
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1)
PF=$(find $O/pf -name "*counter_collection.csv" | head -1)
PW=$(find $O/pw -name "*counter_collection.csv" | head -1)
PS=$(find $O/ps -name "*counter_collection.csv" | head -1)
head -1 $KT
python tools/pmc_frontend_traffic.py $O/r04_pmc_traffic.json $KT $PF $PW $PS | cut -c1-260
tail -1 $O/kt.log | cut -c1-600
