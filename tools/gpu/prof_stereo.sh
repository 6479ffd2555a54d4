set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5stereo
mkdir -p $O
cd /tmp && HIP_FORCE_DEV_KERNARG=0 timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stereo -- python3 $R/bench.py --workload stereo --seqs 512 --handles 2 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; }
cd $R
head -40 $O/stereo_kernel_stats.csv | sed 's/(oslam::[^"]*"/"/' | cut -c1-160
tail -1 $O/bench.json | cut -c1-300
# front-end PMC passes (S2 stage at 512 frames per launch)
O2=gpurun_out/r5fe
mkdir -p $O2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O2/kt -o kt -- python3 tools/frontend_pmc.py 3 > $O2/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O2/pf -o pf -- python3 tools/frontend_pmc.py 3 > $O2/pf.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O2/pw -o pw -- python3 tools/frontend_pmc.py 3 > $O2/pw.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O2/ps -o ps -- python3 tools/frontend_pmc.py 3 > $O2/ps.log 2>&1
python tools/pmc_frontend_traffic.py $O2/r05_pmc_traffic.json $(find $O2/kt -name "*kernel_trace.csv" | head -1) $(find $O2/pf -name "*counter_collection.csv" | head -1) $(find $O2/pw -name "*counter_collection.csv" | head -1) $(find $O2/ps -name "*counter_collection.csv" | head -1) | cut -c1-200
