cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4trace2
mkdir -p $O
export HIP_FORCE_DEV_KERNARG=0
lm=deferred
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/$lm -o bt -- python3 bench.py --lm $lm --seqs 1024 --handles 2 --preroll 200 --no-extras --no-cpu-baseline > $O/bench_$lm.json 2> $O/bench_$lm.err
echo "rc=$? ($lm)"
DB=$(find $O/$lm -name "*results.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_kernel_stats.py $DB > $O/kernel_stats_$lm.csv; head -16 $O/kernel_stats_$lm.csv | cut -c1-150; rm -f $DB; fi
MODES=1 NB=40 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/lbaprof -o lw -- python3 tools/lba_win_prof.py > $O/lbaprof.log 2>&1
grep mode $O/lbaprof.log | cut -c1-130
python tools/rocpd_kernel_stats.py $O/lbaprof/lw_results.db > $O/lw_kernel_stats.csv; rm -f $O/lbaprof/lw_results.db
head -9 $O/lw_kernel_stats.csv | cut -c1-140
POSE_PROF_B=4096 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/poseprof -o pp -- python3 tools/pose_prof.py > $O/poseprof.log 2>&1
python tools/rocpd_kernel_stats.py $O/poseprof/pp_results.db > $O/pose_kernel_stats.csv; rm -f $O/poseprof/pp_results.db
head -4 $O/pose_kernel_stats.csv | cut -c1-160
