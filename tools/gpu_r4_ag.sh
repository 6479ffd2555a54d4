cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4ag
mkdir -p $O
for sch in 0 1; do
SCHUR=$sch MODES=1 NB=128,256 timeout -k 10 300 python3 tools/lba_win_prof.py 2>&1 | grep mode | cut -c1-110
done
SCHUR=1 MODES=1 NB=128 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/lbaprof -o lw -- python3 tools/lba_win_prof.py > $O/lbaprof.log 2>&1
python tools/rocpd_kernel_stats.py $O/lbaprof/lw_results.db > $O/lw_kernel_stats_tiles.csv; rm -f $O/lbaprof/lw_results.db
head -9 $O/lw_kernel_stats_tiles.csv | cut -c1-140
