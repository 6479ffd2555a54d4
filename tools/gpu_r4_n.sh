set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4n
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > $O/pytest_lba.log 2>&1 || (tail -30 $O/pytest_lba.log; exit 1)
tail -2 $O/pytest_lba.log
echo "34-KF windows (n = 198: global-memory matrix-core solver)"
MODES=1 NB=40 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/lbaprof -o lw -- python3 tools/lba_win_prof.py 34 0 2000 17 > $O/lbaprof.log 2>&1
grep mode $O/lbaprof.log | cut -c1-130
python tools/rocpd_kernel_stats.py $O/lbaprof/lw_results.db > $O/lw34_kernel_stats.csv
head -6 $O/lw34_kernel_stats.csv | cut -c1-130
echo "mode 2, 27-KF"
MODES=2 NB=256 python3 tools/lba_win_prof.py | grep mode | cut -c1-130
