cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4lba
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > $O/pytest_lba.log 2>&1 || { tail -30 $O/pytest_lba.log; exit 1; }
tail -2 $O/pytest_lba.log
for v in 0 1; do
  OSLAM_LBA_SCHUR_VINV=$v MODES=1 NB=40,128,256 timeout -k 10 300 python3 tools/lba_win_prof.py 2>&1 | grep mode | cut -c1-110
done
MODES=1 NB=128 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/lbaprof -o lw -- python3 tools/lba_win_prof.py > $O/lbaprof.log 2>&1
python tools/rocpd_kernel_stats.py $O/lbaprof/lw_results.db > $O/lw_kernel_stats.csv; rm -f $O/lbaprof/lw_results.db
head -8 $O/lw_kernel_stats.csv | cut -c1-150
