cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4ac
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > $O/pytest_lba.log 2>&1 || { tail -30 $O/pytest_lba.log; exit 1; }
tail -2 $O/pytest_lba.log
for v in default LW3; do
  if [ $v = default ]; then unset OSLAM_LIB_PATH; else export OSLAM_LIB_PATH=$PWD/tools/_build/liboslam_hip_$v.so; fi
  MODES=1 NB=40,160 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/lbaprof_$v -o lw -- python3 tools/lba_win_prof.py > $O/lbaprof_$v.log 2>&1
  grep mode $O/lbaprof_$v.log | cut -c1-130
  python tools/rocpd_kernel_stats.py $O/lbaprof_$v/lw_results.db > $O/lw_kernel_stats_$v.csv
  head -7 $O/lw_kernel_stats_$v.csv | cut -c1-130
done
