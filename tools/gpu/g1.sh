set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > gpurun_out/r05_lba_tests_a.log 2>&1 || { tail -30 gpurun_out/r05_lba_tests_a.log; exit 1; }
tail -3 gpurun_out/r05_lba_tests_a.log
NB=40,160 MODES=1 python tools/lba_win_prof.py > gpurun_out/r05_prof_rec1.log 2>&1
OSLAM_LBA_REC=0 NB=40,160 MODES=1 python tools/lba_win_prof.py > gpurun_out/r05_prof_rec0.log 2>&1
cat gpurun_out/r05_prof_rec1.log gpurun_out/r05_prof_rec0.log
cd /tmp && NB=40 MODES=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_prof_a -o rec1 -- python $GRAFT_REPO_ROOT/tools/lba_win_prof.py > /dev/null 2>&1
