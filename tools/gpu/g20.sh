set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > gpurun_out/r05_lba_tests_f.log 2>&1 || { tail -30 gpurun_out/r05_lba_tests_f.log; exit 1; }
tail -2 gpurun_out/r05_lba_tests_f.log
echo "fold on"; NB=40,128 MODES=1 python tools/lba_win_prof.py 2>&1 | grep windows
echo "fold off"; OSLAM_LBA_FOLD_CTRL=0 NB=40,128 MODES=1 python tools/lba_win_prof.py 2>&1 | grep windows
echo "fold on"; NB=40,128 MODES=1 python tools/lba_win_prof.py 2>&1 | grep windows
