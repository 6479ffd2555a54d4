set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > gpurun_out/r05_lba_tests_e.log 2>&1 || { tail -30 gpurun_out/r05_lba_tests_e.log; exit 1; }
tail -2 gpurun_out/r05_lba_tests_e.log
NB=40,128 MODES=1 python tools/lba_win_prof.py 2>&1 | grep windows
R=$GRAFT_REPO_ROOT
for NB in 40; do
cd /tmp && NB=$NB MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_g -o nb$NB -- python $R/tools/lba_win_prof.py > /dev/null 2>&1
done
