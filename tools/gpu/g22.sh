set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > gpurun_out/r05_lba_tests_g.log 2>&1 || { tail -30 gpurun_out/r05_lba_tests_g.log; exit 1; }
tail -2 gpurun_out/r05_lba_tests_g.log
NB=40,128 MODES=1 python tools/lba_win_prof.py 2>&1 | grep windows
R=$GRAFT_REPO_ROOT
cd /tmp && NB=40 MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_h -o nb40 -- python3 $R/tools/lba_win_prof.py > /dev/null 2>&1
cd /tmp && NB=128 MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_h -o nb128 -- python3 $R/tools/lba_win_prof.py > /dev/null 2>&1
cd $R
for f in $(find gpurun_out/r05_prof_h -name "*kernel_stats.csv"); do echo $f; head -8 $f | cut -d, -f1-4 | sed 's/(oslam::LbaProblem[^"]*"/"/' | cut -c1-90; done
