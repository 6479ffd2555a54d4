set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q 2>&1 | tail -3
NB=40,128 MODES=1 python tools/lba_win_prof.py 2>&1 | grep windows
NB=100 MODES=1 python tools/lba_win_prof.py 10 0 2200 4 2>&1 | grep windows
R=$GRAFT_REPO_ROOT
cd /tmp && NB=128 MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_h -o nb128 -- python $R/tools/lba_win_prof.py > /dev/null 2>&1
