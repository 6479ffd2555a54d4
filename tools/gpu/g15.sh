set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for S in 0 4; do
echo "SOLVER=$S"; SOLVER=$S NB=25,100 MODES=1 python tools/lba_win_prof.py 10 0 2200 4 2>&1 | grep windows
done
R=$GRAFT_REPO_ROOT
cd /tmp && SOLVER=0 NB=100 MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_small -o s0 -- python $R/tools/lba_win_prof.py 10 0 2200 4 > /dev/null 2>&1
