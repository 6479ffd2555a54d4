set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m object_slam_amd.build > /dev/null 2>&1
cd /tmp && OSLAM_LBA_REC=${REC:-1} NB=${NB:-128} MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_d -o nb${NB:-128} -- python $R/tools/lba_win_prof.py > /dev/null 2>&1 || true
