set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && NB=${NB:-40} MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_a -o ${TAG:-rec1} -- python $R/tools/lba_win_prof.py > $R/gpurun_out/r05_prof_a_${TAG:-rec1}.log 2>&1
ls $R/gpurun_out/r05_prof_a
head -20 $R/gpurun_out/r05_prof_a/${TAG:-rec1}_kernel_stats.csv
