set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > gpurun_out/r05_lba_tests_b.log 2>&1 || { tail -30 gpurun_out/r05_lba_tests_b.log; exit 1; }
tail -3 gpurun_out/r05_lba_tests_b.log
for R in 2 1 2 1; do
echo "REC=$R"; OSLAM_LBA_REC=$R NB=40,128 MODES=1 python tools/lba_win_prof.py 2>&1 | grep windows
done
R=$GRAFT_REPO_ROOT
cd /tmp && NB=40 MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof_b -o rec2 -- python $R/tools/lba_win_prof.py > /dev/null 2>&1
head -8 $R/gpurun_out/r05_prof_b/rec2_kernel_stats.csv
