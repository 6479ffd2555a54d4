set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4d
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q -k "headline or mixing or panel_edges or s5_large or matrix_core" > gpurun_out/r4d/pytest_lba.log 2>&1 || (tail -30 gpurun_out/r4d/pytest_lba.log; exit 1)
tail -3 gpurun_out/r4d/pytest_lba.log
MODES=1 NB=40,160 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r4d/lbaprof -o lw -- python3 tools/lba_win_prof.py > gpurun_out/r4d/lbaprof.log 2>&1
grep mode gpurun_out/r4d/lbaprof.log
python tools/rocpd_kernel_stats.py gpurun_out/r4d/lbaprof/lw_results.db > gpurun_out/r4d/lw_kernel_stats.csv
head -9 gpurun_out/r4d/lw_kernel_stats.csv | cut -c1-120
