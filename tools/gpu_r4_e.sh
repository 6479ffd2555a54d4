set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4e
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q -k "headline or mixing or panel_edges or s5_large or matrix_core or schedule" > gpurun_out/r4e/pytest_lba.log 2>&1 || (tail -30 gpurun_out/r4e/pytest_lba.log; exit 1)
tail -2 gpurun_out/r4e/pytest_lba.log
MODES=1 NB=40 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r4e/lbaprof -o lw -- python3 tools/lba_win_prof.py > gpurun_out/r4e/lbaprof.log 2>&1
grep mode gpurun_out/r4e/lbaprof.log
python tools/rocpd_kernel_stats.py gpurun_out/r4e/lbaprof/lw_results.db > gpurun_out/r4e/lw_kernel_stats.csv
head -9 gpurun_out/r4e/lw_kernel_stats.csv | cut -c1-110
OSLAM_LBA_PROFILE=1 python object_slam_amd/build.py -f > gpurun_out/r4e/build.log 2>&1
python tools/chol_lds_phase_prof.py 27
