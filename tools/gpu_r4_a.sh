set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
python tools/mfma_f64_rate.py > gpurun_out/r4a/f64rate.json 2> gpurun_out/r4a/f64rate.err
cat gpurun_out/r4a/f64rate.json
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py -x -q > gpurun_out/r4a/pytest_lba.log 2>&1 || (tail -30 gpurun_out/r4a/pytest_lba.log; exit 1)
tail -3 gpurun_out/r4a/pytest_lba.log
MODES=1 NB=40 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r4a/lbaprof -o lw -- python3 tools/lba_win_prof.py > gpurun_out/r4a/lbaprof.log 2>&1
tail -3 gpurun_out/r4a/lbaprof.log
ls gpurun_out/r4a/lbaprof
