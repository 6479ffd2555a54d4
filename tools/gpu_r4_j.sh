set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4j
timeout -k 10 600 python -m pytest tests/test_slam_driver_gpu.py -q -k "200_frames" > gpurun_out/r4j/pytest.log 2>&1 || (tail -30 gpurun_out/r4j/pytest.log; exit 1)
tail -2 gpurun_out/r4j/pytest.log
bash tools/gpu_r4_pmc.sh > gpurun_out/r4j/pmc.log 2>&1 || (tail -20 gpurun_out/r4j/pmc.log; exit 1)
tail -16 gpurun_out/r4j/pmc.log
