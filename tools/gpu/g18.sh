cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_SLAM_CULL_CHECK=1 timeout -k 10 1100 python -m pytest tests/test_slam_driver_gpu.py tests/test_adapter_gpu.py tests/test_examples_gpu.py tests/test_mp_table_gpu.py -x -q -s > gpurun_out/r05_cullcheck.log 2>&1
grep -n "CULL_CHECK\|Fatal\|passed\|failed" gpurun_out/r05_cullcheck.log | head -10
