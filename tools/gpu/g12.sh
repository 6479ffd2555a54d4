set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lba_gpu.py tests/test_slam_driver_gpu.py -x -q -k "refuses or confined" 2>&1 | tail -25
